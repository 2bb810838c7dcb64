#!/bin/bash
# experimental build of libcozo_gpu with extra -D flags -> scratch/lib/libcozo_gpu_<name>.so  (COZO_GPU_LIB selects it)
#   scratch/build_variant.sh prof -DCZ_PHASE_TIMING ; scratch/build_variant.sh nt0 -DCZ_ROWS_NT=0
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
SRC=${SRC:-$R/cozo_amd/csrc}
mkdir -p $R/scratch/lib/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value $*"
for f in $(cd $SRC && ls *.hip | sed "s/\.hip$//"); do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $R/scratch/lib/obj_$NAME/$f.o &
done
wait
g++ -shared -fPIC $R/scratch/lib/obj_$NAME/*.o -o $R/scratch/lib/libcozo_gpu_$NAME.so
rm -rf $R/scratch/lib/obj_$NAME
echo built $R/scratch/lib/libcozo_gpu_$NAME.so
