"""scratch: libcozo_gpu.so with graph.hip compiled under extra -D flags -> scratch/lib/libcozo_gpu_<tag>.so (COZO_GPU_LIB selects it).
    python scratch/build_variant.py <tag> -DNAME=VALUE ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import build as B
B.build()
tag, defs = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "scratch", "lib"), exist_ok=True)
obj = os.path.join(ROOT, "scratch", "lib", f"graph_{tag}.o")
subprocess.check_call([B.HIPCC, *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, "graph.hip"), "-o", obj])
objs = [os.path.join(B.OBJDIR, os.path.basename(s) + ".o") for s in B._sources() if not s.endswith("graph.hip")] + [obj]
so = os.path.join(ROOT, "scratch", "lib", f"libcozo_gpu_{tag}.so")
subprocess.check_call(["g++", "-shared", "-fPIC", *objs, "-o", so])
os.remove(obj)
print(so)
