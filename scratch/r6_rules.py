"""scratch (round 6): bench_graph_rules on one graph kind in isolation.  python scratch/r6_rules.py rmat|uniform [n] [e] [cpu]"""
import os, sys, time, types, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
assert L.cz_init(0) == 0
kind = sys.argv[1] if len(sys.argv) > 1 else "rmat"
args = types.SimpleNamespace(pr_nodes=int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000, pr_edges=int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000,
                             skip_cpu=not (len(sys.argv) > 4 and sys.argv[4] == "cpu"))
t0 = time.time()
out = Bn.bench_graph_rules(args, torch, torch.device("cuda:0"), kind=kind)
print("wall", time.time() - t0)
print(out.get("graph"))
for k in ("bfs", "connected_components", "sssp", "clustering_coefficients", "label_propagation"):
    o = out.get(k, {})
    print(k, {kk: o.get(kk) for kk in ("wall_ms", "device_ms", "colour_classes", "iterations", "levels", "reached", "components", "labels_left", "max_degree", "cancelled", "parity_checked")},
          json.dumps(o.get("cpu_baseline"))[:200] if o.get("cpu_baseline") else "")
