"""scratch (round 2): cz_distance_batch on the bench corpus at 1M and 10M base rows, knobs CZ_RUNS_STRETCH / CZ_PAIRS_GROUPED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
stream = torch.cuda.current_stream().cuda_stream
args = Bn.parse()
q = Bn.gen_vectors(torch, 1024, 768, "lowrank", 43, dev)
for n in (1_000_000, 10_000_000):
    x = Bn.gen_vectors(torch, n, 768, "lowrank", 42, dev)
    for env in ({}, {"CZ_RUNS_U": "2"}, {"CZ_PAIRS_GROUPED": "0"}):
        for k, v in env.items():
            os.environ[k] = v
        r = Bn.bench_distance_batch(args, torch, x, q, stream, dev)
        print(f"n={n} {env}: {r['ms']:.3f} ms frac {r['roofline']['frac']:.3f}", flush=True)
        for k in env:
            del os.environ[k]
    del x
    torch.cuda.empty_cache()
