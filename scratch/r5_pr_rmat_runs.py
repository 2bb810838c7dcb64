"""scratch (round 5): the R-MAT PageRank sweep of bench.py alone, RUNS timed sweeps -- the process rocprofv3 --pmc wraps for the
pagerank_*_rmat entry of profiles/pmc_traffic.json.   python scratch/r5_pr_rmat_runs.py [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--skip-hnsw", "--skip-cpu", "--skip-secondary", "--pr-iters", sys.argv[1] if len(sys.argv) > 1 else "3"]
import bench
from cozo_amd import _lib
L = _lib.lib()
import torch
import torch.distributed as dist
args = bench.parse()
args.multi = False
torch.cuda.set_device(0)
assert L.cz_init(0) == 0
res = bench.bench_pagerank(args, torch, dist, 0, 1, torch.device("cuda", 0), kind="rmat")
print("rmat", res["form"], res["ms_per_iteration"], res["roofline"]["avg_launch_ms"], res["roofline"]["algorithmic_bytes_per_launch"])
