"""scratch: which use of RCCL from libcozo_gpu makes the interpreter crash at exit ("double free or corruption")?
usage: r2_rccl_exit.py <mode>   mode: load | id | rank | rank_ag | multi"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cozo_amd import _lib
L = _lib.lib()
assert L.cz_init(0) == 0
mode = sys.argv[1]
from cozo_amd.comm import Comm, pagerank_multi
if mode == "load":
    pass
elif mode == "id":
    Comm.unique_id()
elif mode in ("rank", "rank_ag"):
    c = Comm(Comm.unique_id(), 0, 1)
    if mode == "rank_ag":
        import torch
        x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
        c.all_gather(x, 4000)
        torch.cuda.synchronize()
    c.close()
elif mode == "torch_only":
    import torch
    x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
elif mode == "rank_torch":
    import torch
    x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    c = Comm(Comm.unique_id(), 0, 1)
    torch.cuda.synchronize()
    c.close()
elif mode == "rank_ag_stream":
    import torch
    x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    c = Comm(Comm.unique_id(), 0, 1)
    c.all_gather(x, 4000, st.cuda_stream)
    torch.cuda.synchronize()
    c.close()
elif mode == "rank_ar":
    import torch
    e = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda:0")
    c = Comm(Comm.unique_id(), 0, 1)
    c.all_reduce_sum_f64(e, 2)
    torch.cuda.synchronize()
    c.close()
elif mode == "sharded":
    from cozo_amd import graph as G
    off = np.array([0, 1, 2, 3], dtype=np.uint32)
    src = np.array([1, 2, 0], dtype=np.uint32)
    od = np.array([1, 1, 1], dtype=np.uint32)
    plan = G.PageRankPlan(off, src, od, 3, 0, 3, 0.85)
    c = Comm(Comm.unique_id(), 0, 1)
    print(c.pagerank_sharded(plan, 3, 1e-4, 10))
    c.close()
    plan.close()
elif mode == "sharded_torch":
    import torch
    x = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    from cozo_amd import graph as G
    off = np.array([0, 1, 2, 3], dtype=np.uint32)
    src = np.array([1, 2, 0], dtype=np.uint32)
    od = np.array([1, 1, 1], dtype=np.uint32)
    plan = G.PageRankPlan(off, src, od, 3, 0, 3, 0.85)
    c = Comm(Comm.unique_id(), 0, 1)
    print(c.pagerank_sharded(plan, 3, 1e-4, 10))
    c.close()
    plan.close()
elif mode == "multi":
    off = np.array([0, 1, 2, 3], dtype=np.uint32)
    src = np.array([1, 2, 0], dtype=np.uint32)
    od = np.array([1, 1, 1], dtype=np.uint32)
    print(pagerank_multi(off, src, od, 1))
print("done", mode, flush=True)
