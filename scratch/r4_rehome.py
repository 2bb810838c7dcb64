"""scratch (round 4): is the search slower on an index that was BUILT in this process than on the same graph handed to
cz_hnsw_index_create in this process?  (the bench line measured 0.64-0.69 of the peak after an in-process build and 0.67-0.72 on the
same graph loaded by a fresh process.)  build -> time -> export -> destroy -> create from the arrays -> time, same queries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k, B, ef = int(os.environ.get("HS_N", 10_000_000)), 768, 10, 1024, int(os.environ.get("HS_EFS", 144))
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)

    def timed(ix, tag):
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            tot = int(nd.sum().item())
            print(f"{tag}: {ms:.3f} ms/batch  {tot * 4 * dim / ms / 1e6 / 8000:.3f} of peak  probe {tuple(round(v) for v in ix.hbm_probe())}", flush=True)
        return ids.clone(), dd.clone()

    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"built in {time.time() - t0:.1f}s", flush=True)
    r1 = timed(ix, "built here, corpus still resident")
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    r1b = timed(ix, "built here, corpus freed")
    if os.environ.get("HS_COMPACT"):
        ix.compact()
        r1c = timed(ix, "built here, link tables moved to fresh allocations (cz_hnsw_index_compact)")
        print("same results:", bool(torch.equal(r1[0], r1c[0]) and torch.equal(r1[1], r1c[1])), flush=True)
        return
    nodes, nbrs, entry = ix.export()
    ix.close()
    torch.cuda.empty_cache()
    ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
    r2 = timed(ix2, "same graph through cz_hnsw_index_create")
    print("same results:", bool(torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])), flush=True)
    # and once more: a second create while the first is still alive (allocated later, next to it)
    ix3 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
    timed(ix3, "a second copy created beside it")
    timed(ix2, "the first copy again")
main()
