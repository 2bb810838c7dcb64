"""scratch: is the search time sensitive to WHERE the index lives in HBM?  One process, one index content, several
device copies created under different allocation histories (cz_hnsw_index_create from the exported host arrays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k, ef, B = 1_000_000, 768, 10, 96, 1024
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)
    def timeit(index, label):
        run = lambda: index.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{label:40s} {ms:.3f} ms  {int(nd.sum()) * 4 * dim / ms / 1e6:.0f} GB/s", flush=True)
    timeit(ix, "built in place")
    nodes, nbrs, entry = ix.export(); vec = ix.export_vectors()
    timeit(ix, "built in place (again)")
    copies = []
    for trial in range(4):
        junk = [torch.empty(int(s), dtype=torch.uint8, device=dev) for s in ([], [3 << 20, 77 << 20], [1 << 30], [5 << 20] * 40)[trial]]
        c = GpuHnswIndex(man, vec, [None] + nodes[1:], nbrs, entry)
        timeit(c, f"copy {trial} (junk allocations before: {len(junk)})")
        copies.append(c); del junk
    ix.close()
    torch.cuda.empty_cache()
    c = GpuHnswIndex(man, vec, [None] + nodes[1:], nbrs, entry)
    timeit(c, "copy after freeing the original")
    for cc in copies: timeit(cc, "earlier copy, re-timed")
main()
