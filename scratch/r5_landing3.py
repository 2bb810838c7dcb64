"""scratch (round 5): WHICH array's landing moves the search?  One 10M index; one of its arrays at a time is given a new place in device
memory (cz_debug_index_rehome: allocated while the old copy is still held, so it lands elsewhere), the search is timed after every move.
HS_N (default 10M) x 768, ef 144, batch 1024."""
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
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=int(os.environ.get("HS_EFC", 64)))
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)

    def timed(ix, tag):
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(4): run()
        torch.cuda.synchronize()
        out = []
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12): run()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 12)
        tot = int(nd.sum().item())
        va = int(L.cz_debug_index_table_address(ix._h))
        print(f"{tag:64s} table va 0x{va:012x}  {' '.join(f'{m:.3f}' for m in out)} ms  {tot * 4 * dim / min(out) / 1e6 / 8000:.3f} of peak", flush=True)

    def rehome(ix, what, contiguous):
        rc = L.cz_debug_index_rehome(ix._h, what, contiguous)
        assert rc == 0, L.cz_last_error()

    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"built {n} in {time.time() - t0:.1f}s", flush=True)
    del x
    torch.cuda.empty_cache()
    names = {0: "vector table", 1: "level-0 links", 2: "upper-level tables", 3: "visited workspaces dropped"}
    timed(ix, "built")
    seq = [(3, 1), (1, 1), (1, 1), (1, 0), (1, 1), (0, 1), (0, 1), (0, 0), (0, 1), (2, 1), (3, 1), (1, 1), (0, 1), (1, 0), (0, 0), (3, 1), (1, 1), (0, 1)]
    for what, contiguous in seq:
        rehome(ix, what, contiguous)
        timed(ix, f"  moved: {names[what]} -> {'contiguous' if contiguous else 'paged'}")


main()
