"""scratch (round 5): placement by trial (cz_hnsw_index_settle) on the 10M index: random landings of created handles, each settled
afterwards; then handles created with the automatic settle.  HS_N x 768, ef 144, batch 1024."""
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
        print(f"{tag:64s} {' '.join(f'{m:.3f}' for m in out)} ms  {tot * 4 * dim / min(out) / 1e6 / 8000:.3f} of peak", flush=True)
        return ids.clone(), dd.clone()

    os.environ["CZ_TABLE_SETTLE"] = "0"
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"built {n} in {time.time() - t0:.1f}s (no settle)", flush=True)
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    ref_ids, ref_dd = timed(ix, "built, as it landed")
    t0 = time.time()
    print("   settle ->", ix.settle(ef=ef, trials=3), f"{time.time() - t0:.1f}s", flush=True)
    a, b = timed(ix, "built, settled")
    assert torch.equal(a, ref_ids) and torch.equal(b, ref_dd)
    nodes, nbrs, entry = ix.export()
    ix.close()
    torch.cuda.empty_cache()
    MB = 1 << 20
    for i, js in enumerate([0, 3 * MB + 4096, 0, 5 * 1024 * MB + 4096, 0, 0]):
        junk = torch.empty(js, dtype=torch.uint8, device=dev) if js else None
        ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
        del junk
        torch.cuda.empty_cache()
        timed(ix2, f"created #{i}, as it landed")
        t0 = time.time()
        print("   settle ->", ix2.settle(ef=ef, trials=3), f"{time.time() - t0:.1f}s", flush=True)
        a, b = timed(ix2, f"created #{i}, settled")
        assert torch.equal(a, ref_ids) and torch.equal(b, ref_dd)
        print("   report ->", ix2.settle(ef=ef, trials=0), flush=True)
        timed(ix2, f"created #{i}, settled (again)")
        ix2.close()
        torch.cuda.empty_cache()
    os.environ.pop("CZ_TABLE_SETTLE")
    for i in range(3):
        t0 = time.time()
        ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
        dt = time.time() - t0
        timed(ix2, f"created with the automatic settle #{i} ({dt:.1f}s, {ix2.settle()})")
        ix2.close()
        torch.cuda.empty_cache()


main()
