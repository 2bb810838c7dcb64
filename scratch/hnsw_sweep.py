"""scratch: hnsw_knn_kernel knobs on the bench index (1M x 768 cosine): rows in flight per lane group (CZ_HNSW_U),
batch size (tail effect), per-query n_dist distribution; plus the batched-distance microbench (random pairs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch, distance_batch_device

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k = int(os.environ.get("HS_N", 1_000_000)), 768, 10
    ef = int(os.environ.get("HS_EF", 96))
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    # ---- batched distance: P random (query, base row) pairs, 4*d bytes each
    for P in (1 << 20, 1 << 22):
        g = torch.Generator(device=dev); g.manual_seed(1)
        q = Bn.gen_vectors(torch, 1024, dim, "lowrank", 43, dev)
        pairs = torch.stack([torch.randint(0, 1024, (P,), generator=g, device=dev, dtype=torch.int32),
                             torch.randint(0, n, (P,), generator=g, device=dev, dtype=torch.int32)], 1).contiguous()
        out = torch.empty(P, dtype=torch.float64, device=dev)
        for metric in ("Cosine", "L2"):
            for _ in range(2):
                distance_batch_device(metric, x, q, pairs, out, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                distance_batch_device(metric, x, q, pairs, out, stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"distance_batch {metric} P={P}: {ms:.3f} ms, {P * 4 * dim / ms / 1e6:.0f} GB/s ({P * 4 * dim / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"build {time.time() - t0:.1f}s n_dist {ix.last_build_n_dist:.3e}", flush=True)
    del x
    BMAX = 8192
    q = torch.cat([Bn.gen_vectors(torch, 1024, dim, "lowrank", 43 + i, dev) for i in range(BMAX // 1024)])
    ids = torch.empty((BMAX, k), dtype=torch.int32, device=dev)
    dd = torch.empty((BMAX, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(BMAX, dtype=torch.int32, device=dev)
    nd = torch.zeros(BMAX, dtype=torch.int64, device=dev)
    ref = None
    variants = [(U, "0") for U in os.environ.get("HS_US", "0").split(",")]
    variants = variants * int(os.environ.get("HS_REPEAT", "1"))
    for U, fl in variants:
        os.environ["CZ_HNSW_U"] = U
        for B in [int(b) for b in os.environ.get("HS_BS", "1024,4096,8192").split(",")]:
            def run():
                ix.hnsw_knn_batch_device(q[:B], HnswSearch(k=k, ef=ef), ids[:B], dd[:B], cnt[:B], nd[:B], stream)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 if B == 1024 else 4
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot = int(nd[:B].sum().item())
            same = None
            if B == 1024:
                if ref is None: ref = (ids[:B].clone(), dd[:B].clone(), nd[:B].clone())
                else: same = bool(torch.equal(ref[0], ids[:B]) and torch.equal(ref[1], dd[:B]) and torch.equal(ref[2], nd[:B]))
            ndf = nd[:B].to(torch.float64)
            qs = torch.quantile(ndf, torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev)).tolist()
            print(f"U={U} flags={fl} B={B}: {ms:.3f} ms, {B / ms * 1e3:.0f} q/s, {tot * 4 * dim / ms / 1e6:.0f} GB/s ({tot * 4 * dim / ms / 1e6 / 8000:.3f}), "
                  f"n_dist mean {ndf.mean().item():.0f} p50 {qs[0]:.0f} p90 {qs[1]:.0f} p99 {qs[2]:.0f} max {ndf.max().item():.0f}, same_as_first={same}", flush=True)
main()
