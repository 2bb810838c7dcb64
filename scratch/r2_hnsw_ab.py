"""scratch (round 2): A/B of the traversal's knobs on the bench corpus, one process, one box:
  build:  CZ_BUILD_LAZY = 1 | 0 (lazy / eager shrinking) -> build time, distance evaluations, recall ladder
  search: CZ_HNSW_SPEC = 1 | 0 (next expansion prepared during eval) x CZ_HNSW_VISITED = hash | bitmap
Every search variant must return bit-identical ids / distances / n_dist on the same index."""
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
    n, dim, k, B = int(os.environ.get("HS_N", 1_000_000)), 768, 10, 1024
    kind = os.environ.get("HS_DIST", "lowrank")
    stream = torch.cuda.current_stream().cuda_stream
    x = Bn.gen_vectors(torch, n, dim, kind, 42, dev)
    q = Bn.gen_vectors(torch, B, dim, kind, 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev)
    dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    nd = torch.zeros(B, dtype=torch.int64, device=dev)
    gt = torch.empty((B, k), dtype=torch.int32, device=dev)
    gtd = torch.empty((B, k), dtype=torch.float64, device=dev)
    for lazy in os.environ.get("HS_LAZY", "1,0").split(","):
        os.environ["CZ_BUILD_LAZY"] = lazy
        t0 = time.time()
        ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
        torch.cuda.synchronize()
        bs = time.time() - t0
        print(f"build lazy={lazy}: {bs:.1f}s n_dist {ix.last_build_n_dist:.3e} ({ix.last_build_n_dist / n:.0f}/vector)", flush=True)
        ix.bruteforce_knn_device(q, k, gt, gtd, stream, gemm=True)
        torch.cuda.synchronize()
        gt64 = gt.to(torch.int64) & 0xFFFFFFFF
        for ef in (64, 96, 128, 160, 192):
            ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
            torch.cuda.synchronize()
            print(f"  ef={ef}: recall {Bn.recall_at_k(torch, ids.to(torch.int64) & 0xFFFFFFFF, gt64):.4f} n_dist/q {nd.sum().item() / B:.0f}", flush=True)
        ef = int(os.environ.get("HS_EF", 96))
        ref = None
        for rep in range(int(os.environ.get("HS_REPEAT", 2))):
            for spec, vis in (("1", "hash"), ("0", "hash"), ("1", "bitmap"), ("0", "bitmap")):
                os.environ["CZ_HNSW_SPEC"] = spec
                os.environ["CZ_HNSW_VISITED"] = vis
                def run():
                    ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                tot = int(nd.sum().item())
                if ref is None: ref = (ids.clone(), dd.clone(), nd.clone()); same = None
                else: same = bool(torch.equal(ref[0], ids) and torch.equal(ref[1], dd) and torch.equal(ref[2], nd))
                print(f"  search spec={spec} visited={vis} ef={ef}: {ms:.3f} ms, {B / ms * 1e3:.0f} q/s, {tot * 4 * dim / ms / 1e6:.0f} GB/s ({tot * 4 * dim / ms / 1e6 / 8000:.3f}) same_as_first={same}", flush=True)
        del os.environ["CZ_HNSW_SPEC"], os.environ["CZ_HNSW_VISITED"]
        ix.close()
main()
