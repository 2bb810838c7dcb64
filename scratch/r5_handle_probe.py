"""scratch (for the round after 4): WHICH allocation makes the build's own handle slower to search than the same index created from its
arrays (profiles/r04_built_vs_created.txt)?  Uses only the public API:
  built                       the handle cz_hnsw_build returned
  built + new workspace       CZ_HNSW_VSLOTS doubled: the pooled visited workspace is too small, a fresh one is allocated (the build's
                              temporaries are gone by then); compare with `created + new workspace`
  built + one vector inserted cz_hnsw_insert of ONE vector re-allocates the vector table (n + 1 rows) and re-packs every link table
  created                     export -> destroy -> cz_hnsw_index_create
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
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)

    def timed(ix, tag, reps=3):
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(3): run()
        torch.cuda.synchronize()
        out = []
        for rep in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out.append(ms)
        tot = int(nd.sum().item())
        print(f"{tag:56s} {' '.join(f'{m:.3f}' for m in out)} ms/batch  {tot * 4 * dim / min(out) / 1e6 / 8000:.3f} of peak", flush=True)

    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"built {n} in {time.time() - t0:.1f}s", flush=True)
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    timed(ix, "built")
    os.environ["CZ_HNSW_VSLOTS"] = "65536"
    timed(ix, "built + new (larger) visited workspace")
    os.environ.pop("CZ_HNSW_VSLOTS")
    timed(ix, "built, pooled workspace again")
    nodes, nbrs, entry = ix.export()
    rng = np.random.default_rng(5)
    ix.insert(xh[:1] + rng.standard_normal((1, dim)).astype(np.float32) * 0.01, seed=9, max_batch=1)
    torch.cuda.synchronize()
    timed(ix, "built + ONE vector inserted (vector table and link tables re-allocated)")
    ix.close()
    torch.cuda.empty_cache()
    ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
    timed(ix2, "created from the exported arrays")
    os.environ["CZ_HNSW_VSLOTS"] = "65536"
    timed(ix2, "created + new (larger) visited workspace")
    os.environ.pop("CZ_HNSW_VSLOTS")
main()
