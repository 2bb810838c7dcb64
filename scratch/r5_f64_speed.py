"""scratch (round 5): the F64 search kernel's rate -- a 1M x 768 index built on f32 copies of the vectors, the SAME link tables over the f64
vectors (cz_hnsw_index_create_f64), batch 1024, ef 96: ms per batch, evaluations, fraction of 8 TB/s at 8 * d bytes per evaluation; the f32
handle beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
import bench as Bn
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
n, dim, k, B, ef = 1_000_000, 768, 10, 1024, 96
stream = torch.cuda.current_stream().cuda_stream
x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
nodes, nbrs, entry = ix.export()
xh = x.cpu().numpy()
man64 = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200, dtype="F64")
x64 = xh.astype(np.float64) + np.random.default_rng(1).standard_normal(xh.shape) * 1e-10
ix64 = GpuHnswIndex(man64, x64, nodes, nbrs, entry)
q64 = q.to(torch.float64)
for tag, h, qq in (("f32", ix, q), ("f64", ix64, q64)):
    ids = torch.empty((B, k), dtype=torch.int32, device=dev); dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.zeros(B, dtype=torch.int64, device=dev)
    h.settle(ef=ef, trials=3)
    run = lambda: h.hnsw_knn_batch_device(qq, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
    for _ in range(4): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ev = int(nd.sum().item())
    esz = 8 if tag == "f64" else 4
    print(f"{tag}: {ms:.3f} ms per batch of {B}, {ev / B:.0f} evaluations per query, {B / ms * 1e3:.0f} queries/s, {ev * esz * dim / ms / 1e6 / 8000:.3f} of 8 TB/s at {esz}*d bytes per evaluation", flush=True)
