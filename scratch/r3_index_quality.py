"""Index quality on BASELINE.md's 16-cluster corpus: recall@10 against ef for the four settings of
keep_pruned_connections / extend_candidates (hnsw.rs:499-511, 524-536), all built on the device (batched).
    python scratch/r3_index_quality.py [n] [dim] [m] [ef_construction] [kind]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
m = int(sys.argv[3]) if len(sys.argv) > 3 else 32
efc = int(sys.argv[4]) if len(sys.argv) > 4 else 200
kind = sys.argv[5] if len(sys.argv) > 5 else "clustered"
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
x = bench.gen_vectors(torch, n, dim, kind, 42, dev)
q = bench.gen_vectors(torch, 512, dim, kind, 43, dev)
B, k = q.shape[0], 10
print(f"{kind}: {n} x {dim}, m = {m}, ef_construction = {efc}, {B} queries, recall@{k}", flush=True)
gt = None
for keep, ext in ((False, False), (True, False), (False, True), (True, True)):
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=m, ef_construction=efc, keep_pruned_connections=keep,
                            extend_candidates=ext)
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if gt is None:
        g32 = torch.empty((B, k), dtype=torch.int32, device=dev)
        gd = torch.empty((B, k), dtype=torch.float64, device=dev)
        ix.bruteforce_knn_device(q, k, g32, gd, stream, gemm=True)
        torch.cuda.synchronize()
        gt = g32.to(torch.int64) & 0xFFFFFFFF
    ids = torch.empty((B, k), dtype=torch.int32, device=dev)
    dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    nd = torch.zeros(B, dtype=torch.int64, device=dev)
    deg0 = ix.degrees()[0]
    nbr0 = ix.export()[1][0]
    live = float((nbr0 != 0xFFFFFFFF).sum(axis=1).mean())
    line = []
    for ef in (64, 128, 256, 512, 768):
        ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        torch.cuda.synchronize()
        rec = bench.recall_at_k(torch, ids.to(torch.int64) & 0xFFFFFFFF, gt)
        line.append(f"ef {ef}: {rec:.4f} ({int(nd.sum().item()) // B} evals)")
    print(f"  keep_pruned={int(keep)} extend={int(ext)}: build {dt:6.1f} s, {ix.last_build_n_dist / n:7.0f} evals/vector, "
          f"{live:5.1f} links/node on level 0 (degree {float(deg0.mean()):5.1f});  " + ";  ".join(line), flush=True)
    ix.close()
