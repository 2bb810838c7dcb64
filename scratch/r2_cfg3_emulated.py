"""scratch (round 2): BASELINE.json configs[3] -- 10M x 768 split into 8 sub-indices of 1.25M, per-shard hnsw_knn + top-k merge --
emulated on ONE GPU, shard after shard: merged recall@10 against the exact neighbours over all 10M (per-shard exact lists merged),
per-shard step time (what each of 8 GPUs would spend) and their sum (what one GPU spends doing all 8)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn

dev = torch.device("cuda:0")
assert L.cz_init(0) == 0
args = Bn.parse()
N, S, B, k = int(os.environ.get("CFG3_N", 10_000_000)), 8, 1024, 10
per = (N + S - 1) // S
q = Bn.gen_vectors(torch, B, 768, "lowrank", 43, dev)
x = Bn.gen_vectors(torch, N, 768, "lowrank", 42, dev)
efs = [64, 96, 128, 144, 192]
exact, found, times, build = [], {ef: [] for ef in efs}, {ef: [] for ef in efs}, []
for s in range(S):
    lo, hi = s * per, min(N, (s + 1) * per)
    run = Bn.HnswRun(args, torch, dev, hi - lo, "lowrank", q, x=x[lo:hi])
    build.append(run.build_s)
    gt = torch.empty((B, k), dtype=torch.int32, device=dev)
    gtd = torch.empty((B, k), dtype=torch.float64, device=dev)
    run.ix.bruteforce_knn_device(q, k, gt, gtd, run.stream, gemm=True)
    exact.append(((gt.to(torch.int64) & 0xFFFFFFFF) + lo, gtd.clone()))
    for ef in efs:
        run.search(ef); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run.search(ef)
        e1.record(); torch.cuda.synchronize()
        times[ef].append(e0.elapsed_time(e1) / 10)
        found[ef].append(((run.ids.to(torch.int64) & 0xFFFFFFFF) + lo, run.dd.clone()))
    run.x = None
    run.close()
    print(f"shard {s}: built in {build[-1]:.1f}s; ms/batch {[round(times[ef][-1], 3) for ef in efs]}", flush=True)

def merge(lists):  # smallest distance first, ties by id: merge_shard_topk's order
    ids = torch.cat([a for a, _ in lists], 1)
    dd = torch.cat([b for _, b in lists], 1)
    key = torch.argsort(ids, dim=1, stable=True)
    ids, dd = torch.gather(ids, 1, key), torch.gather(dd, 1, key)
    order = torch.argsort(dd, dim=1, stable=True)[:, :k]
    return torch.gather(ids, 1, order)

gt_all = merge(exact)
out = dict(workload=f"{N} x 768 (lowrank) as {S} sub-indices of {per}, batch {B}, k = {k}; one GPU, shard after shard", shard_build_s=build, by_ef=[])
for ef in efs:
    rec = Bn.recall_at_k(torch, merge(found[ef]), gt_all)
    out["by_ef"].append(dict(ef=ef, merged_recall_at_10=rec, ms_per_batch_per_shard=times[ef], slowest_shard_ms=max(times[ef]),
                             all_shards_on_one_gpu_ms=sum(times[ef]), queries_per_s_8_gpus_ideal=B / max(times[ef]) * 1e3,
                             queries_per_s_one_gpu=B / sum(times[ef]) * 1e3))
    print(json.dumps(out["by_ef"][-1]), flush=True)
print(json.dumps(out))
