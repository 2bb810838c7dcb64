"""scratch (round 4): is hnsw_knn_kernel's launch a single wave of workgroups whose time is set by the slowest query?
124 VGPRs -> 4 workgroups per CU -> 1024 slots = the whole batch of 1024 resident at once.  Measured here on one index:
per-query n_dist distribution, and throughput against the batch size per launch (a larger batch back-fills freed slots)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
import boxstate
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k = int(os.environ.get("HS_N", 1_000_000)), 768, 10
    ef = int(os.environ.get("HS_EF", 96))
    kind = os.environ.get("HS_DIST", "lowrank")
    stream = torch.cuda.current_stream().cuda_stream
    print(json.dumps(boxstate.static_state(torch)), flush=True)
    x = Bn.gen_vectors(torch, n, dim, kind, 42, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"build {time.time() - t0:.1f}s", flush=True)
    del x
    torch.cuda.empty_cache()
    BMAX = 8192
    qall = Bn.gen_vectors(torch, BMAX, dim, kind, 43, dev)
    sysd = boxstate.device_sysfs(torch)
    for B in (256, 512, 768, 1024, 1280, 2048, 4096, 8192):
        q = qall[:B].contiguous()
        ids = torch.empty((B, k), dtype=torch.int32, device=dev)
        dd = torch.empty((B, k), dtype=torch.float64, device=dev)
        cnt = torch.empty(B, dtype=torch.int32, device=dev)
        nd = torch.zeros(B, dtype=torch.int64, device=dev)
        def run():
            ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(3): run()
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with boxstate.Sampler(sysd) as smp:
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ndf = nd.to(torch.float64)
        tot = float(ndf.sum().item())
        qs = torch.quantile(ndf, torch.tensor([0.0, 0.05, 0.5, 0.95, 0.99, 1.0], dtype=torch.float64, device=dev)).tolist()
        sm = smp.summary()
        print(f"B={B:5d} ef={ef}: {ms:.3f} ms  {B / ms * 1e3:8.0f} q/s  {tot * 4 * dim / ms / 1e6 / 8000:.3f} of peak   n_dist/q mean {tot / B:.0f} "
              f"min/p5/p50/p95/p99/max {' '.join(f'{v:.0f}' for v in qs)}  max/mean {qs[-1] / (tot / B):.3f}  "
              f"sclk {sm.get('sclk_mhz', {}).get('mean', 0):.0f} power {sm.get('power_w', {}).get('mean', 0):.0f}W samples {sm['samples']}", flush=True)
    ix.close()
main()
