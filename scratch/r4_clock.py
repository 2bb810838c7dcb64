"""scratch (round 4): how much of the box-to-box spread of the random-row kernels is the shader clock the SMU's DPM settles on?
One 1M x 768 index, batch 1024, ef 96 + 4M-pair cz_distance_batch; measured with power_dpm_force_performance_level = auto | high | low
(written through sysfs as root on the GPU box; restored to auto afterwards), the clocks sampled during every timed loop."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import torch
import bench as Bn
import boxstate
from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest, HnswSearch, distance_batch_device

def set_level(sysd, level):
    try:
        with open(os.path.join(sysd, "power_dpm_force_performance_level"), "w") as f:
            f.write(level)
        return open(os.path.join(sysd, "power_dpm_force_performance_level")).read().strip()
    except OSError as e:
        return f"failed: {e}"

def main():
    dev = torch.device("cuda:0")
    assert L.cz_init(0) == 0
    n, dim, k, B, ef = 1_000_000, 768, 10, 1024, 96
    stream = torch.cuda.current_stream().cuda_stream
    sysd = boxstate.device_sysfs(torch)
    print(json.dumps({k_: v for k_, v in boxstate.static_state(torch).items() if k_ in ("pci", "unique_id", "perf_level", "node_gpus_busy", "node_power_w")}), flush=True)
    x = Bn.gen_vectors(torch, n, dim, "lowrank", 42, dev)
    q = Bn.gen_vectors(torch, B, dim, "lowrank", 43, dev)
    man = HnswIndexManifest(vec_dim=dim, distance="Cosine", m_neighbours=32, ef_construction=200)
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    ids = torch.empty((B, k), dtype=torch.int32, device=dev)
    dd = torch.empty((B, k), dtype=torch.float64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    nd = torch.zeros(B, dtype=torch.int64, device=dev)
    P = 1 << 22
    g = torch.Generator(device=dev); g.manual_seed(1)
    pairs = torch.stack([torch.randint(0, B, (P,), generator=g, device=dev, dtype=torch.int32),
                         torch.randint(0, n, (P,), generator=g, device=dev, dtype=torch.int32)], 1).contiguous()
    outd = torch.empty(P, dtype=torch.float64, device=dev)
    def timed(fn, reps):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with boxstate.Sampler(sysd, 0.002) as smp:
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
        sm = smp.summary()
        return e0.elapsed_time(e1) / reps, sm.get("sclk_mhz", {}), sm.get("power_w", {})
    for level in ("auto", "high", "auto", "low", "auto", "high"):
        got = set_level(sysd, level)
        time.sleep(0.3)
        ms, sc, pw = timed(lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream), 40)
        tot = float(nd.sum().item())
        ms2, sc2, pw2 = timed(lambda: distance_batch_device("Cosine", x, q, pairs, outd, stream), 40)
        print(f"level {level:5s} (reads {got}): hnsw {ms:.3f} ms frac {tot * 4 * dim / ms / 1e6 / 8000:.3f} sclk {sc.get('min', 0):.0f}/{sc.get('mean', 0):.0f}/{sc.get('max', 0):.0f} power {pw.get('mean', 0):.0f}W | "
              f"distance {ms2:.3f} ms frac {P * 4 * dim / ms2 / 1e6 / 8000:.3f} sclk {sc2.get('min', 0):.0f}/{sc2.get('mean', 0):.0f}/{sc2.get('max', 0):.0f} power {pw2.get('mean', 0):.0f}W", flush=True)
    print("restored:", set_level(sysd, "auto"))
    ix.close()
main()
