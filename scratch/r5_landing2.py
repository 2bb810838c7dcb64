"""scratch (round 5): the same 10M index created again and again in ONE process, other allocations in the way or not -- does the search
speed follow where the (contiguous) vector table lands?  Prints, per creation: the table's device address and its alignment, whether
it is contiguous, the search fraction (2 x 20 launches), the mean shader clock during the launches.
HS_N (default 10M) x 768, ef 144, batch 1024."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cozo_amd import _lib
L = _lib.lib()
import numpy as np
import torch
import bench as Bn
import boxstate
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
    sysfs = boxstate.device_sysfs(torch, 0)

    def timed(ix, tag):
        run = lambda: ix.hnsw_knn_batch_device(q, HnswSearch(k=k, ef=ef), ids, dd, cnt, nd, stream)
        for _ in range(5): run()
        torch.cuda.synchronize()
        out = []
        with boxstate.Sampler(sysfs) as smp:
            for rep in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run()
                e1.record(); torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) / 20)
        clk = smp.summary().get("sclk_mhz", {})
        tot = int(nd.sum().item())
        va = int(L.cz_debug_index_table_address(ix._h))
        print(f"{tag:44s} va 0x{va:012x} mod2M {va % (2 << 20):8d} mod1G {(va % (1 << 30)) >> 20:5d}M  contiguous {int(ix.table_contiguous)}  "
              f"{' '.join(f'{m:.3f}' for m in out)} ms  {tot * 4 * dim / min(out) / 1e6 / 8000:.3f} of peak  sclk {clk.get('mean', 0):.0f} "
              f"({clk.get('min', 0):.0f}-{clk.get('max', 0):.0f})", flush=True)

    t0 = time.time()
    ix = GpuHnswIndex.build(man, x, seed=7, max_batch=4096, device_ptr=True, n=n, stream=stream)
    torch.cuda.synchronize()
    print(f"built {n} in {time.time() - t0:.1f}s", flush=True)
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    timed(ix, "built")
    timed(ix, "built (again)")
    nodes, nbrs, entry = ix.export()
    ix.close()
    torch.cuda.empty_cache()
    MB = 1 << 20
    # (junk in the way while the index is created, CZ_AUX_CONTIGUOUS): the link tables / visited workspaces paged or contiguous
    plan = [(0, "0"), (0, "1"), (3 * MB + 4096, "0"), (3 * MB + 4096, "1"), (0, "0"), (0, "1"), (5 * 1024 * MB + MB + 4096, "0"),
            (5 * 1024 * MB + MB + 4096, "1"), (0, "0"), (0, "1"), (37 * 1024 * MB + 12288, "0"), (37 * 1024 * MB + 12288, "1"), (0, "0"), (0, "1")]
    for i, (js, aux) in enumerate(plan):
        os.environ["CZ_AUX_CONTIGUOUS"] = aux
        junk = torch.empty(js, dtype=torch.uint8, device=dev) if js else None
        ix2 = GpuHnswIndex(man, xh, nodes, nbrs, entry)
        del junk
        torch.cuda.empty_cache()
        timed(ix2, f"created #{i} aux={aux} junk {js / MB:.0f} MB")
        ix2.close()
        torch.cuda.empty_cache()


main()
