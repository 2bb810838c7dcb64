"""scratch: PageRank on a STORED relation end to end, stage by stage (SURVEY section 8d asks for the end-to-end figure next to
the kernel loop): stored key bytes of an (int, int)-keyed edge relation -> libcozo_ingest (ids + in-CSR + out-degrees) ->
cz_pagerank on host arrays (upload, plan build, the reference's default run, scores back) -> N node-value decodes.
   python scratch/e2e_pagerank_stored.py [--rows 100000000] [--nodes 10000000] [--no-gpu]
Run on the GPU box for the device stage; --no-gpu stops after the ingest (what a CPU-only box can do)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cozo_amd import build as B, codec  # noqa: E402
from cozo_amd.ingest import StoredGraph  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=100_000_000)
p.add_argument("--nodes", type=int, default=10_000_000)
p.add_argument("--no-gpu", action="store_true")
a = p.parse_args()
B.build_ingest()
t0 = time.perf_counter()
rng = np.random.default_rng(42)
key = rng.integers(0, a.nodes, a.rows, dtype=np.int64) * a.nodes + rng.integers(0, a.nodes, a.rows, dtype=np.int64)
key = np.unique(key)  # a relation is a sorted set of (from, to)
frm, to = key // a.nodes, key % a.nodes
keep = frm != to
frm, to = frm[keep], to[keep]
e = frm.size
rec = np.zeros((e, 28), dtype=np.uint8)
rec[:, 7] = 1
for c, col in enumerate((frm, to)):  # memcmp encoding of a non-negative int (data/memcmp.rs:127-145)
    rec[:, 8 + 10 * c] = 0x05
    rec[:, 9 + 10 * c:17 + 10 * c] = (col.astype(np.float64).view(np.uint64) | np.uint64(0x8000000000000000)).byteswap().view(np.uint8).reshape(e, 8)
rows = codec.StoredRows(rec.tobytes(), np.arange(e + 1, dtype=np.uint64) * 28, b"", np.zeros(e + 1, dtype=np.uint64), 2)
del rec, key
print(f"relation: {e} rows over {a.nodes} node values, {len(rows.keys) / 1e9:.2f} GB of key bytes (made in {time.perf_counter() - t0:.1f} s)", flush=True)

t0 = time.perf_counter()
g = StoredGraph(rows)
t_ids = time.perf_counter() - t0
t0 = time.perf_counter()
in_off, in_src, _ = g.csr(True)
out_off, _, _ = g.csr(False)
out_deg = np.diff(out_off).astype(np.uint32)
t_csr = time.perf_counter() - t0
print(f"ingest: ids {t_ids:.2f} s, in-CSR + out-degrees {t_csr:.2f} s ({e / (t_ids + t_csr) / 1e6:.1f} M rows/s, "
      f"CZI_THREADS={os.environ.get('CZI_THREADS', 'default')}, {os.cpu_count()} cores); {g.n} nodes", flush=True)
if a.no_gpu:
    sys.exit(0)
from cozo_amd import _lib, graph as G  # noqa: E402
if _lib.lib().cz_init(0) != 0:
    raise SystemExit(_lib.lib().cz_last_error().decode())
best = None
for _ in range(2):  # the first call warms allocations
    t0 = time.perf_counter()
    scores, it, err = G.pagerank(in_off, in_src, out_deg, 0.85, 1e-4, 10)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(f"cz_pagerank on host arrays (upload + plan + {it} iterations + scores back): {best * 1e3:.1f} ms", flush=True)
t0 = time.perf_counter()
data, off = g.node_keys()
first = [codec.decode_datavalue(data, int(off[i]))[0] for i in range(min(g.n, 100_000))]
t_dec = (time.perf_counter() - t0) * g.n / max(1, len(first))
print(f"node values back (Python decode, extrapolated from {len(first)}): {t_dec:.1f} s; "
      f"end to end without it: {t_ids + t_csr + best:.2f} s for {e} edges x {it} iterations", flush=True)
