"""Shared builders for the parity tests (synthetic inputs, seeded)."""
import numpy as np


def vectors(n, d, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":  # the reference's rand_vec distribution (data/functions.rs:2149-2156)
        return rng.random((n, d), dtype=np.float32)
    if kind == "normal":
        return rng.standard_normal((n, d), dtype=np.float32)
    if kind == "lowrank":  # embedding-like: rank-r latent + small isotropic noise
        r = 16
        z = rng.standard_normal((n, r), dtype=np.float32)
        w = np.random.default_rng(12345).standard_normal((r, d), dtype=np.float32)
        return (z @ w + 0.1 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    raise ValueError(kind)


def build_index(O, x, metric, m, ef_c, seed=7, **kw):
    b = O.HnswBuilder(x.shape[1], metric, m, ef_c, **kw)
    b.insert(x, O.random_levels(x.shape[0], m, seed))
    return b, b.export()


def gpu_index(flat, distance, m):
    from cozo_amd.hnsw import GpuHnswIndex, HnswIndexManifest
    man = HnswIndexManifest(vec_dim=flat.dim, distance=distance, m_neighbours=m)
    return GpuHnswIndex(man, flat.vectors, [None] + flat.level_nodes[1:], flat.level_nbrs, flat.entry)


def random_relation(n, e, seed, self_loops=False):
    """A relation of (from, to) int rows: a sorted set, like a stored relation scanned in key order."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = rng.integers(0, n, e)
    if not self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    rows = np.unique(np.stack([src, dst], 1), axis=0)
    return rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)


def graph_from_relation(O, frm, to, undirected=False, weights=None):
    fi, ti, ind = O.assign_ids(frm, to)
    n = len(ind)
    if weights is None:
        ooff, otgt = O.build_csr(n, fi, ti, undirected=undirected)
        ioff, isrc = O.build_csr(n, ti, fi, undirected=undirected)
        return dict(n=n, fi=fi, ti=ti, ind=ind, ooff=ooff, otgt=otgt, ioff=ioff, isrc=isrc,
                    outdeg=np.diff(ooff).astype(np.uint32))
    ooff, otgt, ow = O.build_csr(n, fi, ti, weights=weights, undirected=undirected)
    return dict(n=n, fi=fi, ti=ti, ind=ind, ooff=ooff, otgt=otgt, ow=ow)
