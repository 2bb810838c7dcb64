"""ctypes binding of libcozo_gpu.so (include/cozo_gpu.h).

The product path has no CPU fallback: if the HIP library is missing or no gfx950 device is visible the
calls fail loudly (CozoGpuError / OSError)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COZO_GPU_LIB: an alternative build of the same library (e.g. the phase-timing profiling build of scratch/)
SO_PATH = os.environ.get("COZO_GPU_LIB") or os.path.join(_HERE, "lib", "libcozo_gpu.so")

CZ_NONE = 0xFFFFFFFF
CZ_DEVICE_PTRS = 1
CZ_HNSW_EXTEND_CANDIDATES = 256
CZ_BF_GEMM = 8
CZ_PR_GATHER = 2
CZ_PR_BLOCKED = 4
CZ_PR_ACCUMULATE = 1024
CZ_PR_EXCHANGE_ALLREDUCE = 32
CZ_PR_OVERLAP_EXCHANGE = 64
CZ_PR_ERR_F64_DIFF = 128
CZ_PR_INPLACE_AS_JACOBI = 2048
CZ_ADJ_SYMMETRIC = 512
CZ_UNIQUE_ID_BYTES = 128
CZ_L2, CZ_COSINE, CZ_IP = 0, 1, 2
CZ_OK, CZ_E_INVALID, CZ_E_NO_DEVICE, CZ_E_HIP, CZ_E_CANCELLED, CZ_E_OOM, CZ_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)


class CozoGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcozo_gpu error {code}: {msg}")
        self.code = code


class ProcessKilled(CozoGpuError):
    """CZ_E_CANCELLED: the poison flag was set (runtime/db.rs:1932-1940 `ProcessKilled`)."""


class HnswDesc(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("dim", C.c_uint32),
        ("metric", C.c_int32),
        ("n_levels", C.c_int32),
        ("entry", C.c_uint32),
        ("level_size", u32p),
        ("level_width", i32p),
        ("level_nodes", C.POINTER(u32p)),
        ("level_nbrs", C.POINTER(u32p)),
    ]


class Predicate(C.Structure):
    _fields_ = [("column", C.c_void_p), ("op", C.c_int32), ("const_type", C.c_int32), ("f64_value", C.c_double),
                ("i64_value", C.c_int64)]


CZ_COL_F64, CZ_COL_I64 = 0, 1
CZ_OPS = {"<": 0, "<=": 1, "==": 2, ">=": 3, ">": 4, "!=": 5}


class PagerankTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("plan_build_ms", C.c_double), ("iterate_ms", C.c_double), ("d2h_ms", C.c_double),
                ("cache_hit", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/cozo_gpu.h declares: name -> (restype, argtypes)
# cz_bfs_level_fn (include/cozo_gpu.h): int (*)(void *ctx, uint32_t start, const uint32_t *nodes, uint32_t n)
BFS_LEVEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32)

SYMBOLS = {
    "cz_init": (C.c_int, [C.c_int]),
    "cz_shutdown": (None, []),
    "cz_device_count": (C.c_int, []),
    "cz_last_error": (C.c_char_p, []),
    "cz_version": (C.c_char_p, []),
    "cz_hnsw_index_probe": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_debug_sort_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_debug_seq_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "cz_hbm_probe": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_random_access_probe": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_hnsw_index_create": (C.c_int, [C.POINTER(HnswDesc), C.c_void_p, C.POINTER(C.c_void_p)]),
    "cz_hnsw_index_create_f64": (C.c_int, [C.POINTER(HnswDesc), C.c_void_p, C.POINTER(C.c_void_p)]),
    "cz_hnsw_index_destroy": (None, [C.c_void_p]),
    "cz_hnsw_index_bytes": (C.c_uint64, [C.c_void_p]),
    "cz_hnsw_index_table_contiguous": (C.c_int, [C.c_void_p]),
    "cz_hnsw_index_settle": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "cz_debug_index_table_address": (C.c_uint64, [C.c_void_p]),
    "cz_debug_index_rehome": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cz_hnsw_build": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                C.c_uint64, C.c_uint32, u64p, C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p]),
    "cz_hnsw_insert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64,
                                 C.c_uint32, u64p, C.c_uint32, C.c_void_p]),
    "cz_hnsw_remove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "cz_hnsw_set_key_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "cz_hnsw_index_info": (C.c_int, [C.c_void_p, u32p, u32p, i32p, i32p, u32p]),
    "cz_hnsw_index_level_info": (C.c_int, [C.c_void_p, C.c_int32, u32p, i32p]),
    "cz_hnsw_index_export_level": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "cz_hnsw_index_export_vectors": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_hnsw_index_export_degrees": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "cz_hnsw_set_row_of": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "cz_hnsw_search_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_void_p]),
    "cz_hnsw_search_batch_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_void_p]),
    "cz_column_upload": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]),
    "cz_column_destroy": (None, [C.c_void_p]),
    "cz_hnsw_search_filtered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double,
                                          C.POINTER(Predicate), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_hnsw_search_filtered_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double,
                                          C.POINTER(Predicate), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_hnsw_index_distance_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_distance_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_distance_batch_f64": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_knn_bruteforce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                    C.c_void_p]),
    "cz_pagerank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_double,
                              C.c_uint32, C.c_void_p, u32p, f64p, C.c_void_p]),
    "cz_pagerank_cached": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                     C.c_float, C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, u32p, f64p, C.c_void_p,
                                     C.POINTER(PagerankTiming)]),
    "cz_pagerank_cache_clear": (None, []),
    "cz_pagerank_plan_timing": (C.c_int, [C.c_void_p, f64p, f64p]),
    "cz_pagerank_plan_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_float, C.POINTER(C.c_void_p), C.c_uint32]),
    "cz_pagerank_plan_destroy": (None, [C.c_void_p]),
    "cz_pagerank_plan_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_pagerank_plan_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_pagerank_plan_scores": (C.c_void_p, [C.c_void_p]),
    "cz_pagerank_plan_edges": (C.c_uint64, [C.c_void_p]),
    "cz_pagerank_plan_is_blocked": (C.c_int, [C.c_void_p]),
    "cz_pagerank_plan_formulation": (C.c_int, [C.c_void_p]),
    "cz_pagerank_plan_shape": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_pagerank_plan_read_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_pagerank_plan_nodes": (C.c_uint32, [C.c_void_p]),
    "cz_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cz_comm_create_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cz_comm_destroy": (None, [C.c_void_p]),
    "cz_comm_rank": (C.c_int, [C.c_void_p]),
    "cz_comm_size": (C.c_int, [C.c_void_p]),
    "cz_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cz_comm_all_reduce_sum_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cz_pagerank_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, u32p, f64p,
                                      C.c_void_p, C.c_void_p]),
    "cz_pagerank_sharded_overlapped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32,
                                                 u32p, f64p, C.c_void_p, C.c_void_p]),
    "cz_pagerank_inplace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_uint32,
                                      C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32),
                                      C.c_void_p]),
    "cz_pagerank_inplace_plan_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_uint32,
                                                  C.POINTER(C.c_void_p)]),
    "cz_pagerank_inplace_plan_destroy": (None, [C.c_void_p]),
    "cz_pagerank_inplace_plan_run": (C.c_int, [C.c_void_p, C.c_double, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.c_void_p,
                                               C.c_void_p]),
    "cz_pagerank_inplace_plan_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_pagerank_inplace_plan_sweeps": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_pagerank_inplace_plan_read_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cz_pagerank_inplace_plan_info": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_comm_multi_shutdown": (None, []),
    "cz_pagerank_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_double,
                                    C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, u32p, f64p, C.c_void_p]),
    "cz_bfs_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_sssp_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_connected_components_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, u32p, C.c_void_p]),
    "cz_hnsw_multi_build": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32,
                                      C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cz_hnsw_multi_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cz_hnsw_multi_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_hnsw_multi_shards": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_hnsw_multi_destroy": (None, [C.c_void_p]),
    "cz_hnsw_search_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_connected_components_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                                                  u32p, u32p, C.c_void_p]),
    "cz_bfs_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                                 C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "cz_sssp_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_sssp_sharded_last_stats": (C.c_int, [C.c_void_p]),
    "cz_bfs_shared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "cz_bfs_shared_until": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, BFS_LEVEL_FN, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_bfs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_connected_components": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, u32p, C.c_void_p]),
    "cz_clustering_coefficients": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint32]),
    "cz_sssp_goals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_sssp_goals_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_sssp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                          C.c_void_p, C.c_void_p]),
    "cz_graph_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cz_graph_destroy": (None, [C.c_void_p]),
    "cz_graph_acquire": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int)]),
    "cz_graph_release": (None, [C.c_uint64, C.c_uint64, C.c_void_p]),
    "cz_graph_cache_clear": (None, []),
    "cz_bfs_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p]),
    "cz_connected_components_on": (C.c_int, [C.c_void_p, C.c_void_p, u32p, C.c_void_p]),
    "cz_sssp_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cz_graph_last_timing": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_label_propagation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, u32p, u32p,
                                       C.c_void_p, C.c_uint32]),
    "cz_closeness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cz_betweenness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
}

_lib = None
_hip_runtime_path = None


def _preload_hip_runtime() -> str:
    """libcozo_gpu.so carries no NEEDED entry for the HIP runtime: a process must hold exactly ONE
    libamdhip64 / libhsa-runtime64 pair, and PyTorch wheels bundle their own.  Bring one in (RTLD_GLOBAL)
    before the library's static constructors register its code objects.
    COZO_HIP_RUNTIME = auto (default: torch's bundled runtime if torch is installed, else /opt/rocm) |
    torch | system | /path/to/libamdhip64.so"""
    global _hip_runtime_path
    if _hip_runtime_path:
        return _hip_runtime_path
    choice = os.environ.get("COZO_HIP_RUNTIME", "auto")
    cands = []
    if choice in ("auto", "torch"):
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec and spec.origin:
                cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
        except Exception:
            pass
    if choice in ("auto", "system"):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cands += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so"),
                  "libamdhip64.so.7", "libamdhip64.so"]
    if choice not in ("auto", "torch", "system"):
        cands = [choice]
    errs = []
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            _hip_runtime_path = c
            # the collectives library must come from the same distribution as the runtime (a PyTorch wheel bundles its own
            # librccl.so next to its libamdhip64.so; mixing it with /opt/rocm's drags a second set of ROCm support
            # libraries into the process -- observed: "double free or corruption" at interpreter exit)
            rccl = os.path.join(os.path.dirname(c), "librccl.so")
            if os.path.isabs(c) and os.path.exists(rccl):
                os.environ.setdefault("COZO_RCCL_LIB", rccl)
            return c
        except OSError as e:  # pragma: no cover
            errs.append(f"{c}: {e}")
    raise OSError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errs))


def lib() -> C.CDLL:
    """Load the library (no device needed to load; compute calls need a gfx950 GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise OSError(f"{SO_PATH} is missing: build it with `python -m cozo_amd.build` "
                          "(there is no CPU fallback for the cozo_amd product path)")
        _preload_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc == CZ_OK:
        return
    msg = lib().cz_last_error().decode("utf-8", "replace")
    if rc == CZ_E_CANCELLED:
        raise ProcessKilled(rc, msg)
    raise CozoGpuError(rc, msg)


def ptr(a):
    """void* of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)
