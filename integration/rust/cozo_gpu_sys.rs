//! FFI declarations for libcozo_gpu.so (include/cozo_gpu.h) and libcozo_ingest.so (include/cozo_ingest.h).
//! Not compiled in this repository's image (no rustc); kept in step with the headers by tests/test_rust_shim.py.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

pub const CZ_NONE: u32 = 0xFFFF_FFFF;
pub const CZ_DEVICE_PTRS: u32 = 1;
pub const CZ_HNSW_EXTEND_CANDIDATES: u32 = 256;
pub const CZ_PR_GATHER: u32 = 2;
pub const CZ_PR_BLOCKED: u32 = 4;
pub const CZ_PR_ACCUMULATE: u32 = 1024;
pub const CZ_BF_GEMM: u32 = 8;
pub const CZ_PR_EXCHANGE_ALLREDUCE: u32 = 32;
pub const CZ_PR_OVERLAP_EXCHANGE: u32 = 64;
pub const CZ_PR_ERR_F64_DIFF: u32 = 128;
pub const CZ_PR_INPLACE_AS_JACOBI: u32 = 2048;
pub const CZ_ADJ_SYMMETRIC: u32 = 512;
pub const CZ_UNIQUE_ID_BYTES: u32 = 128;

pub const CZ_OK: c_int = 0;
pub const CZ_E_INVALID: c_int = -1;
pub const CZ_E_NO_DEVICE: c_int = -2;
pub const CZ_E_HIP: c_int = -3;
pub const CZ_E_CANCELLED: c_int = -4; // -> ProcessKilled (runtime/db.rs:1932-1940)
pub const CZ_E_OOM: c_int = -5;
pub const CZ_E_UNSUPPORTED: c_int = -6;

pub const CZ_L2: c_int = 0;
pub const CZ_COSINE: c_int = 1;
pub const CZ_IP: c_int = 2;

#[repr(C)]
pub struct cz_hnsw_index {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cz_pagerank_plan {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cz_pagerank_inplace_plan {
    _private: [u8; 0],
}

#[repr(C)]
pub struct cz_column {
    _private: [u8; 0],
}

pub const CZ_COL_F64: c_int = 0;
pub const CZ_COL_I64: c_int = 1;
pub const CZ_OP_LT: c_int = 0;
pub const CZ_OP_LE: c_int = 1;
pub const CZ_OP_EQ: c_int = 2;
pub const CZ_OP_GE: c_int = 3;
pub const CZ_OP_GT: c_int = 4;
pub const CZ_OP_NE: c_int = 5;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct cz_predicate {
    pub column: *const cz_column,
    pub op: i32,
    pub const_type: i32,
    pub f64_value: c_double,
    pub i64_value: i64,
}

#[repr(C)]
pub struct cz_hnsw_multi {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cz_comm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cz_graph {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct cz_pagerank_timing {
    pub h2d_ms: c_double,
    pub plan_build_ms: c_double,
    pub iterate_ms: c_double,
    pub d2h_ms: c_double,
    pub cache_hit: i32,
    pub reserved: i32,
}

#[repr(C)]
pub struct cz_hnsw_desc {
    pub n: u32,
    pub dim: u32,
    pub metric: i32,
    pub n_levels: i32,
    pub entry: u32,
    pub level_size: *const u32,
    pub level_width: *const i32,
    pub level_nodes: *const *const u32,
    pub level_nbrs: *const *const u32,
}

#[link(name = "cozo_gpu")]
#[link(name = "amdhip64")] // the host binary brings the one HIP runtime of the process
/// cozo_gpu.h `cz_bfs_level_fn`: the nodes a BFS level discovered; 0 = go on, > 0 = enough, < 0 = failed
#[allow(non_camel_case_types)]
pub type cz_bfs_level_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, start: u32, nodes: *const u32, n: u32) -> c_int>;

extern "C" {
    pub fn cz_init(device: c_int) -> c_int;
    pub fn cz_shutdown();
    pub fn cz_device_count() -> c_int;
    pub fn cz_last_error() -> *const c_char;
    pub fn cz_version() -> *const c_char;
    pub fn cz_hbm_probe(table: *const c_void, rows: u64, row_bytes: u32, n_fetch: u64, reps: u32, stream_gbs: *mut c_double,
                        row_fetch_gbs: *mut c_double) -> c_int;
    pub fn cz_random_access_probe(n_words: u64, word_bytes: u32, n_access: u64, reps: u32, loads_g_per_s: *mut c_double,
                                  atomic_min_g_per_s: *mut c_double) -> c_int;
    pub fn cz_debug_sort_pairs(keys: *const u32, vals: *const u32, n: u64, bits: u32, out_keys: *mut u32, out_vals: *mut u32,
                               out_scan: *mut u32) -> c_int;
    pub fn cz_debug_seq_sum(terms: *const c_float, row_off: *const u64, init: *const c_float, n_rows: u32, lanes: c_int, per_lane: c_int,
                            out: *mut c_float) -> c_int;

    pub fn cz_hnsw_index_create(desc: *const cz_hnsw_desc, vectors: *const c_float, out: *mut *mut cz_hnsw_index) -> c_int;
    pub fn cz_hnsw_index_create_f64(desc: *const cz_hnsw_desc, vectors: *const c_double, out: *mut *mut cz_hnsw_index) -> c_int;
    pub fn cz_hnsw_index_destroy(ix: *mut cz_hnsw_index);
    pub fn cz_hnsw_index_bytes(ix: *const cz_hnsw_index) -> u64;
    pub fn cz_hnsw_index_table_contiguous(ix: *const cz_hnsw_index) -> c_int;
    pub fn cz_hnsw_index_settle(ix: *mut cz_hnsw_index, ef: u32, trials: u32, ms_before: *mut c_double, ms_after: *mut c_double,
                                n_tried: *mut u32) -> c_int;
    pub fn cz_debug_index_table_address(ix: *const cz_hnsw_index) -> u64;
    pub fn cz_debug_index_rehome(ix: *mut cz_hnsw_index, what: c_int, contiguous: c_int) -> c_int;
    pub fn cz_hnsw_index_probe(ix: *const cz_hnsw_index, n_fetch: u64, reps: u32, stream_gbs: *mut c_double, row_fetch_gbs: *mut c_double) -> c_int;
    pub fn cz_hnsw_build(vectors: *const c_float, n: u32, dim: u32, metric: c_int, m: u32, ef_construction: u32,
                         keep_pruned_connections: c_int, levels: *const i32, seed: u64, max_batch: u32, n_dist: *mut u64,
                         out: *mut *mut cz_hnsw_index, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_hnsw_index_info(ix: *const cz_hnsw_index, n: *mut u32, dim: *mut u32, metric: *mut i32, n_levels: *mut i32,
                              entry: *mut u32) -> c_int;
    pub fn cz_hnsw_index_level_info(ix: *const cz_hnsw_index, level: i32, size: *mut u32, width: *mut i32) -> c_int;
    pub fn cz_hnsw_index_export_level(ix: *const cz_hnsw_index, level: i32, node_ids: *mut u32, nbrs: *mut u32) -> c_int;
    pub fn cz_hnsw_index_export_vectors(ix: *const cz_hnsw_index, out: *mut c_float) -> c_int;
    pub fn cz_hnsw_index_export_degrees(ix: *const cz_hnsw_index, level: i32, degree: *mut c_double) -> c_int;
    pub fn cz_hnsw_set_row_of(ix: *mut cz_hnsw_index, row_of: *const u32, n: u32) -> c_int;
    pub fn cz_hnsw_search_batch(ix: *mut cz_hnsw_index, queries: *const c_float, b: u32, k: u32, ef: u32, has_radius: c_int,
                                radius: c_double, out_ids: *mut u32, out_dist: *mut c_double, out_count: *mut u32,
                                out_n_dist: *mut u64, poison: *const u8, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_hnsw_search_batch_f64(ix: *mut cz_hnsw_index, queries: *const c_double, b: u32, k: u32, ef: u32, has_radius: c_int,
                                    radius: c_double, out_ids: *mut u32, out_dist: *mut c_double, out_count: *mut u32,
                                    out_n_dist: *mut u64, poison: *const u8, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_distance_batch_f64(metric: c_int, base: *const c_double, n: u32, dim: u32, queries: *const c_double, nq: u32,
                                 pairs: *const u32, p: u64, out: *mut c_double, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_distance_batch(metric: c_int, base: *const c_float, n: u32, dim: u32, queries: *const c_float, nq: u32,
                             pairs: *const u32, p: u64, out: *mut c_double, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_hnsw_index_distance_batch(ix: *mut cz_hnsw_index, queries: *const c_float, nq: u32, pairs: *const u32, p: u64,
                                        out: *mut c_double, flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_knn_bruteforce(ix: *mut cz_hnsw_index, queries: *const c_float, b: u32, k: u32, out_ids: *mut u32,
                             out_dist: *mut c_double, flags: u32, stream: *mut c_void) -> c_int;

    pub fn cz_pagerank(in_offsets: *const u32, in_sources: *const u32, out_degree: *const u32, n: u32, e: u64,
                       damping: c_float, tolerance: c_double, max_iter: u32, scores: *mut c_float, iters_run: *mut u32,
                       final_err: *mut c_double, poison: *const u8) -> c_int;
    pub fn cz_pagerank_inplace(in_offsets: *const u32, in_sources: *const u32, out_degree: *const u32, n: u32, e: u64, damping: c_float,
                               tolerance: c_double, max_iter: u32, flags: u32, scores: *mut c_float, iters_run: *mut u32,
                               final_err: *mut c_double, n_levels: *mut u32, poison: *const u8) -> c_int;
    // the resident form of the in-place reading (round 6): layout kept in HBM, one hipGraph per sweep parity
    pub fn cz_pagerank_inplace_plan_create(in_offsets: *const u32, in_sources: *const u32, out_degree: *const u32, n: u32, e: u64,
                                           damping: c_float, flags: u32, out: *mut *mut cz_pagerank_inplace_plan) -> c_int;
    pub fn cz_pagerank_inplace_plan_destroy(p: *mut cz_pagerank_inplace_plan);
    pub fn cz_pagerank_inplace_plan_run(p: *mut cz_pagerank_inplace_plan, tolerance: c_double, max_iter: u32, iters_run: *mut u32,
                                        final_err: *mut c_double, poison: *const u8, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_inplace_plan_init(p: *mut cz_pagerank_inplace_plan, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_inplace_plan_sweeps(p: *mut cz_pagerank_inplace_plan, n: u32, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_inplace_plan_read_scores(p: *mut cz_pagerank_inplace_plan, scores: *mut c_float, flags: u32,
                                                stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_inplace_plan_info(p: *const cz_pagerank_inplace_plan, shape: *mut u64, build_ms: *mut c_double,
                                         h2d_ms: *mut c_double) -> c_int;
    pub fn cz_pagerank_cached(key_hi: u64, key_lo: u64, in_offsets: *const u32, in_sources: *const u32,
                              out_degree: *const u32, n: u32, e: u64, damping: c_float, tolerance: c_double, max_iter: u32,
                              flags: u32, scores: *mut c_float, iters_run: *mut u32, final_err: *mut c_double,
                              poison: *const u8, timing: *mut cz_pagerank_timing) -> c_int;
    pub fn cz_pagerank_cache_clear();
    pub fn cz_pagerank_plan_timing(p: *const cz_pagerank_plan, h2d_ms: *mut c_double, build_ms: *mut c_double) -> c_int;
    pub fn cz_pagerank_plan_create(in_offsets: *const u32, in_sources: *const u32, out_degree: *const u32, n: u32,
                                   row_begin: u32, row_end: u32, damping: c_float, out: *mut *mut cz_pagerank_plan,
                                   flags: u32) -> c_int;
    pub fn cz_pagerank_plan_destroy(p: *mut cz_pagerank_plan);
    pub fn cz_pagerank_plan_init(p: *mut cz_pagerank_plan, contrib_dev: *mut c_float, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_plan_step(p: *mut cz_pagerank_plan, contrib_in_dev: *const c_float, contrib_out_dev: *mut c_float,
                                 err_out_dev: *mut c_double, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_plan_scores(p: *mut cz_pagerank_plan) -> *mut c_float;
    pub fn cz_pagerank_plan_edges(p: *const cz_pagerank_plan) -> u64;
    pub fn cz_pagerank_plan_is_blocked(p: *const cz_pagerank_plan) -> c_int;
    pub fn cz_pagerank_plan_formulation(p: *const cz_pagerank_plan) -> c_int;
    pub fn cz_pagerank_plan_shape(p: *const cz_pagerank_plan, out12: *mut u32) -> c_int;
    pub fn cz_pagerank_plan_read_scores(p: *mut cz_pagerank_plan, out: *mut c_float, flags: u32, stream: *mut c_void) -> c_int;

    pub fn cz_hnsw_insert(ix: *mut cz_hnsw_index, vectors: *const c_float, n_new: u32, m: u32, ef_construction: u32,
                          keep_pruned_connections: c_int, levels: *const i32, seed: u64, max_batch: u32, n_dist: *mut u64,
                          flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_hnsw_remove(ix: *mut cz_hnsw_index, nodes: *const u32, n_nodes: u32) -> c_int;
    pub fn cz_hnsw_set_key_order(ix: *mut cz_hnsw_index, rank: *const u32, n: u32) -> c_int;
    pub fn cz_column_upload(values: *const c_void, n: u32, ty: i32, out: *mut *mut cz_column) -> c_int;
    pub fn cz_column_destroy(c: *mut cz_column);
    pub fn cz_hnsw_search_filtered(ix: *mut cz_hnsw_index, queries: *const c_float, b: u32, k: u32, ef: u32, has_radius: c_int,
                                   radius: c_double, preds: *const cz_predicate, n_preds: u32, out_ids: *mut u32,
                                   out_dist: *mut c_double, out_count: *mut u32, out_n_dist: *mut u64, poison: *const u8,
                                   flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_hnsw_search_filtered_f64(ix: *mut cz_hnsw_index, queries: *const c_double, b: u32, k: u32, ef: u32, has_radius: c_int,
                                       radius: c_double, preds: *const cz_predicate, n_preds: u32, out_ids: *mut u32,
                                       out_dist: *mut c_double, out_count: *mut u32, out_n_dist: *mut u64, poison: *const u8,
                                       flags: u32, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_plan_nodes(p: *const cz_pagerank_plan) -> u32;

    // multi-GPU, one node (RCCL over xGMI)
    pub fn cz_comm_unique_id(id: *mut u8) -> c_int;
    pub fn cz_comm_create_rank(id: *const u8, rank: c_int, world: c_int, out: *mut *mut cz_comm) -> c_int;
    pub fn cz_comm_destroy(c: *mut cz_comm);
    pub fn cz_comm_rank(c: *const cz_comm) -> c_int;
    pub fn cz_comm_size(c: *const cz_comm) -> c_int;
    pub fn cz_comm_all_gather(c: *mut cz_comm, buf_dev: *mut c_void, bytes_per_rank: u64, stream: *mut c_void) -> c_int;
    pub fn cz_comm_all_reduce_sum_f64(c: *mut cz_comm, buf_dev: *mut c_double, n: u64, stream: *mut c_void) -> c_int;
    pub fn cz_pagerank_sharded(comm: *mut cz_comm, plan: *mut cz_pagerank_plan, rows_per_rank: u32, tolerance: c_double,
                               max_iter: u32, flags: u32, iters_run: *mut u32, final_err: *mut c_double, poison: *const u8,
                               stream: *mut c_void) -> c_int;
    pub fn cz_bfs_multi(out_offsets: *const u32, out_targets: *const u32, n: u32, e: u64, n_gpus: c_int, starts: *const u32, n_starts: u32,
                        goals: *const u32, n_goals: u32, share_visited: c_int, parent: *mut u32, depth: *mut u32, order: *mut u32,
                        n_reached: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_sssp_multi(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64, n_gpus: c_int,
                         starts: *const u32, n_starts: u32, dist: *mut c_float, parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_connected_components_multi(offsets: *const u32, targets: *const u32, n: u32, e: u64, n_gpus: c_int, group: *mut u32,
                                         n_groups: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_pagerank_sharded_overlapped(comm: *mut cz_comm, plan_first: *mut cz_pagerank_plan, plan_second: *mut cz_pagerank_plan,
                                          rows_per_rank: u32, half_rows: u32, tolerance: c_double, max_iter: u32, iters_run: *mut u32,
                                          final_err: *mut c_double, poison: *const u8, stream: *mut c_void) -> c_int;
    pub fn cz_comm_multi_shutdown();
    pub fn cz_pagerank_multi(in_offsets: *const u32, in_sources: *const u32, out_degree: *const u32, n: u32, e: u64,
                             damping: c_float, tolerance: c_double, max_iter: u32, n_gpus: c_int, flags: u32,
                             scores: *mut c_float, iters_run: *mut u32, final_err: *mut c_double, poison: *const u8) -> c_int;
    pub fn cz_hnsw_multi_build(vectors: *const c_float, n: u32, dim: u32, metric: c_int, m: u32, ef_construction: u32,
                               keep_pruned_connections: c_int, seed: u64, max_batch: u32, n_gpus: c_int, flags: u32, n_dist: *mut u64,
                               out: *mut *mut cz_hnsw_multi) -> c_int;
    pub fn cz_hnsw_multi_create(shards: *const *const cz_hnsw_desc, vectors: *const *const c_float, id_offsets: *const u64, n_gpus: c_int,
                                out: *mut *mut cz_hnsw_multi) -> c_int;
    pub fn cz_hnsw_multi_search(m: *mut cz_hnsw_multi, queries: *const c_float, b: u32, k: u32, ef: u32, ids: *mut u64, dist: *mut c_double,
                                count: *mut u32) -> c_int;
    pub fn cz_hnsw_multi_shards(m: *const cz_hnsw_multi, id_offsets: *mut u64) -> c_int;
    pub fn cz_hnsw_multi_destroy(m: *mut cz_hnsw_multi);
    pub fn cz_hnsw_search_sharded(comm: *mut cz_comm, shard: *mut cz_hnsw_index, queries_dev: *const c_float, b: u32, k: u32,
                                  ef: u32, id_offset: u64, out_ids_dev: *mut u64, out_dist_dev: *mut c_double,
                                  out_count_dev: *mut u32, stream: *mut c_void) -> c_int;

    pub fn cz_connected_components_sharded(comm: *mut cz_comm, offsets_local: *const u32, targets: *const u32, n: u32, row_begin: u32,
                                           row_end: u32, e_local: u64, group: *mut u32, n_groups: *mut u32, rounds: *mut u32,
                                           poison: *const u8) -> c_int;
    pub fn cz_bfs_sharded(comm: *mut cz_comm, out_offsets_local: *const u32, out_targets: *const u32, n: u32, row_begin: u32,
                          row_end: u32, e_local: u64, starts: *const u32, n_starts: u32, goals: *const u32, n_goals: u32,
                          share_visited: c_int, parent: *mut u32, depth: *mut u32, order: *mut u32, n_reached: *mut u32,
                          poison: *const u8) -> c_int;
    pub fn cz_sssp_sharded(comm: *mut cz_comm, out_offsets_local: *const u32, out_targets: *const u32, weights: *const c_float,
                           n: u32, row_begin: u32, row_end: u32, e_local: u64, starts: *const u32, n_starts: u32,
                           dist: *mut c_float, parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_sssp_sharded_last_stats(out4: *mut u64) -> c_int;

    pub fn cz_bfs(out_offsets: *const u32, out_targets: *const u32, n: u32, e: u64, starts: *const u32, n_starts: u32,
                  goals: *const u32, n_goals: u32, share_visited: c_int, parent: *mut u32, depth: *mut u32, order: *mut u32,
                  n_reached: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_bfs_shared(out_offsets: *const u32, out_targets: *const u32, n: u32, e: u64, starts: *const u32, n_starts: u32,
                         parent: *mut u32, order: *mut u32, first: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_bfs_shared_until(out_offsets: *const u32, out_targets: *const u32, n: u32, e: u64, starts: *const u32, n_starts: u32,
                               on_level: cz_bfs_level_fn, ctx: *mut c_void, parent: *mut u32, order: *mut u32, first: *mut u32,
                               poison: *const u8) -> c_int;
    pub fn cz_connected_components(offsets: *const u32, targets: *const u32, n: u32, e: u64, group: *mut u32,
                                   n_groups: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_clustering_coefficients(offsets: *const u32, targets: *const u32, n: u32, e: u64, n_triangles: *mut u64,
                                      degree: *mut u32, poison: *const u8, flags: u32) -> c_int;
    pub fn cz_sssp(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64,
                   starts: *const u32, n_starts: u32, dist: *mut c_float, parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_graph_upload(offsets: *const u32, targets: *const u32, weights: *const c_float, n: u32, e: u64, out: *mut *mut cz_graph) -> c_int;
    pub fn cz_graph_destroy(g: *mut cz_graph);
    pub fn cz_graph_acquire(key_hi: u64, key_lo: u64, offsets: *const u32, targets: *const u32, weights: *const c_float, n: u32, e: u64,
                            out: *mut *mut cz_graph, cache_hit: *mut c_int) -> c_int;
    pub fn cz_graph_release(key_hi: u64, key_lo: u64, g: *mut cz_graph);
    pub fn cz_graph_cache_clear();
    pub fn cz_bfs_on(g: *const cz_graph, starts: *const u32, n_starts: u32, goals: *const u32, n_goals: u32, share_visited: c_int,
                     parent: *mut u32, depth: *mut u32, order: *mut u32, n_reached: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_connected_components_on(g: *const cz_graph, group: *mut u32, n_groups: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_sssp_goals(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64, starts: *const u32,
                         n_starts: u32, goals: *const u32, n_goals: u32, dist: *mut c_float, parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_sssp_goals_on(g: *const cz_graph, starts: *const u32, n_starts: u32, goals: *const u32, n_goals: u32, dist: *mut c_float,
                            parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_sssp_on(g: *const cz_graph, starts: *const u32, n_starts: u32, dist: *mut c_float, parent: *mut u32, poison: *const u8) -> c_int;
    pub fn cz_label_propagation(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64,
                                max_iter: u32, labels: *mut u32, iters_run: *mut u32, n_colours: *mut u32, poison: *const u8, flags: u32) -> c_int;
    pub fn cz_graph_last_timing(upload_ms: *mut c_double, device_ms: *mut c_double, download_ms: *mut c_double) -> c_int;
    pub fn cz_closeness(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64,
                        centrality: *mut c_double, poison: *const u8) -> c_int;
    pub fn cz_betweenness(out_offsets: *const u32, out_targets: *const u32, weights: *const c_float, n: u32, e: u64,
                          centrality: *mut c_double, poison: *const u8) -> c_int;
}

// ---- libcozo_ingest.so (host only) ---------------------------------------------------------------------------------
pub const CZI_OK: c_int = 0;
pub const CZI_E_INVALID: c_int = -1;
pub const CZI_E_CORRUPT: c_int = -2;
pub const CZI_E_NOT_AN_EDGE: c_int = -3; // -> NotAnEdgeError (fixed_rule/mod.rs:846-850)
pub const CZI_E_BAD_WEIGHT: c_int = -4; // -> BadEdgeWeightError (fixed_rule/mod.rs:852-860)
pub const CZI_E_UNSUPPORTED: c_int = -5;
pub const CZI_E_TOO_LARGE: c_int = -6;
pub const CZI_E_MISSING_ROW: c_int = -7;
pub const CZI_E_OOM: c_int = -8;

pub const CZI_UNDIRECTED: u32 = 1;
pub const CZI_WEIGHTED: u32 = 2;
pub const CZI_ALLOW_NEGATIVE_WEIGHTS: u32 = 4;
pub const CZI_ORDERED_IDS: u32 = 8;

#[repr(C)]
pub struct czi_graph {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czi_hnsw {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czi_row_buf {
    _private: [u8; 0],
}

#[repr(C)]
pub struct czi_rows {
    pub keys: *const u8,
    pub key_off: *const u64,
    pub vals: *const u8,
    pub val_off: *const u64,
    pub n_rows: u64,
    pub n_key_cols: u32,
}

#[link(name = "cozo_ingest")]
extern "C" {
    pub fn czi_last_error() -> *const c_char;
    pub fn czi_version() -> *const c_char;

    pub fn czi_graph_ingest(rel: *const czi_rows, flags: u32, out: *mut *mut czi_graph) -> c_int;
    pub fn czi_graph_free(g: *mut czi_graph);
    pub fn czi_graph_node_count(g: *const czi_graph) -> u32;
    pub fn czi_graph_edge_count(g: *const czi_graph) -> u64;
    pub fn czi_graph_csr(g: *const czi_graph, inverse: c_int, offsets: *mut u32, targets: *mut u32, weights: *mut c_float) -> c_int;
    pub fn czi_graph_node_keys(g: *const czi_graph, bytes: *mut *const u8, off: *mut *const u64) -> c_int;
    pub fn czi_graph_lookup(g: *const czi_graph, key: *const u8, len: u64) -> u32;

    pub fn czi_hnsw_ingest(idx: *const czi_rows, base: *const czi_rows, vec_fields: *const u32, n_fields: u32, dim: u32,
                           metric: i32, m_max: u32, m_max0: u32, out: *mut *mut czi_hnsw) -> c_int;
    pub fn czi_hnsw_free(h: *mut czi_hnsw);
    pub fn czi_hnsw_desc(h: *const czi_hnsw, desc: *mut cz_hnsw_desc, vectors: *mut *const c_float) -> c_int;
    pub fn czi_hnsw_ingest_f64(idx: *const czi_rows, base: *const czi_rows, vec_fields: *const u32, n_fields: u32, dim: u32,
                               metric: i32, m_max: u32, m_max0: u32, out: *mut *mut czi_hnsw) -> c_int;
    pub fn czi_hnsw_desc_f64(h: *const czi_hnsw, desc: *mut cz_hnsw_desc, vectors: *mut *const c_double) -> c_int;
    pub fn czi_hnsw_nodes(h: *const czi_hnsw, base_row: *mut *const u64, field: *mut *const u32, sub: *mut *const i32) -> c_int;
    pub fn czi_hnsw_row_counts(h: *const czi_hnsw, n_rows: *mut u64, n_self: *mut u64, n_live_links: *mut u64,
                               n_ignored: *mut u64) -> c_int;
    pub fn czi_hnsw_encode_rows(desc: *const cz_hnsw_desc, vectors: *const c_float, node_keys: *const u8,
                                node_key_off: *const u64, level_dist: *const *const c_double, relation_id: u64,
                                out: *mut *mut czi_row_buf) -> c_int;
    pub fn czi_hnsw_encode_rows_degrees(desc: *const cz_hnsw_desc, vectors: *const c_float, node_keys: *const u8,
                                        node_key_off: *const u64, level_dist: *const *const c_double,
                                        level_degree: *const *const c_double, relation_id: u64, out: *mut *mut czi_row_buf) -> c_int;
    pub fn czi_row_buf_rows(b: *const czi_row_buf, rows: *mut czi_rows) -> c_int;
    pub fn czi_row_buf_free(b: *mut czi_row_buf);
}
