//! cozo-core/src/runtime/hnsw_gpu.rs -- HNSW search and index construction on libcozo_gpu.so.
//! Not compiled in this repository's image (no rustc); the executable specification of the same logic is
//! cozo_amd/host/src/hnsw.cpp (GpuHnswIndex::{from_stored, index_rows, hnsw_knn_batch}, HnswSearchRA::iter) and
//! cozo_amd/hnsw.py + cozo_amd/ingest.py.
//!
//! Three pieces:
//!   1. `SessionTx::gpu_hnsw_index`  -- one scan of `tbl:idx` + one of the base relation, as bytes, through libcozo_ingest
//!      into a device-resident index; cached per (index relation id, tx snapshot).
//!   2. `HnswSearchRA::iter`          -- the patch of query/ra.rs:1085-1121: the parent tuples become ONE batch.
//!   3. `SessionTx::hnsw_build_gpu`  -- `::hnsw create` (runtime/relation.rs:1010-1201): build on the device, write the
//!      `tbl:idx` rows back with store_tx.put so that the index stays an ordinary relation.

use std::ffi::CStr;
use std::os::raw::c_int;

use itertools::Itertools;
use miette::{bail, ensure, miette, Result};

use crate::data::program::HnswSearch;
use crate::data::relation::VecElementType;
use crate::data::tuple::{Tuple, TupleT};
use crate::data::value::{DataValue, Vector};
use crate::runtime::relation::RelationHandle;
use crate::runtime::transact::SessionTx;

use super::cozo_gpu_sys::*;
use crate::fixed_rule::algos::gpu::StoredBytes; // the (key bytes, value bytes) buffers of a scan

/// A batch of query vectors in the index' element type: `hnsw_knn` converts the query to the manifest's dtype before anything else
/// (runtime/hnsw.rs:879-884), and an F64 index computes every distance in f64 (VectorCache::dist's F64 arms, :73-78, 86-95, 102-106).
pub(crate) enum QueryBatch {
    F32(Vec<f32>),
    F64(Vec<f64>),
}

pub(crate) struct GpuHnswIndex {
    pub handle: *mut cz_hnsw_index,
    /// the manifest's element type: which of the library's two search entry points the handle takes
    pub f64: bool,
    /// node id -> (position of the base row in the scan, field, sub index) = CompoundKey (runtime/hnsw.rs:55)
    pub node_row: Vec<u64>,
    pub node_field: Vec<u32>,
    pub node_sub: Vec<i32>,
    /// key bytes of every base row (scan order): `base_handle.get` of a result goes through them
    pub base_keys: StoredBytes,
}
unsafe impl Send for GpuHnswIndex {}
unsafe impl Sync for GpuHnswIndex {} // immutable after creation; the library's entry points are re-entrant
impl Drop for GpuHnswIndex {
    fn drop(&mut self) {
        unsafe { cz_hnsw_index_destroy(self.handle) }
    }
}

fn scan_bytes(tx: &SessionTx<'_>, rel: &RelationHandle) -> Result<StoredBytes> {
    let lower = Tuple::default().encode_as_key(rel.id);
    let upper = Tuple::default().encode_as_key(rel.id.next());
    let mut b = StoredBytes::new();
    for kv in tx.store_tx.range_scan(&lower, &upper) {
        let (k, v) = kv?;
        b.keys.extend_from_slice(&k);
        b.key_off.push(b.keys.len() as u64);
        b.vals.extend_from_slice(&v);
        b.val_off.push(b.vals.len() as u64);
    }
    Ok(b)
}

fn check_ingest(rc: c_int) -> Result<()> {
    ensure!(rc == CZI_OK, "libcozo_ingest: {}", unsafe { CStr::from_ptr(czi_last_error()) }.to_string_lossy());
    Ok(())
}
fn check(rc: c_int) -> Result<()> {
    ensure!(rc == CZ_OK, "libcozo_gpu: {}", unsafe { CStr::from_ptr(cz_last_error()) }.to_string_lossy());
    Ok(())
}

impl<'a> SessionTx<'a> {
    /// 1. the export
    pub(crate) fn gpu_hnsw_index(&self, config: &HnswSearch) -> Result<GpuHnswIndex> {
        let mf = &config.manifest;
        let idx = scan_bytes(self, &config.idx_handle)?;
        let base = scan_bytes(self, &config.base_handle)?;
        let k = config.base_handle.metadata.keys.len() as u32;
        let (idx_rows, base_rows) = (idx.view(2 * k + 5), base.view(k));
        let fields = mf.vec_fields.iter().map(|f| *f as u32).collect_vec();
        let mut h = std::ptr::null_mut();
        let f64 = mf.dtype == VecElementType::F64;
        check_ingest(unsafe {
            if f64 {
                czi_hnsw_ingest_f64(&idx_rows, &base_rows, fields.as_ptr(), fields.len() as u32, mf.vec_dim as u32, mf.distance as i32,
                                    mf.m_max as u32, mf.m_max0 as u32, &mut h)
            } else {
                czi_hnsw_ingest(&idx_rows, &base_rows, fields.as_ptr(), fields.len() as u32, mf.vec_dim as u32, mf.distance as i32,
                                mf.m_max as u32, mf.m_max0 as u32, &mut h)
            }
        })?;
        let mut desc = std::mem::MaybeUninit::<cz_hnsw_desc>::uninit();
        let (mut vectors, mut vectors64) = (std::ptr::null(), std::ptr::null());
        unsafe {
            if f64 { czi_hnsw_desc_f64(h, desc.as_mut_ptr(), &mut vectors64) } else { czi_hnsw_desc(h, desc.as_mut_ptr(), &mut vectors) }
        };
        let desc = unsafe { desc.assume_init() };
        let (mut row, mut field, mut sub) = (std::ptr::null(), std::ptr::null(), std::ptr::null());
        unsafe { czi_hnsw_nodes(h, &mut row, &mut field, &mut sub) };
        let n = desc.n as usize;
        let mut handle = std::ptr::null_mut();
        let rc = if desc.n_levels == 0 {
            CZ_OK
        } else if f64 {
            unsafe { cz_hnsw_index_create_f64(&desc, vectors64, &mut handle) }
        } else {
            unsafe { cz_hnsw_index_create(&desc, vectors, &mut handle) }
        };
        let out = GpuHnswIndex {
            handle,
            f64,
            node_row: unsafe { std::slice::from_raw_parts(row, n) }.to_vec(),
            node_field: unsafe { std::slice::from_raw_parts(field, n) }.to_vec(),
            node_sub: unsafe { std::slice::from_raw_parts(sub, n) }.to_vec(),
            base_keys: base,
        };
        unsafe { czi_hnsw_free(h) };
        check(rc)?;
        Ok(out)
    }

    /// SessionTx::hnsw_knn (runtime/hnsw.rs:869-1012) for a whole batch of queries: the traversal on the device, the row
    /// assembly of :939-1006 unchanged (base row, bind columns in all_bindings() order, radius, filter bytecode, truncate).
    pub(crate) fn hnsw_knn_batch(
        &self, gpu: &GpuHnswIndex, queries: &QueryBatch, b: usize, config: &HnswSearch,
        filter_bytecode: &Option<(Vec<crate::data::expr::Bytecode>, crate::parse::SourceSpan)>, stack: &mut Vec<DataValue>,
    ) -> Result<Vec<Vec<Tuple>>> {
        if gpu.handle.is_null() {
            return Ok(vec![vec![]; b]); // empty index (:903-909)
        }
        // no filter: cut to k before rows are fetched; with a filter all ef candidates survive until it has run (:943-947)
        let kk = if config.filter.is_some() { config.ef } else { config.k.min(config.ef) };
        let (mut ids, mut dist, mut cnt) = (vec![0u32; b * kk], vec![0f64; b * kk], vec![0u32; b]);
        check(unsafe {
            match queries {
                QueryBatch::F32(q) => {
                    ensure!(!gpu.f64, "an F64 index takes f64 queries");
                    cz_hnsw_search_batch(gpu.handle, q.as_ptr(), b as u32, kk as u32, config.ef as u32, config.radius.is_some() as c_int,
                                         config.radius.unwrap_or(0.0), ids.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr(),
                                         std::ptr::null_mut(), std::ptr::null() /* hnsw_knn takes no Poison in the reference either */, 0,
                                         std::ptr::null_mut())
                }
                QueryBatch::F64(q) => {
                    ensure!(gpu.f64, "an F32 index takes f32 queries");
                    cz_hnsw_search_batch_f64(gpu.handle, q.as_ptr(), b as u32, kk as u32, config.ef as u32, config.radius.is_some() as c_int,
                                             config.radius.unwrap_or(0.0), ids.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr(),
                                             std::ptr::null_mut(), std::ptr::null(), 0, std::ptr::null_mut())
                }
            }
        })?;
        let keys = &config.base_handle.metadata.keys;
        let mut out = Vec::with_capacity(b);
        for q in 0..b {
            let mut ret = vec![];
            for j in 0..cnt[q] as usize {
                let (node, distance) = (ids[q * kk + j] as usize, dist[q * kk + j]);
                let r = gpu.node_row[node] as usize;
                let key_bytes = &gpu.base_keys.keys[gpu.base_keys.key_off[r] as usize..gpu.base_keys.key_off[r + 1] as usize];
                let mut cand_tuple = match self.store_tx.get(key_bytes, false)? {
                    Some(v) => crate::runtime::relation::decode_tuple_from_kv(key_bytes, &v, None),
                    None => bail!("corrupted index"),
                };
                let (fld, sub) = (gpu.node_field[node] as usize, gpu.node_sub[node]);
                if config.bind_field.is_some() {
                    let name = if fld < keys.len() { keys[fld].name.clone() } else { config.base_handle.metadata.non_keys[fld - keys.len()].name.clone() };
                    cand_tuple.push(DataValue::Str(name));
                }
                if config.bind_field_idx.is_some() {
                    cand_tuple.push(if sub < 0 { DataValue::Null } else { DataValue::from(sub as i64) });
                }
                if config.bind_distance.is_some() {
                    cand_tuple.push(DataValue::from(distance));
                }
                if config.bind_vector.is_some() {
                    let vec = if sub < 0 { cand_tuple[fld].clone() } else {
                        match &cand_tuple[fld] {
                            DataValue::List(v) => v[sub as usize].clone(),
                            v => bail!("corrupted index value {:?}", v),
                        }
                    };
                    cand_tuple.push(vec);
                }
                if let Some((code, span)) = filter_bytecode {
                    if !crate::data::expr::eval_bytecode_pred(code, &cand_tuple, stack, *span)? {
                        continue;
                    }
                }
                ret.push(cand_tuple);
            }
            ret.truncate(config.k); // rows arrive ascending by (distance, node id): the order ret.reverse() produces (:1005)
            out.push(ret);
        }
        Ok(out)
    }
}

// 2. query/ra.rs -- HnswSearchRA::iter with the parent drained into one batch
//
// fn iter<'a>(&'a self, tx: &'a SessionTx<'_>, delta_rule: Option<&MagicSymbol>,
//             stores: &'a BTreeMap<MagicSymbol, EpochStore>) -> Result<TupleIter<'a>> {
//     let bind_idx = /* unchanged, ra.rs:1091-1098 */;
//     let config = self.hnsw_search.clone();
//     let parents: Vec<Tuple> = self.parent.iter(tx, delta_rule, stores)?.try_collect()?;
//     // the query in the index' element type (hnsw.rs:879-884): an F32 index takes f32 (an f64 query is narrowed), an F64 index f64
//     let mut q = if config.manifest.dtype == VecElementType::F64 { QueryBatch::F64(vec![]) } else { QueryBatch::F32(vec![]) };
//     for t in &parents {
//         let DataValue::Vec(v) = &t[bind_idx] else { bail!("Expected vector, got {:?}", t[bind_idx]) };   // ra.rs:1106-1109
//         ensure!(v.len() == config.manifest.vec_dim, "query vector dimension mismatch");
//         match (&mut q, v) {
//             (QueryBatch::F32(q), Vector::F32(v)) => q.extend(v.iter()),
//             (QueryBatch::F32(q), Vector::F64(v)) => q.extend(v.iter().map(|x| *x as f32)),
//             (QueryBatch::F64(q), Vector::F64(v)) => q.extend(v.iter()),
//             (QueryBatch::F64(q), Vector::F32(v)) => q.extend(v.iter().map(|x| *x as f64)),
//         }
//     }
//     let gpu = tx.gpu_hnsw_index_cached(&config)?;
//     let mut stack = vec![];
//     let rows = tx.hnsw_knn_batch(&gpu, &q, parents.len(), &config, &self.filter_bytecode, &mut stack)?;
//     Ok(Box::new(parents.into_iter().zip(rows).flat_map(|(p, rs)| rs.into_iter().map(move |t| {
//         let mut r = p.clone();
//         r.extend(t);
//         Ok(r)
//     }))))
// }

impl<'a> SessionTx<'a> {
    /// 3. `::hnsw create` on the device: every indexed vector of every row, as hnsw_put collects them (hnsw.rs:694-706: each
    /// vec_field, a Vec or every Vec inside a List).  When some row carries several vectors the library is told every node's base
    /// row (cz_hnsw_set_row_of): hnsw_get_neighbours never returns a link between two vectors of one row (:609-610).
    ///
    /// F32 indices only: the device builds, inserts and removes on f32 tables.  The caller (`create_hnsw_index`,
    /// runtime/relation.rs:1010-1201) keeps the reference's own `hnsw_put` loop for a manifest with dtype F64 -- such an index is then
    /// SEARCHED on the device like any other (gpu_hnsw_index above takes its stored rows as f64).
    pub(crate) fn hnsw_build_gpu(&mut self, config: &HnswSearch) -> Result<()> {
        let mf = &config.manifest;
        ensure!(mf.dtype == VecElementType::F32, "hnsw_build_gpu: an F64 index is built by the reference's hnsw_put loop");
        let k = config.base_handle.metadata.keys.len();
        let (mut vectors, mut node_keys, mut node_key_off, mut row_of) = (Vec::<f32>::new(), Vec::<u8>::new(), vec![0u64], Vec::<u32>::new());
        for (row, tuple) in config.base_handle.scan_all(self).enumerate() {
            let tuple = tuple?;
            let mut push = |v: &ndarray::Array1<f32>, fld: usize, sub: i64| {
                vectors.extend(v.iter());
                // the CompoundKey columns [row key.., field, sub index] in their key encoding (data/memcmp.rs:47)
                use crate::data::memcmp::MemCmpEncoder;
                for c in &tuple[..k] {
                    node_keys.encode_datavalue(c);
                }
                node_keys.encode_datavalue(&DataValue::from(fld as i64));
                node_keys.encode_datavalue(&DataValue::from(sub));
                node_key_off.push(node_keys.len() as u64);
                row_of.push(row as u32);
            };
            for fld in &mf.vec_fields {
                match &tuple[*fld] {
                    DataValue::Vec(Vector::F32(v)) => push(v, *fld, -1),
                    DataValue::List(l) => {
                        for (sub, item) in l.iter().enumerate() {
                            if let DataValue::Vec(Vector::F32(v)) = item {
                                push(v, *fld, sub as i64)
                            }
                        }
                    }
                    _ => {}
                }
            }
        }
        let n = (node_key_off.len() - 1) as u32;
        if n == 0 {
            return Ok(());
        }
        let flags = if mf.extend_candidates { CZ_HNSW_EXTEND_CANDIDATES } else { 0 };
        let shared = row_of.windows(2).any(|w| w[0] == w[1]);
        let mut h = std::ptr::null_mut();
        if shared {
            // an empty handle, the base rows, the insert (include/cozo_gpu.h cz_hnsw_set_row_of)
            check(unsafe {
                cz_hnsw_build(std::ptr::null(), 0, mf.vec_dim as u32, mf.distance as c_int, mf.m_neighbours as u32, mf.ef_construction as u32,
                              mf.keep_pruned_connections as c_int, std::ptr::null(), 0, 0, std::ptr::null_mut(), &mut h, 0, std::ptr::null_mut())
            })?;
            check(unsafe { cz_hnsw_set_row_of(h, row_of.as_ptr(), n) })?;
            check(unsafe {
                cz_hnsw_insert(h, vectors.as_ptr(), n, mf.m_neighbours as u32, mf.ef_construction as u32, mf.keep_pruned_connections as c_int,
                               std::ptr::null(), rand::random(), 0, std::ptr::null_mut(), flags, std::ptr::null_mut())
            })?;
        } else {
            check(unsafe {
                cz_hnsw_build(vectors.as_ptr(), n, mf.vec_dim as u32, mf.distance as c_int, mf.m_neighbours as u32, mf.ef_construction as u32,
                              mf.keep_pruned_connections as c_int, std::ptr::null(), rand::random(), 0, std::ptr::null_mut(), &mut h, flags,
                              std::ptr::null_mut())
            })?;
        }
        // export the link tables, recompute the link distances with the kernels' arithmetic, encode the rows
        let (mut nn, mut dim, mut metric, mut n_levels, mut entry) = (0u32, 0u32, 0i32, 0i32, 0u32);
        check(unsafe { cz_hnsw_index_info(h, &mut nn, &mut dim, &mut metric, &mut n_levels, &mut entry) })?;
        let (mut sizes, mut widths) = (vec![0u32; n_levels as usize], vec![0i32; n_levels as usize]);
        let (mut ids, mut nbrs, mut dists, mut degrees) = (vec![], vec![], vec![], vec![]);
        for lv in 0..n_levels as usize {
            check(unsafe { cz_hnsw_index_level_info(h, lv as i32, &mut sizes[lv], &mut widths[lv]) })?;
            // the f64 of the self rows (hnsw.rs:270, 338-357): with extend_candidates not always the number of link rows
            let mut dg = vec![0f64; sizes[lv] as usize];
            check(unsafe { cz_hnsw_index_export_degrees(h, lv as i32, dg.as_mut_ptr()) })?;
            degrees.push(dg);
            let (mut i, mut t) = (vec![0u32; sizes[lv] as usize], vec![0u32; sizes[lv] as usize * widths[lv] as usize]);
            check(unsafe { cz_hnsw_index_export_level(h, lv as i32, i.as_mut_ptr(), t.as_mut_ptr()) })?;
            let (mut pairs, mut slot) = (vec![], vec![]);
            for (s, to) in t.iter().enumerate().filter(|(_, to)| **to != CZ_NONE) {
                pairs.extend([i[s / widths[lv] as usize], *to]);
                slot.push(s);
            }
            let mut d = vec![0f64; slot.len()];
            check(unsafe {
                cz_distance_batch(metric, vectors.as_ptr(), n, dim, vectors.as_ptr(), n, pairs.as_ptr(), slot.len() as u64, d.as_mut_ptr(), 0,
                                  std::ptr::null_mut())
            })?;
            let mut full = vec![0f64; t.len()];
            for (s, x) in slot.iter().zip(d) {
                full[*s] = x;
            }
            ids.push(i);
            nbrs.push(t);
            dists.push(full);
        }
        unsafe { cz_hnsw_index_destroy(h) };
        let (ids_p, nbrs_p, dist_p) = (ids.iter().map(|v| v.as_ptr()).collect_vec(), nbrs.iter().map(|v| v.as_ptr()).collect_vec(),
                                       dists.iter().map(|v| v.as_ptr()).collect_vec());
        let degree_p = degrees.iter().map(|v| v.as_ptr()).collect_vec();
        let desc = cz_hnsw_desc { n, dim, metric, n_levels, entry, level_size: sizes.as_ptr(), level_width: widths.as_ptr(),
                                  level_nodes: ids_p.as_ptr(), level_nbrs: nbrs_p.as_ptr() };
        let mut buf = std::ptr::null_mut();
        check_ingest(unsafe {
            czi_hnsw_encode_rows_degrees(&desc, vectors.as_ptr(), node_keys.as_ptr(), node_key_off.as_ptr(), dist_p.as_ptr(), degree_p.as_ptr(),
                                         config.idx_handle.id.0, &mut buf)
        })?;
        let mut rows = std::mem::MaybeUninit::<czi_rows>::uninit();
        unsafe { czi_row_buf_rows(buf, rows.as_mut_ptr()) };
        let rows = unsafe { rows.assume_init() };
        for r in 0..rows.n_rows as usize {
            let (k0, k1, v0, v1) = unsafe { (*rows.key_off.add(r) as usize, *rows.key_off.add(r + 1) as usize, *rows.val_off.add(r) as usize, *rows.val_off.add(r + 1) as usize) };
            let (key, val) = unsafe { (std::slice::from_raw_parts(rows.keys.add(k0), k1 - k0), std::slice::from_raw_parts(rows.vals.add(v0), v1 - v0)) };
            self.store_tx.put(key, val)?; // what hnsw_put_vector / hnsw_put_fresh_at_levels do row by row (hnsw.rs:277-357, 630-678)
        }
        unsafe { czi_row_buf_free(buf) };
        Ok(())
    }
}

// 4. query/stored.rs:431-450, 486-503 -- put / rm on a relation with an HNSW index, the index resident on the device
//
// The statement's rows are collected first (the reference calls hnsw_put / hnsw_remove per row); the device index takes them
// in one call, the `tbl:idx` rows are encoded again, and only what differs from the stored rows is written.
impl<'a> SessionTx<'a> {
    /// `stored`: the `tbl:idx` rows as the store holds them (one scan, ascending by key -- scan_bytes above);
    /// `encode_rows(gpu)`: the export + cz_distance_batch + czi_hnsw_encode_rows_degrees sequence of hnsw_build_gpu, for the index as it
    /// is now, with the removed nodes' level-0 rows left out.  Both are ascending by key bytes: one merge walk.
    pub(crate) fn hnsw_write_back_delta(&mut self, stored: &StoredBytes, now: &StoredBytes) -> Result<(usize, usize)> {
        let key = |b: &'_ StoredBytes, i: usize| -> (usize, usize) { (b.key_off[i] as usize, b.key_off[i + 1] as usize) };
        let val = |b: &'_ StoredBytes, i: usize| -> (usize, usize) { (b.val_off[i] as usize, b.val_off[i + 1] as usize) };
        let (na, nb) = (stored.key_off.len() - 1, now.key_off.len() - 1);
        let (mut i, mut j, mut puts, mut dels) = (0usize, 0usize, 0usize, 0usize);
        while i < na || j < nb {
            let ord = if j == nb { std::cmp::Ordering::Less } else if i == na { std::cmp::Ordering::Greater } else {
                let ((a0, a1), (b0, b1)) = (key(stored, i), key(now, j));
                stored.keys[a0..a1].cmp(&now.keys[b0..b1])
            };
            match ord {
                std::cmp::Ordering::Less => { // only the store has it: a dropped link, or a row of a removed node (hnsw.rs:434-466, 728-868)
                    let (a0, a1) = key(stored, i);
                    self.store_tx.del(&stored.keys[a0..a1])?;
                    dels += 1;
                    i += 1;
                }
                std::cmp::Ordering::Greater => { // new: a row of an inserted node or a new reverse link (:277-357)
                    let ((b0, b1), (v0, v1)) = (key(now, j), val(now, j));
                    self.store_tx.put(&now.keys[b0..b1], &now.vals[v0..v1])?;
                    puts += 1;
                    j += 1;
                }
                std::cmp::Ordering::Equal => {
                    let ((a0, a1), (v0, v1)) = (val(stored, i), val(now, j));
                    if stored.vals[a0..a1] != now.vals[v0..v1] { // a self row whose degree changed (:338-357)
                        let (b0, b1) = key(now, j);
                        self.store_tx.put(&now.keys[b0..b1], &now.vals[v0..v1])?;
                        puts += 1;
                    }
                    i += 1;
                    j += 1;
                }
            }
        }
        Ok((puts, dels))
    }

    /// hnsw_put for the rows of one statement: cz_hnsw_insert (the new vectors become nodes n .. n + n_new - 1; the shim extends
    /// its node -> CompoundKey table in the same order), then the delta above.  hnsw_remove: cz_hnsw_remove with the nodes of the
    /// deleted rows, then the same delta (their rows and every link row that named them are deleted from the store).
    pub(crate) fn hnsw_put_gpu(&mut self, gpu: &mut GpuHnswIndex, new_vectors: &[f32], new_node_rows: &[u64], config: &HnswSearch) -> Result<()> {
        let mf = &config.manifest;
        let n_new = new_node_rows.len() as u32;
        // the base row of every node, the new ones appended (the caller extends node_field / node_sub the same way); when some row
        // carries several vectors the library needs them: links inside a base row are never read (hnsw.rs:609-610)
        gpu.node_row.extend_from_slice(new_node_rows);
        let mut dense = std::collections::BTreeMap::new();
        let row_of = gpu.node_row.iter().map(|r| { let next = dense.len() as u32; *dense.entry(*r).or_insert(next) }).collect_vec();
        if dense.len() < row_of.len() {
            check(unsafe { cz_hnsw_set_row_of(gpu.handle, row_of.as_ptr(), row_of.len() as u32) })?;
        }
        check(unsafe {
            cz_hnsw_insert(gpu.handle, new_vectors.as_ptr(), n_new, mf.m_neighbours as u32, mf.ef_construction as u32,
                           mf.keep_pruned_connections as c_int, std::ptr::null(), rand::random(), 0, std::ptr::null_mut(),
                           if mf.extend_candidates { CZ_HNSW_EXTEND_CANDIDATES } else { 0 }, std::ptr::null_mut())
        })
    }
}
