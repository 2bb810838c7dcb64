// db_patch.rs -- the lines the GPU shim adds to cozo-core's OWN types (everything else in this directory is new files).
// Not compilable here (no rustc in this image); written against the reference at /root/reference/cozo-core/src.
//
// Why: the device layout of a stored relation (CSR + PageRank plan, cz_pagerank_cached / cz_graph_acquire) is worth
// keeping between `?[] <~ PageRank(*rel[])` calls only if the shim can tell "the same relation, unchanged".  The reference
// has no such notion -- `StoreTx` (storage/mod.rs:31-164) exposes get / put / del / range_scan / commit and nothing about
// versions or snapshots -- so the patch adds a commit counter to `Db` and a copy of what a transaction saw of it to
// `SessionTx`.  Round 2's shim called a `snapshot_version()` on `StoreTx` that does not exist; this file replaces it.
//
// --- runtime/db.rs, struct Db<S> (:97-110): two more fields, initialised to 0 in Db::new (:201-219) ------------------
//     pub(crate) gpu_commits_started: Arc<AtomicU64>,
//     pub(crate) gpu_commits_finished: Arc<AtomicU64>,
//
// --- runtime/transact.rs, struct SessionTx<'a> (:24-30): three more fields ---------------------------------------------
//     pub(crate) gpu_commits_started: Arc<AtomicU64>,   // clones of the Db's counters
//     pub(crate) gpu_commits_finished: Arc<AtomicU64>,
//     pub(crate) gpu_epoch: Option<u64>,                // Some(e): a read-only transaction whose snapshot is commit state e
//
// --- runtime/db.rs, Db::transact (:872-881) and Db::transact_write (:882-891) -----------------------------------------
// `transact` reads the counters around the creation of the store transaction (a seqlock: the snapshot is usable as a
// cache key only if no commit was in flight while it was taken); `transact_write` never is (its own puts are visible to
// its own reads, hnsw_put included).

use std::sync::atomic::{AtomicU64, Ordering};
use std::sync::Arc;

/// What `Db::transact` does instead of building the struct directly (read-only transactions).
pub(crate) fn gpu_epoch_around<T>(started: &Arc<AtomicU64>, finished: &Arc<AtomicU64>, take_snapshot: impl FnOnce() -> T) -> (T, Option<u64>) {
    let f0 = finished.load(Ordering::SeqCst);
    let s0 = started.load(Ordering::SeqCst);
    let snapshot = take_snapshot(); // self.db.transact(false)?  -- storage/mod.rs:15
    let s1 = started.load(Ordering::SeqCst);
    // no commit in flight before the snapshot (s0 == f0) and none begun while it was taken (s1 == s0): the snapshot is
    // exactly the state after `f0` commits
    let epoch = if s0 == f0 && s1 == s0 { Some(f0) } else { None };
    (snapshot, epoch)
}

/// What `SessionTx::commit_tx` (runtime/transact.rs:132-135) becomes for a write transaction: the counters bracket
/// `self.store_tx.commit()` (storage/mod.rs:80).  A failed commit still counts -- the key only has to change whenever
/// the data may have.
pub(crate) fn gpu_commit_bracket(started: &Arc<AtomicU64>, finished: &Arc<AtomicU64>, commit: impl FnOnce() -> miette::Result<()>) -> miette::Result<()> {
    started.fetch_add(1, Ordering::SeqCst);
    let r = commit();
    finished.fetch_add(1, Ordering::SeqCst);
    r
}

impl<'a> crate::runtime::transact::SessionTx<'a> {
    /// The cache key half that stands for "which data": Some(epoch) for a read-only transaction whose snapshot is a
    /// known commit state, None otherwise (write transactions, snapshots taken during a commit) -- then nothing is cached.
    pub(crate) fn gpu_snapshot_key(&self) -> Option<u64> {
        self.gpu_epoch
    }
}
