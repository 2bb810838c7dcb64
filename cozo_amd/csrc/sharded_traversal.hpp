// sharded_traversal.hpp -- ONE BFS / ONE SSSP / ConnectedComponents over a graph whose vertices are partitioned across ranks
// (third row), written once against a backend interface like sharded_pagerank.hpp:
//   * libcozo_gpu instantiates the loops with the HIP + RCCL backend (graph.hip: cz_bfs_sharded / cz_sssp_sharded);
//   * tests/cpp/sharded_driver_test.cpp instantiates the SAME loops with a host backend whose exchange steps run over
//     torch.distributed/gloo with world_size 2, and compares with the single-process oracle.
// Plain C++17, no HIP in here.
//
// Partition: rank r owns the OUT-adjacency of the nodes [row_begin, row_end); every per-node array (depth, parent, claim,
// the packed (cost, parent) words) has full length N on every rank, and every rank runs the same control flow on the same
// reduced values, so all of them take the same branches.
//
// BFS keeps the reference's FIFO semantics (shortest_path_bfs.rs:65-94, bfs.rs:49-98: a node's parent is its FIRST
// discoverer, nodes are discovered in the order the queue pushes them).  The frontier is one globally ordered list; its
// entries are expanded by whichever rank owns them, under their GLOBAL positions:
//   claim   every rank: for its frontier entries, claim[v] = min(claim[v], position) over undiscovered targets v
//           all-reduce(min) of claim (N words)                    <- the exchange that decides every parent
//   count   every rank: how many targets each of ITS entries won (others 0); all-reduce(sum) over the frontier's length
//   scan    exclusive prefix over the counts (identical on every rank) -> where each entry's targets go in the next frontier
//   emit    every rank writes the targets its entries won, in adjacency order, as id + 1 into a zeroed buffer, and their
//           parents; all-reduce(sum) of the buffer (every slot is written by exactly one rank) = the next frontier
//   commit  every rank marks the new nodes' depth
// and once at the end an all-reduce(min) of the parent words (each written by exactly one rank, CZ_NONE elsewhere).
//
// SSSP: dist[v] = min over predecessors of fl32(dist[u] + w) is the unique fixpoint of monotone relaxation, so costs are
// bit-identical to Dijkstra's under any schedule.  Round 4: the schedule is the one-GPU rule's near-far piles, and the exchange is
// SPARSE -- what crosses the links per round is the list of (target, proposed word) pairs, not the N-word state (the first form
// all-reduced N packed words per round: 80 MB at 10M nodes whatever the frontier held).  Every rank keeps the whole state
// (packed (cost << 32 | parent) words, the near list, the far pile, the threshold) and all of them apply the SAME pairs, so the
// replicas stay equal without ever being compared:
//   relax     every rank relaxes the out-edges of the near nodes IT OWNS against the round's starting state and keeps, per
//             target, its best strictly improving proposal: one (word, target) pair per touched target
//   counts    all-gather of one word per rank (its pair count, and its cancellation flag): everyone learns the longest list
//   pairs     all-gather of the lists, padded to the longest
//   apply     every rank lowers the words of the targets with every pair (min); the pair that ends up as a target's word is its
//             one winner, and the winner decides where the target goes: the next near list if its new cost is below the
//             threshold, else the far pile (with that cost: an entry whose node got cheaper later is stale and is dropped)
//   bucket    when the near list runs dry: threshold = the cheapest live far entry + the bucket width (the mean edge weight);
//             live far entries below it become the near list.  No exchange: every rank holds the same pile.
// A proposal never has the cost a node already has, so no parent pointer is ever replaced at equal cost: the predecessor
// graph stays a tree.  Afterwards the parents are made canonical exactly like the single-GPU rule does it (the smallest
// tight predecessor of strictly smaller cost): per-rank candidates + an all-reduce(min).
#pragma once
#include <cstddef>
#include <cstdint>

namespace czs {

enum { TRAVERSAL_OK = 0, TRAVERSAL_CANCELLED = 1 };

// Backend interface, both:
//   int any_poisoned(bool mine, bool *any)           all-reduce(sum) of one word: is ANY rank's cancellation flag set?
// Backend interface, BFS:
//   int bfs_reset(bool keep_visited)                 parent = NONE everywhere; unless keep_visited: depth = claim = NONE
//   int bfs_seed(uint32_t start, bool *already)      *already = depth[start] != NONE; else depth[start] = 0, order[0] = start
//   int bfs_claim(uint32_t lo, uint32_t fsize)       as above, for the owned entries of order[lo .. lo + fsize)
//   int reduce_claim()                               all-reduce(min) over the N claim words
//   int bfs_count(uint32_t lo, uint32_t fsize)       cnt[0 .. fsize): owned entries' counts, 0 elsewhere
//   int reduce_counts(uint32_t fsize)                all-reduce(sum)
//   int bfs_scan(uint32_t fsize, uint32_t *total)    pos = exclusive scan of cnt; *total on the host
//   int bfs_emit(uint32_t lo, uint32_t fsize, uint32_t total, uint32_t next_depth)
//   int reduce_next(uint32_t total)                  all-reduce(sum) of the emit buffer
//   int bfs_commit(uint32_t at, uint32_t total, uint32_t next_depth)   order[at + j] = buffer[j] - 1, depth of those = next_depth
//   int goals_left(uint32_t start, uint32_t *left)   goals still without a backtrace entry (the start never gets one)
//   int reduce_parents()                             all-reduce(min) over the N parent words
template <class B>
int run_sharded_bfs(B &b, uint32_t start, uint32_t N, bool has_goals, bool keep_visited, const volatile uint8_t *poison,
                    uint32_t *reached_out) {
    int rc;
    uint32_t reached = 0;
    if ((rc = b.bfs_reset(keep_visited))) return rc;
    bool already = false;
    bool run = start < N;
    if (run && (rc = b.bfs_seed(start, &already))) return rc;
    if (already) run = false;  // algos/bfs.rs:52-54: a start that an earlier traversal reached is skipped
    if (run) {
        uint32_t lo = 0, fsize = 1, level = 0;
        while (fsize > 0) {
            bool cancel = false;  // collective: every rank leaves at the same level when ANY rank's Poison is set
            if ((rc = b.any_poisoned(poison && *poison, &cancel))) return rc;
            if (cancel) return TRAVERSAL_CANCELLED;
            if ((rc = b.bfs_claim(lo, fsize))) return rc;
            if ((rc = b.reduce_claim())) return rc;
            if ((rc = b.bfs_count(lo, fsize))) return rc;
            if ((rc = b.reduce_counts(fsize))) return rc;
            uint32_t total = 0;
            if ((rc = b.bfs_scan(fsize, &total))) return rc;
            if ((rc = b.bfs_emit(lo, fsize, total, level + 1))) return rc;
            if ((rc = b.reduce_next(total))) return rc;
            if ((rc = b.bfs_commit(lo + fsize, total, level + 1))) return rc;
            lo += fsize;
            fsize = total;
            reached += total;
            level++;
            if (has_goals) {
                uint32_t left = 0;
                if ((rc = b.goals_left(start, &left))) return rc;
                if (left == 0) break;  // the traversal stops after the level in which the last goal was discovered
            }
        }
    }
    if ((rc = b.reduce_parents())) return rc;
    if (reached_out) *reached_out = reached;
    return TRAVERSAL_OK;
}

// Backend interface, SSSP:
//   int sssp_seed(uint32_t start, uint32_t *n_near)  every word = (inf, NONE); the start's = (0.0, NONE); near = [start]; far empty;
//                                                    threshold = the bucket width
//   int sssp_relax(uint32_t n_near)                  the owned near nodes' out-edges -> this rank's pair list
//   int exchange_counts(bool poisoned, uint32_t *longest, bool *any_poisoned)
//   int exchange_pairs(uint32_t longest)             all-gather of the lists, each padded to `longest` pairs
//   int sssp_apply(uint32_t longest, uint32_t *n_near, uint32_t *n_far)   as above; near = this round's winners below the threshold
//   int sssp_next_bucket(uint32_t *n_near, uint32_t *n_far)               as above; both 0 when no live far entry is left
//   int sssp_canonical_parents()                     per-rank smallest tight predecessor of strictly smaller cost ...
//   int reduce_canonical()                           ... all-reduce(min) over N words; a backend's unpack prefers it
template <class B>
int run_sharded_sssp(B &b, uint32_t start, uint32_t N, const volatile uint8_t *poison, uint32_t *rounds_out = nullptr) {
    int rc;
    uint32_t n_near = 0, n_far = 0, rounds = 0;
    if ((rc = b.sssp_seed(start, &n_near))) return rc;
    if (start >= N) n_near = 0;
    for (;;) {
        if (n_near == 0) {
            if (n_far == 0) break;
            if ((rc = b.sssp_next_bucket(&n_near, &n_far))) return rc;  // local: every rank holds the same pile
            continue;
        }
        if ((rc = b.sssp_relax(n_near))) return rc;
        uint32_t longest = 0;
        bool cancel = false;  // collective: every rank leaves in the same round when ANY rank's Poison is set
        if ((rc = b.exchange_counts(poison && *poison, &longest, &cancel))) return rc;
        if (cancel) return TRAVERSAL_CANCELLED;
        rounds++;
        n_near = 0;
        if (longest == 0) continue;  // nothing improved anywhere
        if ((rc = b.exchange_pairs(longest))) return rc;
        if ((rc = b.sssp_apply(longest, &n_near, &n_far))) return rc;
    }
    if (rounds_out) *rounds_out = rounds;
    if ((rc = b.sssp_canonical_parents())) return rc;
    return b.reduce_canonical();
}

// ConnectedComponents over a vertex partition of the SYMMETRISED graph (strongly_connected_components.rs:42-77, strong = false;
// an all-reduce(min) of the u32 label vector per round).  `comp` is a forest over all N nodes whose
// pointers always lead to a lower-or-equal index of the same component, identical on every rank at round boundaries:
//   round   every rank links the endpoints of ITS rows' edges in its copy (union-find, higher root under lower) and compresses;
//           all-reduce(min) of the N pointers -- the elementwise minimum of such forests is such a forest --; compress again
//   stop    when a round leaves the forest as it found it: then no edge of any rank joins two trees, and a root is the smallest
//           member of its component -- the label the reference's group numbering is the rank of.
// Backend interface:
//   int cc_init()                               comp = prev = 0 .. N-1
//   int cc_local_round()                        link the endpoints of every local edge in comp; compress
//   int reduce_labels()                         all-reduce(min) over the N words of comp
//   int cc_settle(bool *changed)                compress comp; *changed = comp != prev (the same on every rank); prev = comp
//   int cc_number_groups()                      group[v] = rank of comp[v] among the roots, ascending
template <class B>
int run_sharded_cc(B &b, const volatile uint8_t *poison, uint32_t *rounds_out) {
    int rc;
    uint32_t rounds = 0;
    if ((rc = b.cc_init())) return rc;
    for (;;) {
        bool cancel = false;
        if ((rc = b.any_poisoned(poison && *poison, &cancel))) return rc;
        if (cancel) return TRAVERSAL_CANCELLED;
        if ((rc = b.cc_local_round())) return rc;
        if ((rc = b.reduce_labels())) return rc;
        bool changed = false;
        if ((rc = b.cc_settle(&changed))) return rc;
        rounds++;
        if (!changed) break;
    }
    if (rounds_out) *rounds_out = rounds;
    return b.cc_number_groups();
}

}  // namespace czs
