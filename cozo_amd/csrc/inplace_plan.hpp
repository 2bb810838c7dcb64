// inplace_plan.hpp -- the static layout of the level-scheduled Gauss-Seidel PageRank sweep (host code only, no HIP).
//
// What it is the layout OF: graph::page_rank under the reading that refreshes `out_scores[u]` inside the per-node loop
// (fixed_rule/algos/pagerank.rs:47-50 -> graph 0.3.1, source not in the reference tree; oracle orc_pagerank_mode(ORC_PR_INPLACE)).
// On one rayon thread that is an ascending Gauss-Seidel sweep: node u adds, in ascending source order, the NEW contribution of
// every in-neighbour v < u and the OLD one of every v >= u.  csrc/pagerank_inplace.hip runs it on the device; this header decides
// where every value travels and is shared with tests/cpp/inplace_plan_test.cpp, which walks the same arrays on the CPU (an emulation
// of the kernels, element for element) against the oracle -- the layout is checked where there is no GPU.
//
// Levels.  level(u) = 0 if no in-neighbour is below u, else 1 + max level(v) over in-neighbours v < u: nodes of one level are
// independent, a sweep is the levels in order.  Nodes are renumbered LEVEL-MAJOR (level, then id): a level's rows, scores and
// contributions are contiguous.  An edge v -> u is
//   Y ("old")     v >= u: the value is v's contribution of the PREVIOUS sweep;
//   U ("urgent")  v <  u and level(u) - level(v) <= urgent_gap: gathered straight from the contribution vector (written by the
//                 launch or two before: an L2-sized window);
//   X ("new")     v <  u, any other: this sweep's contribution of v, produced at least urgent_gap + 1 levels earlier.
// X and Y values travel as in csrc/pagerank.hip's blocked formulation, because a 4-byte gather from a 40 MB vector moves a whole
// cache line (<= 215 G/s even from L2): phase A stages a SLICE of consecutive level-major nodes in LDS and writes the values of the
// slice's out-edges as one coalesced stream, in (slice, class, row block, tile position) order; phase B of a ROW BLOCK (consecutive
// rows of one level, <= `tile` in-edges) reads its values back -- short runs, one per (slice, class) -- drops each at
// its CSR position inside an LDS tile and adds every row in order, one lane per row.  Phase A of level l needs the level's new
// contributions and writes X values for this sweep and Y values for the NEXT one (two Y streams take turns, so a value the current
// sweep still has to read is never overwritten); nobody reads what it writes before level l + urgent_gap + 1, so it rides in the
// launch of level l + urgent_gap, beside that level's phase B, instead of standing between two levels.
// A run is (block's in-edges) / (slices it draws from) values long: tiles and slices are as large as the LDS allows.
// Rows of more than `tile` in-edges ("long") are gathered by a workgroup each, as before.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace czgs {

constexpr uint32_t kOldBit = 0x80000000u;  // long rows' source ids: read the previous sweep's contribution
constexpr uint32_t kYBit = 0x80000000u;    // stream element: position inside the Y stream (else X)

struct Params {
    uint32_t tile = 16384;           // f32 values of a row block's LDS tile (<= 32768: positions are u16)
    uint32_t rows_per_block = 2048;  // = rows per lane x threads of a phase-B workgroup
    uint32_t slice = 16384;          // nodes of a phase-A slice (f32 words of LDS, + 4; <= 65532: local ids are u16)
    uint32_t part = 16384;           // stream positions of one phase-A work item (multiple of 4)
    uint32_t urgent_gap = 1;         // forward edges over at most this many levels are gathered; 0: none (phase A between the levels)
    uint32_t max_levels = 4096;
    uint32_t threads = 0;            // host threads of the two edge passes (0: up to 16; the arrays do not depend on it)
    bool jacobi = false;             // every edge reads the PREVIOUS sweep's contribution: one level, all edges class Y -- graph::page_rank's
                                     // other reading (oracle orc_pagerank) in this layout (an experiment: csrc/pagerank.hip is its product path)
};

struct Block {   // phase B work item
    uint32_t row0, row1;  // level-major rows [row0, row1)
    uint32_t e0;          // off2[row0]: the tile's first CSR position
    uint32_t g0, g1;      // its stream groups (gpos / gperm): four consecutive stream positions each
    uint32_t u0, u1;      // its urgent elements (upos / usrc)
    uint32_t pad;
};
struct Item {    // phase A work item
    uint32_t begin, end;  // positions [begin, end) of the class's stream (begin is a multiple of 4)
    uint32_t node0, n;    // the slice: level-major nodes [node0, node0 + n)
    uint32_t cls, pad0, pad1, pad2;  // 0 = X, 1 = Y
};

// the large per-edge arrays: storage that is NOT zero-filled when it is sized (a std::vector would touch every page once more,
// on one thread: 0.4 s of the 10M / 100M build); pass 2 writes every element, padding included
template <typename T>
struct PodVec {
    using value_type = T;
    std::unique_ptr<T[]> buf;
    size_t n = 0;
    void resize_uninit(size_t k) {
        buf.reset(new T[k ? k : 1]);
        n = k;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T *data() { return buf.get(); }
    const T *data() const { return buf.get(); }
    T &operator[](size_t i) { return buf[i]; }
    const T &operator[](size_t i) const { return buf[i]; }
};

struct Plan {
    Params prm;
    uint32_t N = 0, L = 0;
    uint64_t E = 0;
    std::vector<uint32_t> order;       // [N]  level-major row -> node
    std::vector<uint32_t> first;       // [L + 1] first level-major row of every level
    std::vector<uint32_t> off2;        // [N + 1] in-edge offsets of the level-major rows
    std::vector<uint32_t> od;          // [N]  out-degree, level-major
    std::vector<Block> blocks;
    std::vector<uint32_t> blk_first;   // [L + 1]
    std::vector<Item> items;
    std::vector<uint32_t> item_first;  // [L + 1]
    std::vector<uint32_t> long_rows;   // level-major rows of more than `tile` in-edges
    std::vector<uint32_t> long_first;  // [L + 1]
    std::vector<uint32_t> long_off;    // [n_long + 1] into long_src
    std::vector<uint32_t> long_src;    // level-major source | kOldBit, in the row's own order
    PodVec<uint16_t> asrc[2];          // [nX + 4], [nY + 4]: LDS word of every stream position's source = id inside the slice + (node0 & 3); padding: 0
    uint64_t n_pos[2] = {0, 0};        // stream lengths incl. padding
    PodVec<uint32_t> gpos;             // stream groups of the blocks: position of the group's first value (a multiple of 4) | kYBit
    PodVec<uint16_t> gperm;            //   ... and where its four values go inside the tile (padding: `tile`, the spare words)
    PodVec<uint16_t> upos;             // urgent elements: tile position
    PodVec<uint32_t> usrc;             //   ... and the level-major source
    uint64_t n_edges[3] = {0, 0, 0};   // X, Y, U edges of block rows
    uint64_t n_long_edges = 0;
    std::string error;
};

// in_off [N + 1] (u32 or u64 offsets), in_src [E] ascending inside every row (validated by the caller), out_deg [N]
template <typename OffT>
inline bool build_plan(const OffT *in_off, const uint32_t *in_src, const uint32_t *out_deg, uint32_t N, const Params &prm, Plan &p) {
    p = Plan();
    p.prm = prm;
    p.N = N;
    p.E = N ? (uint64_t)in_off[N] : 0;
    if (prm.tile > 32768 || prm.tile < 64 || prm.slice > 65532 || prm.slice < 4 || (prm.part & 3u) || prm.part == 0 || prm.rows_per_block == 0) {
        p.error = "bad parameters";
        return false;
    }
    if (N >= kOldBit || p.E >= 0x7FFFFFF0ull) {
        p.error = "node ids must stay below 2^31 and E below 2^31 - 16";
        return false;
    }
    if (N == 0) return true;
    const bool trace = getenv("CZ_PLAN_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[inplace plan] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- levels: the in-neighbours below u are final when u is reached
    std::vector<uint32_t> level(N);
    uint32_t L = 0;
    for (uint32_t u = 0; u < N; u++) {
        uint32_t best = 0;
        for (uint64_t e = in_off[u]; e < (uint64_t)in_off[u + 1]; e++) {
            const uint32_t v = in_src[e];
            if (v >= u || prm.jacobi) break;  // ascending: the rest are "old"
            best = std::max(best, level[v] + 1u);
        }
        level[u] = best;
        L = std::max(L, best + 1u);
    }
    if (L > prm.max_levels) {
        p.error = "the graph has " + std::to_string(L) + " dependence levels (more than " + std::to_string(prm.max_levels) +
                  ": a chain-like graph): the level-scheduled sweep would be a launch per level; use cz_pagerank or the CPU path";
        return false;
    }
    p.L = L;
    lap("levels");
    // ---- level-major numbering
    p.first.assign(L + 1, 0);
    for (uint32_t u = 0; u < N; u++) p.first[level[u] + 1]++;
    for (uint32_t l = 0; l < L; l++) p.first[l + 1] += p.first[l];
    p.order.resize(N);
    std::vector<uint32_t> inv(N);
    {
        std::vector<uint32_t> cur(p.first.begin(), p.first.end() - 1);
        for (uint32_t u = 0; u < N; u++) {
            const uint32_t i = cur[level[u]]++;
            p.order[i] = u;
            inv[u] = i;
        }
    }
    p.off2.resize((size_t)N + 1);
    p.od.resize(N);
    p.off2[0] = 0;
    for (uint32_t i = 0; i < N; i++) {
        const uint32_t u = p.order[i];
        p.off2[i + 1] = p.off2[i] + (uint32_t)(in_off[u + 1] - in_off[u]);
        p.od[i] = out_deg[u];
    }
    lap("level-major numbering");
    // ---- slices: consecutive nodes of ONE level
    std::vector<uint32_t> slice_first(L + 1, 0);
    for (uint32_t l = 0; l < L; l++) slice_first[l + 1] = slice_first[l] + (p.first[l + 1] - p.first[l] + prm.slice - 1) / prm.slice;
    const uint32_t S = slice_first[L];
    // what pass 1 needs of an edge's SOURCE, in one 16-byte record (one cache miss per edge instead of two, and no division per edge)
    struct SrcInfo {
        uint32_t inv, level, bucket0;  // level-major id; level; 2 * slice
        uint16_t loc, pad;             // id inside the slice + the slice's misalignment (phase A stages aligned 16-byte vectors, LDS word 0 = node (node0 & ~3))
    };
    std::unique_ptr<SrcInfo[]> info(new SrcInfo[N]);
    for (uint32_t u = 0; u < N; u++) {
        const uint32_t l = level[u], i = inv[u], rel = i - p.first[l], sl = rel / prm.slice;
        info[u] = SrcInfo{i, l, 2u * (slice_first[l] + sl), (uint16_t)(rel - sl * prm.slice + ((p.first[l] + sl * prm.slice) & 3u)), 0};
    }
    // ---- row blocks and long rows, level by level
    p.blk_first.assign(L + 1, 0);
    p.long_first.assign(L + 1, 0);
    p.long_off.push_back(0);
    for (uint32_t l = 0; l < L; l++) {
        p.blk_first[l] = (uint32_t)p.blocks.size();
        p.long_first[l] = (uint32_t)p.long_rows.size();
        uint32_t i = p.first[l];
        while (i < p.first[l + 1]) {
            if (p.off2[i + 1] - p.off2[i] > prm.tile) {
                p.long_rows.push_back(i);
                const uint32_t u = p.order[i];
                for (uint64_t e = in_off[u]; e < (uint64_t)in_off[u + 1]; e++) {
                    const uint32_t v = in_src[e];
                    p.long_src.push_back(inv[v] | ((v >= u || prm.jacobi) ? kOldBit : 0u));
                }
                p.long_off.push_back((uint32_t)p.long_src.size());
                p.n_long_edges += p.off2[i + 1] - p.off2[i];
                i++;
                continue;
            }
            Block b{i, i, p.off2[i], 0, 0, 0, 0, 0};
            while (b.row1 < p.first[l + 1] && b.row1 - b.row0 < prm.rows_per_block && p.off2[b.row1 + 1] - p.off2[b.row1] <= prm.tile &&
                   p.off2[b.row1 + 1] - b.e0 <= prm.tile)
                b.row1++;
            p.blocks.push_back(b);
            i = b.row1;
        }
    }
    p.blk_first[L] = (uint32_t)p.blocks.size();
    p.long_first[L] = (uint32_t)p.long_rows.size();
    lap("row blocks");
    // ---- pass 1: every edge of a block row gets its class; stream edges are counted per (slice, class).  Both passes run on
    // `threads` host threads over CONTIGUOUS ranges of blocks (cells of one bucket are laid out in block order, so a thread's
    // cursors start where the threads before it end): the arrays are the same for every thread count.
    // code: urgent = kOldBit | level-major source; stream = bucket (2 * slice + class), local id in `loc`
    std::unique_ptr<uint32_t[]> code(new uint32_t[p.E ? p.E : 1]);  // (not zero-filled: every block-row edge is written by pass 1, nothing else is read)
    std::unique_ptr<uint16_t[]> loc(new uint16_t[p.E ? p.E : 1]);
    const size_t nblk = p.blocks.size();
    uint32_t T = prm.threads ? prm.threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (p.E < (1u << 20) || nblk < 2 * (size_t)T) T = 1;
    std::vector<size_t> tb(T + 1, 0);  // thread t owns blocks [tb[t], tb[t + 1]): about equal shares of the edges
    {
        size_t bi = 0;
        for (uint32_t t = 1; t <= T; t++) {
            const uint64_t goal = p.E / T * t;
            while (bi < nblk && (t == T || p.blocks[bi].e0 < goal)) bi++;
            tb[t] = bi;
        }
        tb[T] = nblk;
    }
    std::vector<std::vector<uint64_t>> cntT(T, std::vector<uint64_t>((size_t)2 * S, 0));
    std::vector<uint64_t> neT((size_t)T * 3, 0);
    auto pass1 = [&](uint32_t tix) {
        std::vector<uint32_t> lcnt((size_t)2 * S, 0), touched;
        std::vector<uint64_t> &cnt = cntT[tix];
        // The rows come in level-major order, i.e. at random places of the input: row -> offsets -> sources -> source records is a
        // chain of cache misses per row.  It is walked three stages ahead (offsets of row r + 12, sources of row r + 6, records of
        // row r + 3) so that the loop itself finds its lines in the cache.
        auto ahead = [&](uint32_t r) {
            if (r + 12 < N) __builtin_prefetch(&in_off[p.order[r + 12]]);
            if (r + 6 < N) {
                const uint64_t e0 = in_off[p.order[r + 6]];
                __builtin_prefetch(&in_src[e0]);
                __builtin_prefetch(&in_src[e0] + 16);
            }
            if (r + 3 < N) {
                const uint32_t u = p.order[r + 3];
                const uint64_t e1 = std::min<uint64_t>(in_off[u + 1], (uint64_t)in_off[u] + 32);
                for (uint64_t e = in_off[u]; e < e1; e++) __builtin_prefetch(&info[in_src[e]]);
            }
        };
        uint64_t ne[2] = {0, 0};
        for (size_t bi = tb[tix]; bi < tb[tix + 1]; bi++) {
            Block &b = p.blocks[bi];
            uint32_t nu = 0;
            touched.clear();
            for (uint32_t r = b.row0; r < b.row1; r++) {
                ahead(r);
                const uint32_t u = p.order[r], lu = level[u];
                uint32_t t = p.off2[r];  // position in the level-major CSR
                for (uint64_t e = in_off[u]; e < (uint64_t)in_off[u + 1]; e++, t++) {
                    const uint32_t v = in_src[e];
                    const SrcInfo &si = info[v];
                    const bool old = v >= u || prm.jacobi;
                    if (!old && lu - si.level <= prm.urgent_gap) {
                        code[t] = kOldBit | si.inv;
                        nu++;
                    } else {
                        const uint32_t bucket = si.bucket0 + (old ? 1u : 0u);
                        code[t] = bucket;
                        loc[t] = si.loc;
                        if (lcnt[bucket]++ == 0) touched.push_back(bucket);
                        ne[old ? 1 : 0]++;
                    }
                }
            }
            uint32_t ng = 0;  // a CELL (this block's elements of one bucket) is padded to whole groups of four
            for (uint32_t c : touched) {
                const uint32_t k4 = (lcnt[c] + 3u) & ~3u;
                cnt[c] += k4;
                ng += k4 / 4;
                lcnt[c] = 0;
            }
            b.g1 = ng;  // (counts for now)
            b.u1 = nu;
            neT[(size_t)tix * 3 + 2] += nu;
        }
        neT[(size_t)tix * 3] += ne[0];
        neT[(size_t)tix * 3 + 1] += ne[1];
    };
    auto run_threads = [&](auto &&fn) {
        if (T == 1) {
            fn(0u);
            return;
        }
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; t++) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    lap("pass 1 set-up");
    run_threads(pass1);
    lap("pass 1 (classes, counts)");
    std::vector<uint64_t> cnt((size_t)2 * S + 1, 0);
    for (uint32_t t = 0; t < T; t++) {
        for (size_t c = 0; c < (size_t)2 * S; c++) cnt[c] += cntT[t][c];
        for (int k = 0; k < 3; k++) p.n_edges[k] += neT[(size_t)t * 3 + k];
    }
    {
        uint64_t g = 0, u = 0;
        for (Block &b : p.blocks) {
            const uint32_t ng = b.g1, nu = b.u1;
            b.g0 = (uint32_t)g;
            b.g1 = (uint32_t)(g += ng);
            b.u0 = (uint32_t)u;
            b.u1 = (uint32_t)(u += nu);
        }
        if (g >= 0x7FFFFFF0ull) {
            p.error = "stream too long";
            return false;
        }
        p.gpos.resize_uninit(g);
        p.gperm.resize_uninit(g * 4);  // (pass 2 sends the padding to the tile's spare words)
        p.upos.resize_uninit(u);
        p.usrc.resize_uninit(u);
    }
    // ---- stream positions (every cell, hence every (slice, class) segment, starts on a multiple of four); phase-A items
    std::vector<uint64_t> start((size_t)2 * S, 0);
    p.item_first.assign(L + 1, 0);
    {
        uint64_t pos[2] = {0, 0};
        for (uint32_t l = 0; l < L; l++) {
            p.item_first[l] = (uint32_t)p.items.size();
            for (uint32_t s = slice_first[l]; s < slice_first[l + 1]; s++) {
                const uint32_t node0 = p.first[l] + (s - slice_first[l]) * prm.slice;
                const uint32_t n = std::min(prm.slice, p.first[l + 1] - node0);
                for (uint32_t c = 0; c < 2; c++) {
                    const uint64_t k = cnt[2 * s + c];
                    start[2 * s + c] = pos[c];
                    for (uint64_t a = 0; a < k; a += prm.part)
                        p.items.push_back(Item{(uint32_t)(pos[c] + a), (uint32_t)(pos[c] + std::min<uint64_t>(k, a + prm.part)), node0, n, c, 0, 0, 0});
                    pos[c] += k;
                }
            }
        }
        p.item_first[L] = (uint32_t)p.items.size();
        for (uint32_t c = 0; c < 2; c++) {
            if (pos[c] >= 0x7FFFFFF0ull) {
                p.error = "stream too long";
                return false;
            }
            p.n_pos[c] = pos[c];
            p.asrc[c].resize_uninit(pos[c] + 4);  // padding positions read LDS word 0 (pass 2 writes those inside the cells)
            for (int k = 0; k < 4; k++) p.asrc[c][pos[c] + k] = 0;
        }
    }
    // ---- pass 2: block by block, its cells in (slice, class) order -- each a run of consecutive stream positions, a group = four
    // of them --, the urgent elements in row order.  Thread t's cursor of a bucket starts behind the cells of the threads before it.
    std::vector<std::vector<uint64_t>> curT(T);
    {
        std::vector<uint64_t> run(start);
        for (uint32_t t = 0; t < T; t++) {
            curT[t] = run;
            for (size_t c = 0; c < (size_t)2 * S; c++) run[c] += cntT[t][c];
        }
    }
    lap("positions, items, arrays");
    auto pass2 = [&](uint32_t tix) {
        std::vector<uint64_t> &cur = curT[tix];
        std::vector<uint32_t> lcnt((size_t)2 * S, 0), touched, cell4;
        for (size_t bi = tb[tix]; bi < tb[tix + 1]; bi++) {
            const Block &b = p.blocks[bi];
            const uint32_t t0 = b.e0, t1 = p.off2[b.row1];
            touched.clear();
            for (uint32_t t = t0; t < t1; t++) {
                const uint32_t c = code[t];
                if (c & kOldBit) continue;
                if (lcnt[c]++ == 0) touched.push_back(c);
            }
            std::sort(touched.begin(), touched.end());
            uint32_t g = b.g0;
            cell4.clear();
            for (uint32_t c : touched) {  // lcnt becomes (the cell's first slot in gperm) + 1 (0 stays "untouched")
                const uint32_t k4 = (lcnt[c] + 3u) & ~3u;
                for (uint32_t j = 0; j < k4 / 4; j++) p.gpos[g + j] = (uint32_t)(cur[c] + 4u * j) | ((c & 1u) ? kYBit : 0u);
                for (uint32_t j = lcnt[c]; j < k4; j++) {  // the cell's padding: tile word `tile` (spare), LDS word 0
                    p.gperm[4 * (size_t)g + j] = (uint16_t)prm.tile;
                    p.asrc[c & 1u][cur[c] + j] = 0;
                }
                lcnt[c] = 4u * g + 1;
                g += k4 / 4;
                cell4.push_back(k4);
            }
            uint32_t uj = b.u0;
            for (uint32_t t = t0; t < t1; t++) {
                const uint32_t c = code[t];
                if (c & kOldBit) {
                    p.upos[uj] = (uint16_t)(t - t0);
                    p.usrc[uj] = c & ~kOldBit;
                    uj++;
                    continue;
                }
                const uint32_t slot = lcnt[c]++ - 1;  // index into gperm; its group's position + (slot & 3) is the stream position
                const uint32_t pos = (p.gpos[slot >> 2] & ~kYBit) + (slot & 3u);
                p.asrc[c & 1u][pos] = loc[t];
                p.gperm[slot] = (uint16_t)(t - t0);
            }
            for (size_t i = 0; i < touched.size(); i++) {
                lcnt[touched[i]] = 0;
                cur[touched[i]] += cell4[i];  // the bucket's next cell (a later block's) follows this one
            }
        }
    };
    run_threads(pass2);
    lap("pass 2 (placement)");
    return true;
}

}  // namespace czgs
