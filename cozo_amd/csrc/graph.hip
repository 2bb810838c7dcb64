// graph.hip -- BFS / ShortestPathBFS, ConnectedComponents and SSSP fixed rules on gfx950
// (C ABI: cz_bfs, cz_connected_components, cz_sssp).  Integer outputs are bit-exact with the reference's
// sequential algorithms; the parallel formulations below say why.
//
// BFS (fixed_rule/algos/shortest_path_bfs.rs:65-94, algos/bfs.rs:49-98): the reference pops a FIFO queue and
//   scans each node's out-edges in sorted order; a node's parent is its FIRST discoverer.  Level-synchronous
//   equivalent: keep the frontier ORDERED (it is the FIFO content); node v on the next level is claimed by the
//   frontier entry with the smallest position that has an edge to it (atomicMin of positions); the next
//   frontier lists, for each frontier entry in order, the targets it won, in adjacency order -- exactly the
//   order in which the reference pushes them.  Three kernels per level + one prefix sum.
// ConnectedComponents (algos/strongly_connected_components.rs:42-77 with strong = false): Tarjan over the
//   symmetrised graph from roots 0,1,2.. numbers components by their smallest node index.  Min-label
//   propagation + pointer jumping converges to label[v] = smallest index of v's component; the group id is
//   the rank of that label among all roots (prefix sum over `label[i] == i`).
// SSSP (algos/shortest_path_dijkstra.rs:274-339): dist[v] = min over predecessors of fl32(dist[u] + w) is the
//   unique fixpoint of monotone relaxation, so ANY schedule that relaxes with the same f32 add and strict `<` until
//   nothing changes reaches bit-identical costs.  (cost, parent) are packed in one u64 and updated by CAS only on a
//   strict improvement, which keeps every parent pointer tight and the predecessor graph a tree.  The schedule is
//   near-far (delta-stepping with two piles): nodes whose tentative cost is below the current threshold are relaxed
//   round by round, the others wait in the far pile until the threshold reaches them -- a plain frontier Bellman-Ford
//   re-relaxed the 10M / 100M graph for 35 rounds at 1 G edges/s.  Several sources share every launch (one (source, node)
//   pair per queue entry), which is what the all-sources rules (Closeness / Betweenness centrality) need on small graphs.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <memory>
#include <mutex>
#include <vector>

#include <rocprim/block/block_radix_sort.hpp>  // (LabelPropagation: a workgroup sorts a chunk of a long list by label)

#include "common.h"

namespace {

// where the last whole-graph rule called on this host thread spent its time (cz_graph_last_timing): the entry points take
// host arrays, so a call is upload + kernels + results back, and only the middle part says anything about the kernels
struct CallTiming {
    double ms[3] = {0, 0, 0};  // upload (H2D + allocation), device (kernels and their control round trips), download
    std::chrono::steady_clock::time_point last;
    void start() {
        ms[0] = ms[1] = ms[2] = 0;
        last = std::chrono::steady_clock::now();
    }
    // everything since the previous lap (the device drained first) is booked under `slot`
    void lap(int slot) {
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        ms[slot] += std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
    }
};
thread_local CallTiming t_timing;

// CZ_SSSP_TRACE: where a call's wall time goes, mark by mark (host clock, the device drained at each mark; scratch/ experiments)
inline void trace_mark(const char *what) {
    static const bool on = getenv("CZ_SSSP_TRACE") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point prev = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "sssp mark %-28s +%.1f us\n", what, std::chrono::duration<double, std::micro>(now - prev).count());
    prev = now;
}
enum { T_UPLOAD = 0, T_DEVICE = 1, T_DOWNLOAD = 2 };

constexpr int kT = 256;
inline int grid_for(uint64_t n, int per_block = kT) {
    uint64_t b = (n + per_block - 1) / per_block;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, 256 * 16));
}

// ---------------------------------------------------------------------------------------------
// exclusive prefix sum (u32), tiles of 1024, recursive on the tile sums
// ---------------------------------------------------------------------------------------------
constexpr int kScanTile = 1024;

__global__ void __launch_bounds__(kT)
scan_tiles_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t wsum[kT / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = base + j < n ? in[base + j] : 0u;
    uint32_t t = v[0] + v[1] + v[2] + v[3];
    // inclusive scan of t across the wave
    uint32_t x = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if ((int)(threadIdx.x & 63) >= off) x += y;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t excl = woff + x - t;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) out[base + j] = excl;
        excl += v[j];
    }
    if (threadIdx.x == kT - 1) sums[blockIdx.x] = woff + x;
}

__global__ void __launch_bounds__(kT)
scan_add_kernel(uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ tile_off) {
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    const uint32_t o = tile_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < n) out[base + j] += o;
}

// scratch must hold at least scan_scratch_words(n) u32.  *d_total (device) receives the grand total.
size_t scan_scratch_words(uint64_t n) {
    size_t w = 0;
    while (n > 1) {
        n = (n + kScanTile - 1) / kScanTile;
        w += 2 * n + 2;
        if (n == 1) break;
    }
    return w + 8;
}
int exclusive_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, uint32_t *d_total, uint32_t *scratch,
                   hipStream_t s) {
    if (n == 0) {
        CZ_HIP(hipMemsetAsync(d_total, 0, 4, s));
        return CZ_OK;
    }
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    uint32_t *sums = scratch, *sums_scan = scratch + tiles + 1;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(kT), 0, s, d_in, d_out, n, sums);
    if (tiles == 1) {
        CZ_HIP(hipMemcpyAsync(d_total, sums, 4, hipMemcpyDeviceToDevice, s));
        return CZ_OK;
    }
    int rc = exclusive_scan(sums, sums_scan, tiles, d_total, scratch + 2 * (tiles + 1), s);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3(tiles), dim3(kT), 0, s, d_out, n, sums_scan);
    return CZ_OK;
}

// ---------------------------------------------------------------------------------------------
// BFS
// ---------------------------------------------------------------------------------------------
// Frontier expansion: a group of kBfsLanes lanes owns one frontier node and reads its adjacency list coalesced (one
// thread per node walked its list serially and uncoalesced: 3.3 ms per pass at the widest level of the 10M / 100M
// graph).  The three passes keep the reference's FIFO order: claim (atomicMin of the frontier position per target),
// count, emit in (frontier position, adjacency position) order; duplicates of a target are adjacent in the sorted list
// and only the first one counts.  Group-wide counts / offsets come from wave ballots.
// Two helpers cut the random traffic (null = the plain reads):
//   vis   one bit per node, set when the node is discovered: 1.25 MB for 10M nodes, i.e. L2-resident, where depth[] is a
//         40 MB array served by the Infinity Cache -- on the dense levels most targets are already visited and never get past it;
//   won   one byte per edge slot, written by the count pass (did this slot win its target?) and read back coalesced by the
//         emit pass, which then does no random reads at all.
constexpr int kBfsLanes = 16;

__device__ __forceinline__ bool bfs_unvisited(const uint32_t *__restrict__ vis, const uint32_t *__restrict__ depth, uint32_t v) {
    return vis ? !((vis[v >> 5] >> (v & 31)) & 1u) : depth[v] == CZ_NONE;
}

__device__ __forceinline__ unsigned int bfs_group_mask(unsigned long long ballot, int lane) {
    return (unsigned int)(ballot >> (lane & ~(kBfsLanes - 1))) & ((1u << kBfsLanes) - 1u);
}

__global__ void __launch_bounds__(kT)
bfs_claim_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                 uint32_t fsize, const uint32_t *__restrict__ depth, const uint32_t *__restrict__ vis, uint32_t *__restrict__ claim,
                 uint32_t rb, uint32_t re) {
    // [rb, re): the nodes whose adjacency this device holds (`off` is relative to rb); the whole graph on one GPU
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    for (uint32_t i = group; i < fsize; i += ngroups) {
        const uint32_t u = frontier[i];
        if (u < rb || u >= re) continue;
        const uint32_t e1 = off[u - rb + 1];
        for (uint32_t e = off[u - rb] + glane; e < e1; e += kBfsLanes) {
            const uint32_t v = tgt[e];
            if (bfs_unvisited(vis, depth, v)) atomicMin(&claim[v], i);
        }
    }
}

__global__ void __launch_bounds__(kT)
bfs_count_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                 uint32_t fsize, const uint32_t *__restrict__ depth, const uint32_t *__restrict__ vis,
                 const uint32_t *__restrict__ claim, uint32_t *__restrict__ cnt, uint8_t *__restrict__ won, uint32_t rb, uint32_t re) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;  // every group of a wave runs the same trip count (ballots)
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        bool live = i < fsize;
        const uint32_t u = live ? frontier[i] : 0;
        live = live && u >= rb && u < re;
        const uint32_t e0 = live ? off[u - rb] : 0, e1 = live ? off[u - rb + 1] : 0;
        uint32_t c = 0;
        // the widest list among the wave's groups decides the trip count
        uint32_t len = e1 - e0, maxlen = len;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
        for (uint32_t b = 0; b < maxlen; b += kBfsLanes) {
            const uint32_t e = e0 + b + glane;
            bool hit = false;
            if (e < e1) {
                const uint32_t v = tgt[e];
                hit = (e == e0 || tgt[e - 1] != v) && bfs_unvisited(vis, depth, v) && claim[v] == i;
                if (won) won[e] = hit;
            }
            c += __popc(bfs_group_mask(__ballot(hit), lane));
        }
        if (live && glane == 0) cnt[i] = c;
    }
}

__global__ void __launch_bounds__(kT)
bfs_emit_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                uint32_t fsize, uint32_t *__restrict__ depth, uint32_t *__restrict__ vis, const uint32_t *__restrict__ claim,
                const uint8_t *__restrict__ won, const uint32_t *__restrict__ pos, uint32_t *__restrict__ next,
                uint32_t *__restrict__ parent, uint32_t next_depth, uint32_t rb, uint32_t re, uint32_t plus_one) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        bool live = i < fsize;
        const uint32_t u = live ? frontier[i] : 0;
        live = live && u >= rb && u < re;
        const uint32_t e0 = live ? off[u - rb] : 0, e1 = live ? off[u - rb + 1] : 0;
        uint32_t o = live ? pos[i] : 0;
        uint32_t maxlen = e1 - e0;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, s, 64));
        for (uint32_t b = 0; b < maxlen; b += kBfsLanes) {
            const uint32_t e = e0 + b + glane;
            bool hit = false;
            uint32_t v = 0;
            if (e < e1) {
                v = tgt[e];
                // `depth[v] == NONE` is read before any lane of this launch can have set it for v: only the claim
                // winner writes depth[v], and the winner is this very lane group (claim[v] == i)
                // (no `won`: the visited bits are not used either, and this is the same test the count pass made)
                hit = won ? won[e] != 0 : (e == e0 || tgt[e - 1] != v) && claim[v] == i && depth[v] == CZ_NONE;
            }
            const unsigned int m = bfs_group_mask(__ballot(hit), lane);
            if (hit) {
                next[o + __popc(m & ((1u << glane) - 1u))] = v + plus_one;
                parent[v] = u;
                depth[v] = next_depth;
                if (vis) atomicOr(&vis[v >> 5], 1u << (v & 31));
            }
            o += __popc(m);
        }
    }
}

// ---- one pass over the frontier's adjacency per level (round 3; the one-GPU rule) ---------------------------------------
// The three passes above read every edge slot of the frontier three times.  With adjacency lists sorted ascending, the
// reference's FIFO order of the next frontier is simply (position of the discoverer in the frontier, node id): so ONE pass
// claims (atomicMin of the frontier position, as before) and lists every node the first time it is claimed; what follows
// is frontier-sized, not edge-sized: count the listed nodes per claimer, prefix sums, place them, order each claimer's few
// nodes by id.  A claimer with many new nodes (a hub early in the search) re-reads its own list instead, which is in order.
// The list of new nodes is written in chunks of kBfsChunk slots that a WAVE reserves for itself: one atomic on the shared
// counter per chunk, not per wave instruction -- a single word takes ~88 M atomics a second (MI355X_MICROARCH.md), and at
// the widest level of the 10M / 100M graph a million appending instructions made this pass 10 ms.  Slots a wave leaves
// unused (the end of a chunk it could not fit a batch into, the end of its last chunk) hold CZ_NONE and are skipped by
// the passes that read the list; `*n_slots` is the number of slots handed out.
constexpr uint32_t kBfsChunk = 256;
constexpr int kBfsNodes = 4;  // frontier nodes a lane group works on at once
#ifndef CZ_BFS_LONG_LIST
#define CZ_BFS_LONG_LIST 4096
#endif
constexpr uint32_t kBfsLongList = CZ_BFS_LONG_LIST;  // adjacency lists beyond this are cut over workgroups (a 454k-edge hub walked
                                                     // by one 16-lane group was 60 of the 64 ms of a BFS on the skewed test graph)
constexpr uint32_t kBfsStretch = 4096;               // ... in stretches of this many edges
__global__ void __launch_bounds__(kT)
bfs_discover_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                    uint32_t fsize, const uint32_t *__restrict__ vis, uint32_t *__restrict__ claim,
                    uint32_t *__restrict__ fresh, uint32_t *__restrict__ n_slots, uint32_t *__restrict__ long_nodes,
                    uint32_t *__restrict__ n_long) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;  // every group of a wave runs the same trip count (ballots)
    uint32_t w_base = 0, w_used = kBfsChunk;  // this wave's chunk (uniform): none yet
    // lists the nodes of `first` lanes (wave-wide) in the wave's chunk
    auto append = [&](bool first, uint32_t v) {
        const unsigned long long m = __ballot(first);
        if (!m) return;
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (w_used + cnt > kBfsChunk) {  // uniform over the wave: close the chunk, take the next
            if (w_used < kBfsChunk)
                for (uint32_t k = w_used + lane; k < kBfsChunk; k += 64) fresh[w_base + k] = CZ_NONE;
            uint32_t nb = 0;
            if (lane == 0) nb = atomicAdd(n_slots, kBfsChunk);
            w_base = __builtin_amdgcn_readfirstlane(nb);
            w_used = 0;
        }
        if (first) fresh[w_base + w_used + __popcll(m & ((1ull << lane) - 1ull))] = v;
        w_used += cnt;
    };
    // nobody had claimed v yet?  The claim word is READ first: a claim at or below position i (a stale copy can only be too
    // high) needs no atomic -- random atomicMin runs at 27 G/s on this part, random loads at 63 (cz_random_access_probe), and at
    // the widest level nine of ten claims on a node come after its first.
    auto claim_first = [&](uint32_t v, uint32_t i) {
        if ((vis[v >> 5] >> (v & 31)) & 1u) return false;
        if (claim[v] <= i) return false;
        return atomicMin(&claim[v], i) == CZ_NONE;
    };
    // A level is a chain of dependent reads per frontier node (frontier -> offsets -> targets -> visited bit -> claim): one
    // node at a time per lane group leaves the wave waiting on each link of the chain.  kBfsNodes nodes go down the chain
    // together; the first 16 targets of each (all of them, for most nodes of a sparse graph) are handled in that form,
    // longer lists finish in the loop below.
    for (uint32_t r = 0; r < rounds; r += kBfsNodes) {
        uint32_t i[kBfsNodes], e0[kBfsNodes], e1[kBfsNodes], v[kBfsNodes];
        bool first[kBfsNodes];
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) {
            i[k] = group + (r + k) * ngroups;
            e0[k] = (r + k < rounds && i[k] < fsize) ? frontier[i[k]] : CZ_NONE;  // (the node, for now)
        }
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) {
            const uint32_t u = e0[k];
            e0[k] = u != CZ_NONE ? off[u] : 0;
            e1[k] = u != CZ_NONE ? off[u + 1] : 0;
            if (e1[k] - e0[k] > kBfsLongList) {  // a hub: its list goes to bfs_discover_long_kernel, cut over workgroups
                if (glane == 0) long_nodes[atomicAdd(n_long, 1u)] = i[k];
                e1[k] = e0[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) v[k] = e0[k] + glane < e1[k] ? tgt[e0[k] + glane] : CZ_NONE;
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) first[k] = v[k] != CZ_NONE && claim_first(v[k], i[k]);
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) append(first[k], v[k]);
#pragma unroll
        for (int k = 0; k < kBfsNodes; k++) {
            uint32_t maxlen = e1[k] - e0[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
            for (uint32_t b = kBfsLanes; b < maxlen; b += kBfsLanes) {
                const uint32_t e = e0[k] + b + glane;
                bool f = false;
                uint32_t t = 0;
                if (e < e1[k]) {
                    t = tgt[e];
                    f = claim_first(t, i[k]);
                }
                append(f, t);
            }
        }
    }
    if (w_used < kBfsChunk)  // (a wave that never listed a node holds no chunk: w_used == kBfsChunk)
        for (uint32_t k = w_used + lane; k < kBfsChunk; k += 64) fresh[w_base + k] = CZ_NONE;
}

// the lists set aside by bfs_discover_kernel: every workgroup takes stretches of kBfsStretch edges of every listed node
__global__ void __launch_bounds__(kT)
bfs_discover_long_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                         const uint32_t *__restrict__ long_nodes, const uint32_t *__restrict__ n_long, const uint32_t *__restrict__ vis,
                         uint32_t *__restrict__ claim, uint32_t *__restrict__ fresh, uint32_t *__restrict__ n_slots) {
    const int lane = threadIdx.x & 63;
    const uint32_t nl = *n_long;
    uint32_t w_base = 0, w_used = kBfsChunk;
    // (every workgroup looks at every listed node: the chain list -> frontier -> offsets is fetched for 64 nodes at a time, a node
    // per lane -- one node at a time, thousands of listed nodes cost every workgroup ~2 us each whether or not it had a stretch)
    for (uint32_t kb = 0; kb < nl; kb += 64) {
        uint32_t l_i = 0, l_e0 = 0, l_e1 = 0;
        if (kb + lane < nl) {
            l_i = long_nodes[kb + lane];
            const uint32_t lu = frontier[l_i];
            l_e0 = off[lu];
            l_e1 = off[lu + 1];
        }
        const uint32_t here = min(64u, nl - kb);
        for (uint32_t q = 0; q < here; q++) {
        const uint32_t k = kb + q;
        const uint32_t i = (uint32_t)__shfl((int)l_i, (int)q, 64), e0 = (uint32_t)__shfl((int)l_e0, (int)q, 64),
                       e1 = (uint32_t)__shfl((int)l_e1, (int)q, 64);
        // (the stretches of node k start at workgroup 37 k mod G: with thousands of listed nodes of 5-100 thousand edges -- a level of
        // an R-MAT graph -- workgroup 0 used to walk the first stretch of EVERY one of them, 40 ms for one level)
        const uint32_t first_wg = (uint32_t)(((uint64_t)k * 37u) % gridDim.x);
        for (uint32_t s0 = e0 + ((blockIdx.x + gridDim.x - first_wg) % gridDim.x) * kBfsStretch; s0 < e1; s0 += gridDim.x * kBfsStretch) {
            const uint32_t s1 = min(e1, s0 + kBfsStretch);
            for (uint32_t b = s0; b < s1; b += kT) {  // (uniform trip count over the workgroup's waves)
                const uint32_t e = b + threadIdx.x;
                bool first = false;
                uint32_t v = 0;
                if (e < s1) {
                    v = tgt[e];
                    first = !((vis[v >> 5] >> (v & 31)) & 1u) && claim[v] > i && atomicMin(&claim[v], i) == CZ_NONE;
                }
                const unsigned long long m = __ballot(first);
                if (m) {
                    const uint32_t cnt = (uint32_t)__popcll(m);
                    if (w_used + cnt > kBfsChunk) {
                        if (w_used < kBfsChunk)
                            for (uint32_t q = w_used + lane; q < kBfsChunk; q += 64) fresh[w_base + q] = CZ_NONE;
                        uint32_t nb = 0;
                        if (lane == 0) nb = atomicAdd(n_slots, kBfsChunk);
                        w_base = __builtin_amdgcn_readfirstlane(nb);
                        w_used = 0;
                    }
                    if (first) fresh[w_base + w_used + __popcll(m & ((1ull << lane) - 1ull))] = v;
                    w_used += cnt;
                }
            }
        }
        }
    }
    if (w_used < kBfsChunk)
        for (uint32_t q = w_used + lane; q < kBfsChunk; q += 64) fresh[w_base + q] = CZ_NONE;
}

// Round 6: a hub that claims 100 000 nodes in one level took 100 000 atomics on ONE counter here and in bfs_place_kernel (a single
// word takes ~88 M atomics a second: 2.3 + 2.5 ms of a 12 ms BFS on the R-MAT bench graph) -- and the list is written in runs of one
// claimer's nodes (a wave's chunk: a hub's stretch, or the lists of the wave's 16-lane groups).  So the lanes of a wave that hold
// nodes of the same claimer as the first unserved lane add once, together; up to kBfsAggRounds such rounds, the rest one by one (a
// round that serves a single lane ends the search: a wave of 64 different claimers -- the uniform graph -- pays one round).
constexpr int kBfsAggRounds = 4;
// ... and where the counter of frontier position i lives: the hubs of an R-MAT level are NEIGHBOURS in the frontier (the lowest ids of
// the start's list), 32 of them shared a 128-byte line of counters, and the atomics of a level -- one request per wave whatever it
// adds -- queued on that line (tally 0.85 ms, place 0.94 ms for 2.9M new nodes; the next level places 1M nodes in 0.12 ms).
// Position i counts in word (i mod 32) * rows + i / 32, rows = ceil(fsize / 32): neighbours in the frontier are `rows` words apart;
// bfs_counts_in_order_kernel reads the counts back in frontier order for the prefix sums.
__device__ __forceinline__ uint32_t bfs_cnt_at(uint32_t i, uint32_t rows) { return (i & 31u) * rows + (i >> 5); }
__global__ void __launch_bounds__(kT)
bfs_counts_in_order_kernel(const uint32_t *__restrict__ cnt, uint32_t fsize, uint32_t rows, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < fsize; i += gridDim.x * blockDim.x) out[i] = cnt[bfs_cnt_at(i, rows)];
}
__global__ void __launch_bounds__(kT)
bfs_tally_kernel(const uint32_t *__restrict__ fresh, const uint32_t *__restrict__ n_slots, const uint32_t *__restrict__ claim,
                 uint32_t *__restrict__ cnt, uint32_t rows) {
    const uint32_t n = *n_slots;
    const int lane = threadIdx.x & 63;
    for (uint32_t k0 = blockIdx.x * blockDim.x; k0 < n; k0 += gridDim.x * blockDim.x) {  // (uniform over the wave: ballots below)
        const uint32_t k = k0 + threadIdx.x;
        const uint32_t v = k < n ? fresh[k] : CZ_NONE;
        bool rem = v != CZ_NONE;
        const uint32_t i = rem ? claim[v] : CZ_NONE;
        for (int r = 0; r < kBfsAggRounds; r++) {
            const unsigned long long mrem = __ballot(rem);
            if (!mrem) break;
            const int lead = __builtin_ctzll(mrem);
            const uint32_t i0 = (uint32_t)__shfl((int)i, lead, 64);
            const bool same = rem && i == i0;
            const uint32_t c = (uint32_t)__popcll(__ballot(same));
            if (c == 1) break;
            if (lane == lead) atomicAdd(&cnt[bfs_cnt_at(i0, rows)], c);
            if (same) rem = false;
        }
        if (rem) atomicAdd(&cnt[bfs_cnt_at(i, rows)], 1u);
    }
}

// places every new node in its claimer's stretch of the next frontier (any order inside the stretch) and marks it found
__global__ void __launch_bounds__(kT)
bfs_place_kernel(const uint32_t *__restrict__ fresh, const uint32_t *__restrict__ n_fresh, const uint32_t *__restrict__ claim,
                 const uint32_t *__restrict__ frontier, const uint32_t *__restrict__ pos, uint32_t *__restrict__ cnt,
                 uint32_t *__restrict__ next, uint32_t *__restrict__ parent, uint32_t *__restrict__ depth,
                 uint32_t *__restrict__ vis, uint32_t next_depth, uint32_t rows) {
    const uint32_t n = *n_fresh;
    const int lane = threadIdx.x & 63;
    for (uint32_t k0 = blockIdx.x * blockDim.x; k0 < n; k0 += gridDim.x * blockDim.x) {  // (uniform over the wave)
        const uint32_t k = k0 + threadIdx.x;
        const uint32_t v = k < n ? fresh[k] : CZ_NONE;  // (CZ_NONE: a slot its wave did not use)
        const bool live = v != CZ_NONE;
        bool rem = live;
        const uint32_t i = live ? claim[v] : CZ_NONE;
        uint32_t slot = 0;
        for (int r = 0; r < kBfsAggRounds; r++) {  // lanes of one claimer take their slots with one atomic (bfs_tally_kernel)
            const unsigned long long mrem = __ballot(rem);
            if (!mrem) break;
            const int lead = __builtin_ctzll(mrem);
            const uint32_t i0 = (uint32_t)__shfl((int)i, lead, 64);
            const bool same = rem && i == i0;
            const unsigned long long ms = __ballot(same);
            const uint32_t c = (uint32_t)__popcll(ms);
            if (c == 1) break;
            uint32_t base = 0;
            if (lane == lead) base = atomicSub(&cnt[bfs_cnt_at(i0, rows)], c);
            base = (uint32_t)__shfl((int)base, lead, 64);
            if (same) {
                slot = pos[i0] + base - 1u - (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
                rem = false;
            }
        }
        if (rem) slot = pos[i] + atomicSub(&cnt[bfs_cnt_at(i, rows)], 1u) - 1u;
        if (!live) continue;
        next[slot] = v;
        parent[v] = frontier[i];
        depth[v] = next_depth;
        atomicOr(&vis[v >> 5], 1u << (v & 31));
    }
}

// every claimer's stretch ascending by node id = the order of its (sorted) adjacency list.  Short stretches: one thread,
// insertion sort; long ones are listed for bfs_order_big_kernel.
constexpr uint32_t kBfsSmallGroup = 24;
__global__ void __launch_bounds__(kT)
bfs_order_small_kernel(const uint32_t *__restrict__ pos, uint32_t fsize, const uint32_t *__restrict__ total,
                       uint32_t *__restrict__ next, uint32_t *__restrict__ big, uint32_t *__restrict__ n_big) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < fsize; i += gridDim.x * blockDim.x) {
        const uint32_t a = pos[i], z = i + 1 < fsize ? pos[i + 1] : *total;
        const uint32_t len = z - a;
        if (len < 2) continue;
        if (len > kBfsSmallGroup) {
            big[atomicAdd(n_big, 1u)] = i;
            continue;
        }
        uint32_t x[kBfsSmallGroup];
#pragma unroll
        for (uint32_t k = 0; k < kBfsSmallGroup; k++) x[k] = k < len ? next[a + k] : CZ_NONE;
#pragma unroll
        for (uint32_t k = 1; k < kBfsSmallGroup; k++) {  // a fixed network over registers (padding sorts last)
#pragma unroll
            for (uint32_t j = k; j > 0; j--) {
                const uint32_t lo = min(x[j - 1], x[j]), hi = max(x[j - 1], x[j]);
                x[j - 1] = lo;
                x[j] = hi;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kBfsSmallGroup; k++)
            if (k < len) next[a + k] = x[k];
    }
}

// a wave per long stretch: the claimer's adjacency list is walked again and its new nodes written in list order.  Round 6: a
// claimer whose list holds kBfsOrderWg edges or more (an R-MAT hub: 100 000 edges walked 64 at a time by ONE wave, each step a
// chain tgt -> claim -> depth, was 1.8 ms of a level and 3.7 ms of a 12 ms run) is walked by the whole WORKGROUP, four edges per
// thread and step (1 024 edges in flight), the running offset carried through LDS.  Workgroups take the listed stretches four at
// a time: each wave its own when the list is short, then all four waves together each long one among the four.
constexpr uint32_t kBfsOrderWg = 1024;
constexpr int kBfsOrderK = 4;
__global__ void __launch_bounds__(kT)
bfs_order_big_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                     const uint32_t *__restrict__ pos, const uint32_t *__restrict__ big, const uint32_t *__restrict__ n_big,
                     const uint32_t *__restrict__ claim, const uint32_t *__restrict__ depth, uint32_t next_depth,
                     uint32_t *__restrict__ next) {
    constexpr uint32_t kWaves = kT / 64;
    const int lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t n = *n_big;
    __shared__ uint32_t sh_i[kWaves], sh_e0[kWaves], sh_e1[kWaves], sh_o[kWaves], sh_cnt[kBfsOrderK][kWaves];
    for (uint32_t k0 = blockIdx.x * kWaves; k0 < n; k0 += gridDim.x * kWaves) {  // (uniform over the workgroup)
        const uint32_t k = k0 + wv;
        uint32_t i = 0, e0 = 0, e1 = 0, o = 0;
        if (k < n) {
            i = big[k];
            const uint32_t u = frontier[i];
            e0 = off[u];
            e1 = off[u + 1];
            o = pos[i];
        }
        const bool together = e1 - e0 >= kBfsOrderWg;
        if (lane == 0) {
            sh_i[wv] = i;
            sh_e0[wv] = e0;
            sh_e1[wv] = together ? e1 : e0;  // (an empty range: nothing for the workgroup to do)
            sh_o[wv] = o;
        }
        if (!together) {
            for (uint32_t b = e0; b < e1; b += 64) {
                const uint32_t e = b + lane;
                bool hit = false;
                uint32_t v = 0;
                if (e < e1) {
                    v = tgt[e];
                    hit = (e == e0 || tgt[e - 1] != v) && claim[v] == i && depth[v] == next_depth;
                }
                const unsigned long long m = __ballot(hit);
                if (hit) next[o + __popcll(m & ((1ull << lane) - 1ull))] = v;
                o += (uint32_t)__popcll(m);
            }
        }
        __syncthreads();
        for (uint32_t w = 0; w < kWaves; w++) {
            const uint32_t ci = sh_i[w], c0 = sh_e0[w], c1 = sh_e1[w];
            uint32_t co = sh_o[w];
            for (uint32_t b = c0; b < c1; b += kT * kBfsOrderK) {  // (uniform over the workgroup)
                uint32_t v[kBfsOrderK];
                bool hit[kBfsOrderK];
                unsigned long long m[kBfsOrderK];
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) {
                    const uint32_t e = b + j * kT + threadIdx.x;
                    v[j] = e < c1 ? tgt[e] : CZ_NONE;
                    hit[j] = e < c1 && (e == c0 || tgt[e - 1] != v[j]);
                }
                uint32_t cl[kBfsOrderK], dp[kBfsOrderK];  // (unconditional loads: the four chains go out together)
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) cl[j] = claim[hit[j] ? v[j] : 0u];
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) dp[j] = depth[hit[j] ? v[j] : 0u];
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) hit[j] = hit[j] && cl[j] == ci && dp[j] == next_depth;
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) {
                    m[j] = __ballot(hit[j]);
                    if (lane == 0) sh_cnt[j][wv] = (uint32_t)__popcll(m[j]);
                }
                __syncthreads();
                uint32_t run = co;  // edges in order: (j, wave, lane)
#pragma unroll
                for (int j = 0; j < kBfsOrderK; j++) {
                    uint32_t before = 0, all = 0;
#pragma unroll
                    for (uint32_t q = 0; q < kWaves; q++) {
                        const uint32_t c = sh_cnt[j][q];
                        before += q < wv ? c : 0u;
                        all += c;
                    }
                    if (hit[j]) next[run + before + __popcll(m[j] & ((1ull << lane) - 1ull))] = v[j];
                    run += all;
                }
                co = run;
                __syncthreads();
            }
        }
        __syncthreads();  // (sh_* are rewritten by the next four)
    }
}

__global__ void bfs_goals_left_kernel(const uint32_t *__restrict__ goals, uint32_t n_goals, uint32_t N,
                                      const uint32_t *__restrict__ depth, uint32_t start, uint32_t *__restrict__ left) {
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_goals; i += gridDim.x * blockDim.x) {
        const uint32_t g = goals[i];
        // the start itself never gets a backtrace entry: it stays pending for ever (shortest_path_bfs.rs:71-77)
        if (g < N && (depth[g] == CZ_NONE || g == start)) c++;
    }
    if (c) atomicAdd(left, c);
}

__global__ void set_u32_kernel(uint32_t *p, uint32_t idx, uint32_t v) { p[idx] = v; }
// every target a node id?  (the O(E) part of the adjacency checks that stays when the caller vouches for symmetry: ADVICE r4 --
// a vouched adjacency with a target out of range was read out of bounds)
__global__ void targets_in_range_kernel(const uint32_t *__restrict__ tgt, uint64_t E, uint32_t N, uint32_t *__restrict__ bad) {
    bool any = false;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (uint64_t)gridDim.x * blockDim.x) any |= tgt[e] >= N;
    if (any) atomicAdd(bad, 1u);
}
__global__ void scatter_u32_kernel(uint32_t *__restrict__ p, const uint32_t *__restrict__ idx, uint32_t n, uint32_t v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[idx[i]] = v;
}
__global__ void or_u32_kernel(uint32_t *p, uint32_t idx, uint32_t v) { p[idx] |= v; }

// ---------------------------------------------------------------------------------------------
// connected components
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kT) iota_kernel(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

// Union-find with sampling (the "Afforest" scheme): comp[] is a forest whose links always point from a higher to a lower
// index, so the root of every tree is the SMALLEST index of its component -- the label the group numbering needs, whatever
// the order in which links happen.
//   1. every node links with its first two neighbours (2 x N links instead of E), trees are compressed;
//   2. a sample of the nodes names the component most of them are in (on a graph with a giant component: that one);
//   3. nodes OUTSIDE that component link with the rest of their neighbours -- the nodes inside skip their lists: an edge
//      leaving the big component is seen from its other end, the adjacency being symmetric (as_directed_graph(undirected =
//      true), which is what the rule passes; header);
//   4. compress.
// On the symmetrised 10M / 200M uniform graph step 3 skips all but a handful of lists: the rule reads ~2 N adjacency
// entries, not E.  Round 1/2's min-label propagation (one atomicMin per edge per round) took 9.3 + 3.4 ms there.
constexpr int kCcLanes = 16;
constexpr uint32_t kCcNeighbourRounds = 2, kCcSamples = 1024;

__device__ __forceinline__ uint32_t cc_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // other lanes' links must become visible
}

__device__ __forceinline__ void cc_link(uint32_t u, uint32_t v, uint32_t *__restrict__ comp) {
    uint32_t p1 = cc_load(&comp[u]), p2 = cc_load(&comp[v]);
    while (p1 != p2) {
        const uint32_t high = p1 > p2 ? p1 : p2, low = p1 > p2 ? p2 : p1;
        const uint32_t ph = cc_load(&comp[high]);
        if (ph == low) break;                                                 // already hooked there
        if (ph == high && atomicCAS(&comp[high], high, low) == high) break;   // high was a root: hook it under low
        p1 = cc_load(&comp[cc_load(&comp[high])]);                            // somebody moved it: climb and retry
        p2 = cc_load(&comp[low]);
    }
}

__global__ void __launch_bounds__(kT)
cc_link_nth_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, uint32_t nth, uint32_t *__restrict__ comp) {
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < N; u += gridDim.x * blockDim.x) {
        const uint32_t e = off[u] + nth;
        if (e < off[u + 1]) cc_link(u, tgt[e], comp);
    }
}

// pointer jumping in place: every node keeps replacing its pointer by its grandparent's until it points at a root; the
// lanes shorten each other's paths as they go, so a chain of length L costs O(log L) steps per node, not L
__global__ void __launch_bounds__(kT) cc_compress_kernel(uint32_t N, uint32_t *__restrict__ comp) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        uint32_t p = cc_load(&comp[i]);
        for (;;) {
            const uint32_t pp = cc_load(&comp[p]);
            if (pp == p) break;
            __hip_atomic_store(&comp[i], pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = pp;
        }
    }
}

__global__ void __launch_bounds__(kT)
cc_sample_kernel(uint32_t N, const uint32_t *__restrict__ comp, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kCcSamples) out[i] = comp[(uint32_t)(((uint64_t)i * 2654435761ull + 12345ull) % N)];
}

// step 3: a group of kCcLanes lanes per node outside `skip`, the rest of its list read coalesced
__global__ void __launch_bounds__(kT)
cc_link_rest_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, uint32_t first, uint32_t skip,
                    uint32_t *__restrict__ comp) {
    const uint32_t glane = threadIdx.x & (kCcLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kCcLanes, ngroups = gridDim.x * blockDim.x / kCcLanes;
    for (uint32_t u = group; u < N; u += ngroups) {
        if (cc_load(&comp[u]) == skip) continue;
        const uint32_t e1 = off[u + 1];
        for (uint32_t e = off[u] + first + glane; e < e1; e += kCcLanes) cc_link(u, tgt[e], comp);
    }
}

__global__ void __launch_bounds__(kT)
cc_rootflag_kernel(uint32_t N, const uint32_t *__restrict__ label, uint32_t *__restrict__ flag) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) flag[i] = label[i] == i;
}

__global__ void __launch_bounds__(kT)
cc_group_kernel(uint32_t N, const uint32_t *__restrict__ label, const uint32_t *__restrict__ rank, uint32_t *__restrict__ grp) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) grp[i] = rank[label[i]];
}

// ---------------------------------------------------------------------------------------------
// SSSP
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kInfPacked = (0x7F800000ull << 32) | 0xFFFFFFFFull;

__global__ void __launch_bounds__(kT) fill_u64_kernel(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

constexpr int kSsspLanes = 16;

// one queue: entries (source index << 32 | node) -- or plain node ids --, a device counter
template <class T>
struct QueueT {
    T *items;
    uint32_t *count;
};
using SsspQueue = QueueT<unsigned long long>;

// wave-aggregated push: one atomicAdd per wave instruction and queue
template <class T>
__device__ __forceinline__ void sssp_push(const QueueT<T> &q, bool want, T item, int lane) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    uint32_t base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(q.count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (want) q.items[base + __popcll(m & ((1ull << lane) - 1ull))] = item;
}

// One counter per pile would see one returning atomicAdd per wave instruction from every wave of the grid -- and same-address
// atomics are served one after another: a round that pushes 8M entries spent 15.7 of its 16.6 ms there
// (profiles/r02_sssp_push_experiment.txt).  So pushes are staged per WORKGROUP: a wave reserves slots in an LDS buffer with an
// LDS atomic, and every few iterations the workgroup moves what has gathered to the pile with ONE global atomicAdd and a
// coalesced copy.  Entries that do not fit the buffer (a hub's list) take the wave-level path above.
constexpr uint32_t kStageCap = 1024;

template <class T>
struct StagedPileT {
    T buf[kStageCap];
    uint32_t count, base;
};
using StagedPile = StagedPileT<unsigned long long>;

template <class T>
__device__ __forceinline__ void staged_push(const QueueT<T> &q, StagedPileT<T> &st, bool want, T item, int lane) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    uint32_t base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&st.count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    const uint32_t slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (want && slot < kStageCap) st.buf[slot] = item;
    sssp_push(q, want && slot >= kStageCap, item, lane);
}

// every thread of the workgroup, in uniform control flow
template <class T>
__device__ __forceinline__ void staged_flush(const QueueT<T> &q, StagedPileT<T> &st) {
    __syncthreads();
    const uint32_t n = min(st.count, kStageCap);
    if (threadIdx.x == 0 && n) st.base = atomicAdd(q.count, n);
    __syncthreads();
    const uint32_t gb = st.base;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) q.items[gb + i] = st.buf[i];
    __syncthreads();
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
}

// hubs: a 16-lane group walks at most kSsspLongList edges of its node.  What is left of a longer list goes, in stretches of
// kSsspStretch edges, to a queue in global memory that sssp_relax_long_kernel walks with the whole GRID right after the round
// (SsspLongQ; the queue holds S x cz_graph::long_stretches items -- every (source, node) pair is in a pile once per round).  Round 6:
// the remainder used to be walked by the node's own WORKGROUP at the end of the round -- enough to take a 100 000-edge hub off one
// 16-lane group (R-MAT 39 -> 12 ms), but the first rounds of an R-MAT run are nothing but hubs: one workgroup walked 106 000
// edges alone (1.2 ms), the next rounds waited for the workgroups that had drawn the hubs (2.1 + 2.9 + 2.9 ms of a 12 ms run).
// Relaxations commute (CAS on the packed word), so who walks an edge changes nothing.  The group's own share: 256 edges (a workgroup
// iteration lasts as long as its longest list: R-MAT 10M / 100M 10.4 ms with 1 024, 10.1 with 256, 13.3 with 128 -- a stretch item
// costs a workgroup a pass and two flushes whatever it holds).
#ifndef CZ_SSSP_LONG_LIST
#define CZ_SSSP_LONG_LIST 256
#endif
constexpr uint32_t kSsspLongList = CZ_SSSP_LONG_LIST, kSsspStretch = 2048;
struct SsspLongQ {
    uint32_t *items;   // [5][cap]: source index, node, first edge, end edge, the node's cost bits -- one item per stretch
    uint32_t cap;
    uint32_t *count;   // this round's items (nullptr: no queue -- the workgroup walks its own long lists)
    uint32_t *count_zero;  // the next round's counter
};
struct SsspRelaxCtx {
    const uint32_t *tgt;
    const float *w;
    uint32_t N;
    unsigned long long *dp;
    uint32_t *qtag, *ftag;
    uint32_t round_tag, phase_tag, thr_bits;
};
// one edge: the target's packed word replaced when the offer is better; where the target goes next (see the rule below)
__device__ __forceinline__ void sssp_relax_edge(const SsspRelaxCtx &c, uint32_t si, uint32_t u, float du, uint32_t e, bool on, bool &to_near,
                                                bool &to_far, uint32_t &v) {
    to_near = to_far = false;
    v = 0;
    if (!on) return;
    unsigned long long *dps = c.dp + (size_t)si * c.N;
    v = c.tgt[e];
    const float nd = du + c.w[e];  // `cost + path_weight` in f32 (shortest_path_dijkstra.rs:303)
    const uint32_t nb = __float_as_uint(nd);
    const bool proper = nb != __float_as_uint(du);
    const unsigned long long want = ((unsigned long long)nb << 32) | (proper ? 0u : 0x80000000u) | u;
    unsigned long long seen = __hip_atomic_load(&dps[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const bool lower = nb < (uint32_t)(seen >> 32);  // strict `<` (:304); non-negative floats order as their bits
        if (!lower && !(proper && nb == (uint32_t)(seen >> 32) && want < seen)) break;
        const unsigned long long got = atomicCAS(&dps[v], seen, want);
        if (got == seen) {
            if (lower) {  // (an equal-cost change of parent is nothing the node's own edges need to hear about)
                const size_t at = (size_t)si * c.N + v;
                if (nb < c.thr_bits) to_near = atomicExch(&c.qtag[at], c.round_tag) != c.round_tag;
                else to_far = atomicExch(&c.ftag[at], c.phase_tag) != c.phase_tag;
            }
            break;
        }
        seen = got;
    }
}

// relax the out-edges of every (source, node) entry of `cur`: a 16-lane group per entry reads the adjacency coalesced.
// An improved target goes to `near` when its new cost is below the threshold, else to `far`; `qtag` / `ftag` (one word
// per (source, node)) keep a pair from entering the same pile twice in one round / one phase.
// The packed word of a (source, node) pair is (cost bits << 32 | improper << 31 | parent): `improper` marks a parent whose
// own cost equals the node's (a zero-weight edge, or a weight the f32 sum absorbs).  A relaxation replaces the word when it
// lowers the cost (`cost + path_weight < seen`, shortest_path_dijkstra.rs:303-304) -- or, at EQUAL cost, when it comes from a
// proper predecessor and the packed word gets smaller: every tight predecessor offers the final cost once, so the parent ends
// up as the SMALLEST tight predecessor of strictly smaller cost whatever the schedule was (two runs return the same rows;
// the reference's own choice among equal-cost predecessors is its heap's pop order).  A node all of whose tight predecessors
// sit at its own cost keeps the one that lowered its cost: that pointer is acyclic by construction, a smallest-id rule there
// could close a cycle.  Node ids stay below 2^31 for the flag bit.
__global__ void __launch_bounds__(kT)
sssp_relax_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t N,
                  const unsigned long long *__restrict__ cur, uint32_t n_cur, unsigned long long *__restrict__ dp,
                  uint32_t *__restrict__ qtag, uint32_t round_tag, uint32_t *__restrict__ ftag, uint32_t phase_tag,
                  uint32_t thr_bits, SsspQueue near, SsspQueue far, uint32_t *__restrict__ zero_me,
                  const uint32_t *__restrict__ n_cur_dev, uint32_t *__restrict__ bail, SsspLongQ lq) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    // A round launched AHEAD of the host's knowledge of its pile (SsspBatch::run, bursts of small rounds) reads the pile's size
    // where the round before it counted it; zero_me is then a THIRD counter (nobody's input, nobody's output in this round).
    // Its grid was sized on a guess: a pile that outgrew it (an R-MAT hub two rounds behind the start: 1 -> 53 000 -> 1.1M
    // entries) is NOT walked here with a few workgroups -- the round leaves everything as it found it, says so in bail[0..1]
    // (its tag, the pile's size), the rounds launched behind it return at once, and the host launches it again, sized.
    if (n_cur_dev) {
        if (bail[0] != 0) return;  // (whole workgroups return: the decision is the same for every thread of the grid or harmless)
        n_cur = *n_cur_dev;
        if (n_cur > ngroups) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                bail[1] = n_cur;
                bail[0] = round_tag;
            }
            return;
        }
    }
    const uint32_t rounds = (n_cur + ngroups - 1) / ngroups;  // every group of the GRID runs the same trip count (ballots, barriers)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *zero_me = 0;  // the NEXT round's near counter (this round appends to the other one)
        if (lq.count) *lq.count_zero = 0;
    }
    __shared__ StagedPile st_near, st_far;
    // (the long lists' remainders when there is no queue, or no room in it: set aside here, walked by the whole workgroup)
    constexpr uint32_t kGroupsPerWg = kT / kSsspLanes;
    __shared__ uint32_t lq_si[kGroupsPerWg], lq_u[kGroupsPerWg], lq_e0[kGroupsPerWg], lq_e1[kGroupsPerWg], lq_du[kGroupsPerWg], lq_n;
    if (threadIdx.x == 0) {
        st_near.count = st_far.count = 0;
        lq_n = 0;
    }
    __syncthreads();
    const SsspRelaxCtx ctx{tgt, w, N, dp, qtag, ftag, round_tag, phase_tag, thr_bits};
    auto relax = [&](uint32_t si, uint32_t u, float du, uint32_t e, bool on, bool &to_near, bool &to_far, uint32_t &v) {
        sssp_relax_edge(ctx, si, u, du, e, on, to_near, to_far, v);
    };
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < n_cur;
        const unsigned long long ent = live ? cur[i] : 0ull;
        const uint32_t si = (uint32_t)(ent >> 32), u = (uint32_t)ent;
        unsigned long long *dps = dp + (size_t)si * N;
        const uint32_t e0 = live ? off[u] : 0;
        uint32_t e1 = live ? off[u + 1] : 0;
        float du = 0.f;
        if (live) du = __uint_as_float((uint32_t)(__hip_atomic_load(&dps[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32));
        if (e1 - e0 > kSsspLongList) {  // (uniform over the group)
            const uint32_t r0 = e0 + kSsspLongList, nst = (e1 - r0 + kSsspStretch - 1) / kSsspStretch;
            bool queued = false;
            if (lq.count) {  // the remainder as stretches for sssp_relax_long_kernel: one reservation, the group's lanes fill it
                uint32_t base = 0;
                if (glane == 0) base = atomicAdd(lq.count, nst);
                base = (uint32_t)__shfl((int)base, lane & ~(kSsspLanes - 1), 64);
                queued = base + nst <= lq.cap && base + nst >= base;
                for (uint32_t t = glane; t < nst && base + t < lq.cap; t += kSsspLanes) {  // (no room: empty items, the list stays here)
                    const uint32_t q = base + t, a = r0 + t * kSsspStretch;
                    lq.items[q] = si;
                    lq.items[lq.cap + q] = u;
                    lq.items[2 * (size_t)lq.cap + q] = queued ? a : 0u;
                    lq.items[3 * (size_t)lq.cap + q] = queued ? min(e1, a + kSsspStretch) : 0u;
                    lq.items[4 * (size_t)lq.cap + q] = __float_as_uint(du);
                }
            }
            if (!queued && glane == 0) {
                const uint32_t q = atomicAdd(&lq_n, 1u);
                lq_si[q] = si;
                lq_u[q] = u;
                lq_e0[q] = r0;
                lq_e1[q] = e1;
                lq_du[q] = __float_as_uint(du);
            }
            e1 = e0 + kSsspLongList;
        }
        uint32_t maxlen = e1 - e0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
        for (uint32_t b = 0; b < maxlen; b += kSsspLanes) {
            const uint32_t e = e0 + b + glane;
            bool to_near, to_far;
            uint32_t v;
            relax(si, u, du, e, e < e1, to_near, to_far, v);
            const unsigned long long item = ((unsigned long long)si << 32) | v;
            staged_push(near, st_near, to_near, item, lane);
            staged_push(far, st_far, to_far, item, lane);
        }
        __syncthreads();
        const uint32_t nq = lq_n;  // (uniform) the long lists' remainders: every lane of the workgroup takes edges
        for (uint32_t q = 0; q < nq; q++) {
            const uint32_t qsi = lq_si[q], qu = lq_u[q], q0 = lq_e0[q], q1 = lq_e1[q];
            const float qdu = __uint_as_float(lq_du[q]);
            for (uint32_t b = q0; b < q1; b += kT) {
                const uint32_t e = b + threadIdx.x;
                bool to_near, to_far;
                uint32_t v;
                relax(qsi, qu, qdu, e, e < q1, to_near, to_far, v);
                const unsigned long long item = ((unsigned long long)qsi << 32) | v;
                staged_push(near, st_near, to_near, item, lane);
                staged_push(far, st_far, to_far, item, lane);
                if (((b - q0) / kT & 7u) == 7u) {  // (uniform) the staging pile is a few hundred entries
                    staged_flush(near, st_near);
                    staged_flush(far, st_far);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) lq_n = 0;
        __syncthreads();  // (the next round's groups add to it)
        if (nq || (r & 7) == 7 || r + 1 == rounds) {  // 16 entries per iteration and workgroup: a few hundred pushes gather in 8
            staged_flush(near, st_near);
            staged_flush(far, st_far);
        }
    }
}

// the long lists' remainders of the round that has just run (SsspLongQ): stretch i by workgroup i mod the grid, every thread an edge
__global__ void __launch_bounds__(kT)
sssp_relax_long_kernel(const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t N, unsigned long long *__restrict__ dp,
                       uint32_t *__restrict__ qtag, uint32_t round_tag, uint32_t *__restrict__ ftag, uint32_t phase_tag, uint32_t thr_bits,
                       SsspQueue near, SsspQueue far, SsspLongQ lq, const uint32_t *__restrict__ bail) {
    if (bail[0] != 0) return;  // (a launched-ahead round left its pile to the host: nothing was queued, and nothing may be touched)
    const uint32_t n = min(*lq.count, lq.cap);
    if (n == 0) return;
    const int lane = threadIdx.x & 63;
    __shared__ StagedPile st_near, st_far;
    if (threadIdx.x == 0) st_near.count = st_far.count = 0;
    __syncthreads();
    const SsspRelaxCtx ctx{tgt, w, N, dp, qtag, ftag, round_tag, phase_tag, thr_bits};
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {  // (uniform over the workgroup)
        const uint32_t si = lq.items[i], u = lq.items[lq.cap + i], q0 = lq.items[2 * (size_t)lq.cap + i], q1 = lq.items[3 * (size_t)lq.cap + i];
        const float du = __uint_as_float(lq.items[4 * (size_t)lq.cap + i]);
        for (uint32_t b = q0; b < q1; b += kT) {
            const uint32_t e = b + threadIdx.x;
            bool to_near, to_far;
            uint32_t v;
            sssp_relax_edge(ctx, si, u, du, e, e < q1, to_near, to_far, v);
            const unsigned long long item = ((unsigned long long)si << 32) | v;
            staged_push(near, st_near, to_near, item, lane);
            staged_push(far, st_far, to_far, item, lane);
            if (((b - q0) / kT & 3u) == 3u) {  // (uniform) the staging pile holds 1 024 entries: four iterations' worth
                staged_flush(near, st_near);
                staged_flush(far, st_far);
            }
        }
        staged_flush(near, st_near);
        staged_flush(far, st_far);
    }
}

// how many stretches the long lists of a graph make (cz_graph::long_stretches: the capacity of SsspLongQ per source)
__global__ void __launch_bounds__(kT) sssp_long_stretches_kernel(const uint32_t *__restrict__ off, uint32_t N, unsigned long long *__restrict__ out) {
    unsigned long long mine = 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        const uint32_t len = off[v + 1] - off[v];
        if (len > kSsspLongList) mine += (len - kSsspLongList + kSsspStretch - 1) / kSsspStretch;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// the threshold moved: far entries whose CURRENT cost is below it become the next near pile, the rest stay far
__global__ void __launch_bounds__(kT)
sssp_split_kernel(const unsigned long long *__restrict__ farq, uint32_t n_far, uint32_t N,
                  const unsigned long long *__restrict__ dp, uint32_t *__restrict__ qtag, uint32_t round_tag,
                  uint32_t *__restrict__ ftag, uint32_t phase_tag, uint32_t thr_bits, SsspQueue near, SsspQueue far_next,
                  const uint32_t *__restrict__ thr_dev) {
    if (thr_dev) thr_bits = *thr_dev;  // (the threshold sssp_threshold_kernel worked out: no host round trip between the two)
    const int lane = threadIdx.x & 63;
    const uint32_t total = gridDim.x * blockDim.x;
    const uint32_t rounds = (n_far + total - 1) / total;
    __shared__ StagedPile st_near, st_far;
    if (threadIdx.x == 0) st_near.count = st_far.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + r * total;
        bool to_near = false, to_far = false;
        unsigned long long ent = 0;
        if (i < n_far) {
            ent = farq[i];
            const size_t at = (size_t)(ent >> 32) * N + (uint32_t)ent;
            const uint32_t cb = (uint32_t)(dp[at] >> 32);
            if (cb < thr_bits) to_near = atomicExch(&qtag[at], round_tag) != round_tag;
            else to_far = atomicExch(&ftag[at], phase_tag) != phase_tag;
        }
        staged_push(near, st_near, to_near, ent, lane);
        staged_push(far_next, st_far, to_far, ent, lane);
        if ((r & 1) == 1 || r + 1 == rounds) {  // at most one entry per thread and iteration: two iterations fit the buffer
            staged_flush(near, st_near);
            staged_flush(far_next, st_far);
        }
    }
}

// smallest cost bits among the far entries (where the next threshold has to reach)
__global__ void __launch_bounds__(kT)
sssp_far_min_kernel(const unsigned long long *__restrict__ farq, uint32_t n_far, uint32_t N,
                    const unsigned long long *__restrict__ dp, uint32_t *__restrict__ out_min) {
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_far; i += gridDim.x * blockDim.x) {
        const unsigned long long ent = farq[i];
        m = min(m, (uint32_t)(dp[(size_t)(ent >> 32) * N + (uint32_t)ent] >> 32));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m != 0xFFFFFFFFu) atomicMin(out_min, m);
}

// The threshold move of the near-far schedule on the device (one thread): the bucket of the nearest waiting node, exactly the
// host expression it replaces -- only the schedule depends on it, never a result.  misc: SsspBatch's counter block; the counters
// the split and the rounds after it count into start from zero.
__global__ void sssp_threshold_kernel(uint32_t *__restrict__ misc, float thr_old, float delta, int have_min) {
    // have_min == 0: the far pile was not searched for its nearest node (SsspBatch::run: a pile of millions has one in the next
    // bucket) -- the threshold moves by one bucket
    const float fmin = have_min ? __uint_as_float(misc[3]) : thr_old;
    float thr = fmaxf(thr_old + delta, (floorf(fmin / delta) + 1.0f) * delta);
    if (!(thr > fmin)) thr = INFINITY;  // (rounding at huge costs, or no waiting node with a cost: fall back to one pile)
    misc[7] = __float_as_uint(thr);
    misc[0] = 0;
    misc[2] = 0;
    misc[8] = misc[9] = misc[10] = 0;
    misc[13] = misc[14] = 0;  // (the long-list queue's counters, by round parity)
}
// after the split: the far counter the relax kernel appends to continues from the surviving entries
__global__ void sssp_far_carry_kernel(uint32_t *__restrict__ misc) { misc[1] = misc[2]; }

__global__ void __launch_bounds__(kT)
sssp_seed_kernel(const uint32_t *__restrict__ starts, uint32_t n, uint32_t N, unsigned long long *__restrict__ dp,
                 unsigned long long *__restrict__ q, uint32_t *__restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t st = starts[i];
    if (st >= N) return;
    dp[(size_t)i * N + st] = 0x00000000FFFFFFFFull;  // cost 0.0, no parent
    q[atomicAdd(count, 1u)] = ((unsigned long long)i << 32) | st;
}

// Which tight predecessor ends up as a node's parent depends on which CAS came first.  The costs do not, so the parents
// are made canonical afterwards: among the predecessors u with dist[u] + w == dist[v] and dist[u] < dist[v] the smallest
// id wins (one pass over the edges per source).  A node whose tight predecessors all sit at its own cost (zero-weight or
// absorbed edges) keeps the parent the relaxation found: that one is acyclic by construction, a smallest-id rule there
// could close a cycle.  (The reference's own choice among equal-cost predecessors is its heap's pop order.)
__global__ void __launch_bounds__(kT)
sssp_canon_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t N,
                  uint32_t n_src, const unsigned long long *__restrict__ dp, uint32_t *__restrict__ canon, uint32_t rb,
                  uint32_t re) {
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes;
    const uint64_t ngroups = (uint64_t)gridDim.x * blockDim.x / kSsspLanes, total = (uint64_t)n_src * N;
    for (uint64_t i = group; i < total; i += ngroups) {
        const uint32_t si = (uint32_t)(i / N), u = (uint32_t)(i % N);
        if (u < rb || u >= re) continue;  // (a vertex-partitioned graph: this device holds the adjacency of [rb, re) only)
        const unsigned long long *dps = dp + (size_t)si * N;
        const uint32_t cu = (uint32_t)(dps[u] >> 32);
        if (cu == 0x7F800000u) continue;  // unreached
        const float du = __uint_as_float(cu);
        const uint32_t e1 = off[u - rb + 1];
        for (uint32_t e = off[u - rb] + glane; e < e1; e += kSsspLanes) {
            const uint32_t v = tgt[e];
            const uint32_t cv = (uint32_t)(dps[v] >> 32);
            if (cu < cv && __float_as_uint(du + w[e]) == cv) atomicMin(&canon[(size_t)si * N + v], u);
        }
    }
}

__global__ void __launch_bounds__(kT)
sssp_unpack_kernel(const unsigned long long *__restrict__ dp, const uint32_t *__restrict__ canon, uint64_t n,
                   float *__restrict__ dist, uint32_t *__restrict__ parent) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = dp[i];
        dist[i] = __uint_as_float((uint32_t)(c >> 32));
        const uint32_t cp = canon[i];
        parent[i] = cp != CZ_NONE ? cp : (uint32_t)c;
    }
}

// the single-GPU words: (cost << 32 | improper << 31 | parent), 0xFFFFFFFF = no parent
// settled_bits: a run that stopped when its goals were settled (cz_sssp_goals) has final costs below that threshold only; the
// rest are reported unreached (costs are non-negative floats: their bit patterns order like the values).  +inf = a full run.
__global__ void __launch_bounds__(kT)
sssp_unpack_flagged_kernel(const unsigned long long *__restrict__ dp, uint64_t n, float *__restrict__ dist, uint32_t *__restrict__ parent,
                           uint32_t settled_bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = dp[i];
        const uint32_t cb = (uint32_t)(c >> 32);
        const bool final_cost = cb < settled_bits;
        dist[i] = final_cost ? __uint_as_float(cb) : __uint_as_float(0x7F800000u);
        const uint32_t p = (uint32_t)c;
        parent[i] = (p == CZ_NONE || !final_cost) ? CZ_NONE : (p & 0x7FFFFFFFu);
    }
}
// how many (source, goal) pairs are not settled yet: a goal is settled once its cost is below the threshold every remaining
// pile entry is at or above (the near pile is empty when this runs)
__global__ void __launch_bounds__(kT)
sssp_goals_left_kernel(const unsigned long long *__restrict__ dp, uint32_t N, uint32_t ns, const uint32_t *__restrict__ goals,
                       uint32_t n_goals, uint32_t thr_bits, uint32_t *__restrict__ left) {
    const uint64_t total = (uint64_t)ns * n_goals;
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t si = (uint32_t)(i / n_goals), g = goals[i % n_goals];
        // a goal that is no node is never settled: the reference's goal set is then never exhausted and the search runs to the end
        // (shortest_path_dijkstra.rs:296-300) -- skipping it here stopped the search after the first bucket
        if (g >= N || (uint32_t)(dp[(size_t)si * N + g] >> 32) >= thr_bits) mine++;
    }
    if (mine) atomicAdd(left, mine);
}

// BadEdgeWeightError (fixed_rule/mod.rs:258-286) on the device copy: the smallest index of a weight that is negative or NaN,
// and the sum of the weights (its mean is the bucket width of the near-far schedule)
__global__ void __launch_bounds__(kT)
weights_check_kernel(const float *__restrict__ w, uint64_t E, double *__restrict__ sum, unsigned long long *__restrict__ bad) {
    double acc = 0.0;
    unsigned long long first = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = w[i];
        if (!(x >= 0.0f)) first = min(first, (unsigned long long)i);
        else acc += x;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
    if (first != ~0ull) atomicMin(bad, first);
}

int check_csr(const uint32_t *off, const uint32_t *tgt, uint32_t N, uint64_t E) {
    if (N == 0) return CZ_OK;
    if (!off) return cz::set_error(CZ_E_INVALID, "null offsets");
    if (off[0] != 0 || off[N] != E) return cz::set_error(CZ_E_INVALID, "offsets[0] must be 0 and offsets[N] == E");
    if (E > 0 && !tgt) return cz::set_error(CZ_E_INVALID, "null targets");
    if (E >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "E must be < 2^32-1");
    return CZ_OK;
}

}  // namespace

// ---- a relation's CSR resident on the device (cz_graph_upload / cz_graph_cached): the *_on forms of the rules skip the upload --
struct cz_graph {
    cz::DevBuf<uint32_t> off, tgt;
    cz::DevBuf<float> w;
    uint32_t N = 0;
    uint64_t E = 0;
    bool has_w = false;
    double wsum = 0.0;                 // of the valid weights (the near-far bucket width is their mean)
    unsigned long long long_stretches = 0;  // of the out-lists beyond kSsspLongList entries (sssp_long_stretches_kernel; weighted graphs)
    unsigned long long bad = ~0ull;    // smallest index of a negative / NaN weight (BadEdgeWeightError when a rule needs them)
    float bad_value = 0.f;
    // the state arrays of the last cz_sssp_on call on this graph (0.64 GB per source batch at 10M nodes), kept for the next one: a
    // resident graph is what repeated calls run on, and they should not allocate at all.  A handle may be used from several
    // threads at once (cz_graph_upload hands it to whoever holds the pointer): the kept state belongs to the call that holds
    // sssp_mu; a call that finds it taken runs on arrays of its own, and only a state that went through a whole call is kept.
    mutable std::shared_ptr<void> sssp_state;
    mutable std::mutex sssp_mu;
};

namespace {

int graph_fill(cz_graph &G, const uint32_t *offsets, const uint32_t *targets, const float *weights, uint32_t N, uint64_t E) {
    int rc = check_csr(offsets, targets, N, E);
    if (rc) return rc;
    G.N = N;
    G.E = E;
    CZ_HIP(G.off.alloc((size_t)N + 1));
    CZ_HIP(G.tgt.alloc(E));
    CZ_HIP(hipMemcpy(G.off.p, offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) CZ_HIP(hipMemcpy(G.tgt.p, targets, E * 4, hipMemcpyHostToDevice));
    if (weights || E == 0) {
        G.has_w = true;
        CZ_HIP(G.w.alloc(E));
        if (E) {
            CZ_HIP(hipMemcpy(G.w.p, weights, E * 4, hipMemcpyHostToDevice));
            // negative / NaN weights (+inf is legal here: the reference checks the f64 value, and a finite f64 beyond f32's
            // range becomes +inf in its `as f32` cast; such an edge never improves anything -- inf < inf is false -- exactly
            // as in dijkstra :304); checked on the device copy, a host loop over 1e8 weights costs 50 ms
            cz::DevBuf<double> d_sum;
            cz::DevBuf<unsigned long long> d_bad;
            CZ_HIP(d_sum.alloc(1));
            CZ_HIP(d_bad.alloc(1));
            CZ_HIP(hipMemsetAsync(d_sum.p, 0, 8, nullptr));
            CZ_HIP(hipMemsetAsync(d_bad.p, 0xFF, 8, nullptr));
            hipLaunchKernelGGL(weights_check_kernel, dim3(grid_for(E)), dim3(kT), 0, nullptr, G.w.p, E, d_sum.p, d_bad.p);
            cz::DevBuf<unsigned long long> d_long;
            CZ_HIP(d_long.alloc(1));
            CZ_HIP(hipMemsetAsync(d_long.p, 0, 8, nullptr));
            hipLaunchKernelGGL(sssp_long_stretches_kernel, dim3(grid_for(N)), dim3(kT), 0, nullptr, G.off.p, N, d_long.p);
            CZ_HIP(hipMemcpy(&G.long_stretches, d_long.p, 8, hipMemcpyDeviceToHost));
            CZ_HIP(hipMemcpy(&G.bad, d_bad.p, 8, hipMemcpyDeviceToHost));
            CZ_HIP(hipMemcpy(&G.wsum, d_sum.p, 8, hipMemcpyDeviceToHost));
            if (G.bad != ~0ull) G.bad_value = weights[G.bad];
        }
    }
    return CZ_OK;
}

}  // namespace

extern "C" int cz_graph_upload(const uint32_t *offsets, const uint32_t *targets, const float *weights, uint32_t N, uint64_t E,
                               cz_graph **out) {
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    int rc = cz::ensure_device();
    if (rc) return rc;
    std::unique_ptr<cz_graph> g(new cz_graph);
    if ((rc = graph_fill(*g, offsets, targets, weights, N, E))) return rc;
    *out = g.release();
    return CZ_OK;
}

extern "C" void cz_graph_destroy(cz_graph *g) { delete g; }

// Resident graphs under the caller's (relation id, snapshot) key, like cz_pagerank_cached's plans: FixedRule::run is handed the
// relation anew on every call, and on the 10M / 100M graph the CSR upload is 8-15 ms of a 20-32 ms call.  An entry is taken OUT of
// the cache while a caller holds it (cz_graph_acquire) and comes back with cz_graph_release, so entries are never shared
// between threads; most recently used first, CZ_GRAPH_CACHE entries (default 4, 0 = nothing is kept).
namespace {
struct GraphCacheEntry {
    uint64_t hi, lo;
    std::unique_ptr<cz_graph> g;
};
std::mutex g_graph_mu;
std::list<GraphCacheEntry> g_graph_cache;
size_t graph_cache_capacity() {
    const char *e = getenv("CZ_GRAPH_CACHE");
    return e ? (size_t)std::max(0, atoi(e)) : 4;
}
}  // namespace

extern "C" int cz_graph_acquire(uint64_t key_hi, uint64_t key_lo, const uint32_t *offsets, const uint32_t *targets,
                                const float *weights, uint32_t N, uint64_t E, cz_graph **out, int *cache_hit) {
    if (cache_hit) *cache_hit = 0;
    if (!out) return cz::set_error(CZ_E_INVALID, "null out");
    *out = nullptr;
    int rc = cz::ensure_device();
    if (rc) return rc;
    if ((key_hi | key_lo) != 0) {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        for (auto it = g_graph_cache.begin(); it != g_graph_cache.end(); ++it)
            if (it->hi == key_hi && it->lo == key_lo && it->g->N == N && it->g->E == E && (it->g->has_w || !weights)) {
                *out = it->g.release();
                g_graph_cache.erase(it);
                if (cache_hit) *cache_hit = 1;
                return CZ_OK;
            }
    }
    return cz_graph_upload(offsets, targets, weights, N, E, out);
}

extern "C" void cz_graph_release(uint64_t key_hi, uint64_t key_lo, cz_graph *g) {
    if (!g) return;
    std::unique_ptr<cz_graph> own(g);
    if ((key_hi | key_lo) == 0 || graph_cache_capacity() == 0) return;
    std::lock_guard<std::mutex> lk(g_graph_mu);
    g_graph_cache.push_front(GraphCacheEntry{key_hi, key_lo, std::move(own)});
    while (g_graph_cache.size() > graph_cache_capacity()) g_graph_cache.pop_back();
}

extern "C" void cz_graph_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    g_graph_cache.clear();
}

extern "C" int cz_graph_last_timing(double *upload_ms, double *device_ms, double *download_ms) {
    if (upload_ms) *upload_ms = t_timing.ms[T_UPLOAD];
    if (device_ms) *device_ms = t_timing.ms[T_DEVICE];
    if (download_ms) *download_ms = t_timing.ms[T_DOWNLOAD];
    return CZ_OK;
}

namespace {

// the rule on a resident graph (cz_bfs uploads one for the call, cz_bfs_on is handed one)
// merged (cz_bfs_shared; share_visited only): ONE backtrace `parent` [N] and ONE discovery sequence `order` [N] for all starts --
// order[n_reached[i] .. n_reached[i + 1]) is what start i discovered, n_reached has n_starts + 1 entries -- instead of a row of N
// per start.  A start that an earlier one reached costs nothing (the host keeps the visited bits), a traversal resets only the
// claims of the nodes it reached: O(N + E) in all, like the reference's loop, whatever the number of starts.
int bfs_run(const cz_graph &G, const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, int share_visited,
            uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison, bool merged = false,
            cz_bfs_level_fn on_level = nullptr, void *on_level_ctx = nullptr) {
    const uint32_t N = G.N;
    const uint64_t E = G.E;
    int rc = CZ_OK;
    cz::DevBuf<uint32_t> d_depth, d_parent, d_claim, d_order, d_cnt, d_pos, d_scratch, d_goals, d_misc, d_vis, d_fresh, d_big, d_long;
    cz::DevBuf<uint8_t> d_won;
    const size_t vis_words = ((size_t)N + 31) / 32;
    // CZ_BFS_PASSES=3: the round-2 level (claim / count / emit over the edge slots), kept for A/B runs and as what the
    // vertex-partitioned loop (sharded_traversal.hpp) still runs
    const char *pv = getenv("CZ_BFS_PASSES");
    const bool one_pass = !(pv && atoi(pv) == 3);
    CZ_HIP(d_vis.alloc(vis_words));
    if (one_pass) {
        // every chunk but a wave's last is closed with fewer than 64 of its 256 slots unused; grid_for caps a launch at
        // 4096 workgroups of 4 waves
        // every chunk but a wave's last is closed with fewer than 64 of its 256 slots unused; two kernels list per level
        CZ_HIP(d_fresh.alloc(((size_t)N / (kBfsChunk - 63) + (4096 + 1024) * (kT / 64) + 2) * kBfsChunk));
        CZ_HIP(d_big.alloc(N));
        CZ_HIP(d_long.alloc((size_t)E / kBfsLongList + 2));  // at most this many lists are longer than kBfsLongList
    } else {
        CZ_HIP(d_won.alloc(E));
    }
    CZ_HIP(d_depth.alloc(N));
    CZ_HIP(d_parent.alloc(N));
    CZ_HIP(d_claim.alloc(N));
    CZ_HIP(d_order.alloc((size_t)N + 1));
    CZ_HIP(d_cnt.alloc((size_t)N + 32));  // (32 x ceil(fsize / 32) words: bfs_cnt_at)
    CZ_HIP(d_pos.alloc(N));
    CZ_HIP(d_scratch.alloc(scan_scratch_words(N)));
    CZ_HIP(d_misc.alloc(8));
    if (goals && n_goals) {
        CZ_HIP(d_goals.alloc(n_goals));
        CZ_HIP(hipMemcpy(d_goals.p, goals, (size_t)n_goals * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = nullptr;
    t_timing.lap(T_UPLOAD);
    CZ_HIP(hipMemsetAsync(d_depth.p, 0xFF, (size_t)N * 4, s));
    CZ_HIP(hipMemsetAsync(d_claim.p, 0xFF, (size_t)N * 4, s));
    CZ_HIP(hipMemsetAsync(d_vis.p, 0, vis_words * 4, s));
    std::vector<uint32_t> hvis;  // merged: the visited bits, kept on the host too
    uint64_t merged_total = 0;
    if (merged) {
        hvis.assign(vis_words, 0u);
        CZ_HIP(hipMemsetAsync(d_parent.p, 0xFF, (size_t)N * 4, s));
        n_reached[0] = 0;
    }
    bool stop = false;  // on_level said so: `found.len() >= limit` => break 'outer (algos/bfs.rs:88-91)
    for (uint32_t si = 0; si < n_starts; si++) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        const uint32_t start = starts[si];
        uint32_t reached = 0;
        if (merged) {
            if (stop || start >= N || (hvis[start >> 5] >> (start & 31)) & 1u) {  // algos/bfs.rs:52-54 already visited => skip
                n_reached[si + 1] = (uint32_t)merged_total;
                continue;
            }
            hvis[start >> 5] |= 1u << (start & 31);
        }
        if (!share_visited && si > 0) {
            CZ_HIP(hipMemsetAsync(d_depth.p, 0xFF, (size_t)N * 4, s));
            CZ_HIP(hipMemsetAsync(d_vis.p, 0, vis_words * 4, s));
        }
        // the claim words are per START even when `visited` is shared (algos/bfs.rs:43-53): bfs_order_big_kernel recognises a
        // level's new nodes by (claim == frontier position, depth == level + 1), and a node an EARLIER start reached at the same
        // position and depth would pass that test again -- written into the next frontier twice, the last new node pushed past
        // the stretch and never expanded.  Every claimed node is visited, so forgetting the claims loses nothing.
        if (si > 0 && !merged) CZ_HIP(hipMemsetAsync(d_claim.p, 0xFF, (size_t)N * 4, s));
        if (!merged) CZ_HIP(hipMemsetAsync(d_parent.p, 0xFF, (size_t)N * 4, s));
        bool run = start < N;
        if (run && share_visited && !merged) {
            uint32_t dstart;
            CZ_HIP(hipMemcpy(&dstart, d_depth.p + start, 4, hipMemcpyDeviceToHost));
            run = dstart == CZ_NONE;  // algos/bfs.rs:52-54 already visited => skip
        }
        if (run && goals && n_goals == 0) run = false;  // nothing pending: the reference discovers nothing useful
        if (run) {
            hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, d_depth.p, start, 0u);
            hipLaunchKernelGGL(or_u32_kernel, dim3(1), dim3(1), 0, s, d_vis.p, start >> 5, 1u << (start & 31));
            hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, d_order.p, 0u, start);
            uint32_t lo = 0, fsize = 1, level = 0;
            while (fsize > 0) {
                if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
                const uint32_t *fr = d_order.p + lo;
                const int g = grid_for((uint64_t)fsize * kBfsLanes);  // a 16-lane group per frontier node
                if (one_pass) {
                    // d_misc: [0] next frontier size (scan total), [1] goals left, [2] list slots, [3] long stretches, [4] long lists
                    CZ_HIP(hipMemsetAsync(d_misc.p + 2, 0, 12, s));
                    const uint32_t rows = (fsize + 31u) >> 5;  // (the counters' layout: bfs_cnt_at)
                    CZ_HIP(hipMemsetAsync(d_cnt.p, 0, (size_t)rows * 32 * 4, s));
                    hipLaunchKernelGGL(bfs_discover_kernel, dim3(g), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, fsize, d_vis.p, d_claim.p,
                                       d_fresh.p, d_misc.p + 2, d_long.p, d_misc.p + 4);
                    hipLaunchKernelGGL(bfs_discover_long_kernel, dim3(1024), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, d_long.p, d_misc.p + 4,
                                       d_vis.p, d_claim.p, d_fresh.p, d_misc.p + 2);
                    // (the slot count stays on the device; at least 1 024 workgroups: a frontier of ten hubs lists a million nodes)
                    const int gf = std::max(1024, grid_for(std::min<uint64_t>((uint64_t)N + N / 2, (uint64_t)fsize * 64 + 4096)));
                    hipLaunchKernelGGL(bfs_tally_kernel, dim3(gf), dim3(kT), 0, s, d_fresh.p, d_misc.p + 2, d_claim.p, d_cnt.p, rows);
                    hipLaunchKernelGGL(bfs_counts_in_order_kernel, dim3(grid_for(fsize)), dim3(kT), 0, s, d_cnt.p, fsize, rows, d_big.p);
                    rc = exclusive_scan(d_big.p, d_pos.p, fsize, d_misc.p, d_scratch.p, s);  // (d_big is free until bfs_order_small_kernel)
                    if (rc) return rc;
                    hipLaunchKernelGGL(bfs_place_kernel, dim3(gf), dim3(kT), 0, s, d_fresh.p, d_misc.p + 2, d_claim.p, fr, d_pos.p, d_cnt.p,
                                       d_order.p + lo + fsize, d_parent.p, d_depth.p, d_vis.p, level + 1, rows);
                    hipLaunchKernelGGL(bfs_order_small_kernel, dim3(grid_for(fsize)), dim3(kT), 0, s, d_pos.p, fsize, d_misc.p,
                                       d_order.p + lo + fsize, d_big.p, d_misc.p + 3);
                    hipLaunchKernelGGL(bfs_order_big_kernel, dim3(1024), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, d_pos.p, d_big.p,
                                       d_misc.p + 3, d_claim.p, d_depth.p, level + 1, d_order.p + lo + fsize);
                } else {
                hipLaunchKernelGGL(bfs_claim_kernel, dim3(g), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, fsize, d_depth.p, d_vis.p, d_claim.p,
                                   0u, N);
                hipLaunchKernelGGL(bfs_count_kernel, dim3(g), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, fsize, d_depth.p, d_vis.p, d_claim.p,
                                   d_cnt.p, d_won.p, 0u, N);
                rc = exclusive_scan(d_cnt.p, d_pos.p, fsize, d_misc.p, d_scratch.p, s);
                if (rc) return rc;
                hipLaunchKernelGGL(bfs_emit_kernel, dim3(g), dim3(kT), 0, s, G.off.p, G.tgt.p, fr, fsize, d_depth.p, d_vis.p, d_claim.p,
                                   d_won.p, d_pos.p, d_order.p + lo + fsize, d_parent.p, level + 1, 0u, N, 0u);
                }
                CZ_HIP(hipMemsetAsync(d_misc.p + 1, 0, 4, s));
                if (goals)
                    hipLaunchKernelGGL(bfs_goals_left_kernel, dim3(grid_for(n_goals)), dim3(kT), 0, s, d_goals.p, n_goals, N,
                                       d_depth.p, start, d_misc.p + 1);
                uint32_t h[2];
                CZ_HIP(hipMemcpy(h, d_misc.p, 8, hipMemcpyDeviceToHost));
                lo += fsize;
                fsize = h[0];
                if (on_level && fsize) {
                    // the level's discoveries go to the caller now (they would at the end anyway): it evaluates its condition on
                    // exactly the nodes the reference's loop would have looked at, in the same order, and says when it has enough
                    uint32_t *dst = order + merged_total + reached;
                    CZ_HIP(hipMemcpy(dst, d_order.p + lo, (size_t)fsize * 4, hipMemcpyDeviceToHost));
                    const int verdict = on_level(on_level_ctx, start, dst, fsize);
                    if (verdict < 0) return cz::set_error(CZ_E_INVALID, "the level callback failed (%d)", verdict);
                    stop = verdict > 0;
                }
                reached += fsize;
                level++;
                if (goals && h[1] == 0) break;
                if (stop) break;
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "bfs launch: %s", hipGetErrorString(e));
        }
        t_timing.lap(T_DEVICE);
        if (merged) {
            if (reached) {
                // the claims of this traversal's nodes are forgotten (see above), its discovery sequence joins the others'
                hipLaunchKernelGGL(scatter_u32_kernel, dim3(grid_for(reached)), dim3(kT), 0, s, d_claim.p, d_order.p + 1, reached, CZ_NONE);
                uint32_t *dst = order + merged_total;
                if (!on_level) CZ_HIP(hipMemcpy(dst, d_order.p + 1, (size_t)reached * 4, hipMemcpyDeviceToHost));
                for (uint32_t i = 0; i < reached; i++) hvis[dst[i] >> 5] |= 1u << (dst[i] & 31);
                merged_total += reached;
            }
            n_reached[si + 1] = (uint32_t)merged_total;
            t_timing.lap(T_DOWNLOAD);
            continue;
        }
        CZ_HIP(hipMemcpy(parent + (size_t)si * N, d_parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        if (depth) {
            CZ_HIP(hipMemcpy(depth + (size_t)si * N, d_depth.p, (size_t)N * 4, hipMemcpyDeviceToHost));
            if (!run && !share_visited)
                for (uint32_t i = 0; i < N; i++) depth[(size_t)si * N + i] = CZ_NONE;
        }
        if (order && reached) CZ_HIP(hipMemcpy(order + (size_t)si * N, d_order.p + 1, (size_t)reached * 4, hipMemcpyDeviceToHost));
        if (n_reached) n_reached[si] = reached;
        t_timing.lap(T_DOWNLOAD);
    }
    if (merged) CZ_HIP(hipMemcpy(parent, d_parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
}

}  // namespace

extern "C" int cz_bfs_shared(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                             uint32_t n_starts, uint32_t *parent, uint32_t *order, uint32_t *first, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (first)
        for (uint32_t i = 0; i <= n_starts; i++) first[i] = 0;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !parent || !order || !first) return cz::set_error(CZ_E_INVALID, "null starts/parent/order/first");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, nullptr, N, E))) return rc;
    return bfs_run(G, starts, n_starts, nullptr, 0, 1, parent, nullptr, order, first, poison, true);
}

extern "C" int cz_bfs_shared_until(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E, const uint32_t *starts,
                                   uint32_t n_starts, cz_bfs_level_fn on_level, void *ctx, uint32_t *parent, uint32_t *order,
                                   uint32_t *first, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (first)
        for (uint32_t i = 0; i <= n_starts; i++) first[i] = 0;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !parent || !order || !first) return cz::set_error(CZ_E_INVALID, "null starts/parent/order/first");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, nullptr, N, E))) return rc;
    return bfs_run(G, starts, n_starts, nullptr, 0, 1, parent, nullptr, order, first, poison, true, on_level, ctx);
}

extern "C" int cz_bfs(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E,
                      const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals,
                      int share_visited, uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached,
                      const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !parent) return cz::set_error(CZ_E_INVALID, "null starts/parent");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, nullptr, N, E))) return rc;
    return bfs_run(G, starts, n_starts, goals, n_goals, share_visited, parent, depth, order, n_reached, poison);
}

extern "C" int cz_bfs_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals,
                         int share_visited, uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached,
                         const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (!g) return cz::set_error(CZ_E_INVALID, "null graph");
    if (n_starts == 0 || g->N == 0) return CZ_OK;
    if (!starts || !parent) return cz::set_error(CZ_E_INVALID, "null starts/parent");
    return bfs_run(*g, starts, n_starts, goals, n_goals, share_visited, parent, depth, order, n_reached, poison);
}

namespace {

int cc_run(const cz_graph &G, uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison) {
    const uint32_t N = G.N;
    int rc = CZ_OK;
    cz::DevBuf<uint32_t> d_label, d_flag, d_rank, d_scratch, d_misc;
    CZ_HIP(d_label.alloc(N));
    CZ_HIP(d_flag.alloc(std::max<size_t>(N, kCcSamples)));  // (first the sample of step 2, then the root flags)
    CZ_HIP(d_rank.alloc(N));
    CZ_HIP(d_scratch.alloc(scan_scratch_words(N)));
    CZ_HIP(d_misc.alloc(4));
    hipStream_t s = nullptr;
    const int g = grid_for(N);
    t_timing.lap(T_UPLOAD);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kT), 0, s, d_label.p, N);
    for (uint32_t r = 0; r < kCcNeighbourRounds; r++) {
        hipLaunchKernelGGL(cc_link_nth_kernel, dim3(g), dim3(kT), 0, s, G.off.p, G.tgt.p, N, r, d_label.p);
        hipLaunchKernelGGL(cc_compress_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p);
    }
    // the component most of a fixed sample of the nodes is in
    uint32_t sample[kCcSamples];
    hipLaunchKernelGGL(cc_sample_kernel, dim3(kCcSamples / kT), dim3(kT), 0, s, N, d_label.p, d_flag.p);
    CZ_HIP(hipMemcpy(sample, d_flag.p, sizeof sample, hipMemcpyDeviceToHost));
    if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    std::sort(sample, sample + kCcSamples);
    uint32_t skip = sample[0], best = 0;
    for (uint32_t i = 0, j; i < kCcSamples; i = j) {
        for (j = i; j < kCcSamples && sample[j] == sample[i]; j++) {}
        if (j - i > best) best = j - i, skip = sample[i];
    }
    hipLaunchKernelGGL(cc_link_rest_kernel, dim3(grid_for((uint64_t)N * kCcLanes)), dim3(kT), 0, s, G.off.p, G.tgt.p, N,
                       kCcNeighbourRounds, skip, d_label.p);
    hipLaunchKernelGGL(cc_compress_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p);
    hipLaunchKernelGGL(cc_rootflag_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p, d_flag.p);
    rc = exclusive_scan(d_flag.p, d_rank.p, N, d_misc.p, d_scratch.p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cc_group_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p, d_rank.p, d_flag.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "cc launch: %s", hipGetErrorString(e));
    t_timing.lap(T_DEVICE);
    CZ_HIP(hipMemcpy(group, d_flag.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    uint32_t total = 0;
    CZ_HIP(hipMemcpy(&total, d_misc.p, 4, hipMemcpyDeviceToHost));
    if (n_groups) *n_groups = total;
    t_timing.lap(T_DOWNLOAD);
    return CZ_OK;
}

}  // namespace

extern "C" int cz_connected_components(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E,
                                       uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison) {
    if (n_groups) *n_groups = 0;
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (N == 0) return CZ_OK;
    if (!group) return cz::set_error(CZ_E_INVALID, "null group");
    cz_graph G;
    if ((rc = graph_fill(G, offsets, targets, nullptr, N, E))) return rc;
    return cc_run(G, group, n_groups, poison);
}

extern "C" int cz_connected_components_on(const cz_graph *g, uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison) {
    if (n_groups) *n_groups = 0;
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (!g) return cz::set_error(CZ_E_INVALID, "null graph");
    if (g->N == 0) return CZ_OK;
    if (!group) return cz::set_error(CZ_E_INVALID, "null group");
    return cc_run(*g, group, n_groups, poison);
}

// ---- ClusteringCoefficients (fixed_rule/algos/triangles.rs:25-110) ---------------------------------------------
// For node v with adjacency list A (the symmetrised graph's out-neighbours: ascending, parallel edges kept),
// n_triangles(v) = #{ (i, j) : A[i] > A[j] and A[j] is an out-neighbour of A[i] } -- list POSITIONS, so duplicates
// count with their multiplicity exactly as the reference's nested `edges.iter()` loops do (:84-101); membership is
// existence (`for nb in out_neighbors(e_src) { if nb == e_dst { return true } }`).  One wave per node: lane pairs
// (i, j) are enumerated i-major, the membership test is a binary search in A[i]'s sorted list.  Integer output; the
// host computes cc = 2 t / (d (d - 1)) in f64 like :102.
__device__ __forceinline__ bool csr_contains(const uint32_t *__restrict__ tgt, uint32_t lo, uint32_t hi, uint32_t x) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t v = tgt[mid];
        if (v < x) lo = mid + 1;
        else if (v > x) hi = mid;
        else return true;
    }
    return false;
}

__global__ void __launch_bounds__(256)
triangles_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N,
                 unsigned long long *__restrict__ n_tri, uint32_t *__restrict__ degree) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (uint32_t v = wave; v < N; v += n_waves) {
        const uint32_t a = off[v], b = off[v + 1], d = b - a;
        unsigned long long cnt = 0;
        // pairs (i, j), j < i suffices: the list is ascending, so A[i] > A[j] implies j < i.  The d (d - 1) / 2 pairs are laid
        // out flat, p = i (i - 1) / 2 + j, and dealt to the lanes 64 at a time: every lane is busy and a node of degree 20
        // takes 3 steps (one step per i with the lanes j < i -- 19 steps, 15 % of the lanes -- measured 77 ms on the
        // symmetrised 10M / 200M graph)
        const unsigned long long P = (unsigned long long)d * (d - (d > 0)) / 2;
        for (unsigned long long p = lane; p < P; p += 64) {
            uint32_t i = (uint32_t)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
            while ((unsigned long long)i * (i - 1) / 2 > p) i--;
            while ((unsigned long long)(i + 1) * i / 2 <= p) i++;
            const uint32_t j = (uint32_t)(p - (unsigned long long)i * (i - 1) / 2);
            const uint32_t src = tgt[a + i], dst = tgt[a + j];
            if (dst < src && csr_contains(tgt, off[src], off[src + 1], dst)) cnt++;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) {
            n_tri[v] = cnt;
            degree[v] = d;
        }
    }
}

// Round 3: every triangle is found ONCE, from its smallest corner.  The rule always runs on the symmetrised graph
// (ClusteringCoefficients::run converts with undirected = true, triangles.rs:37), where the count above is symmetric in the three
// corners: for a triangle {a < b < c} with multiplicities m_ab, m_ac, m_bc (parallel edges: a pair of nodes linked in both
// directions appears twice in each other's list) the position pairs give  tri(a) += m_ab m_ac, tri(b) += m_ab m_bc,
// tri(c) += m_ac m_bc.  So node v only enumerates pairs of DISTINCT neighbours above itself -- a quarter of the pairs -- tests
// membership once with the multiplicity (equal range instead of existence), and credits all three corners with atomics
// (triangles are rare next to pairs: 4 125 incidences against 10^9 pairs on the 10M / 200M uniform graph).
// Precondition: the adjacency is symmetric with symmetric multiplicities; the caller vouches for it (CZ_ADJ_SYMMETRIC) or
// tri_symmetry_kernel verifies it exactly, and the call falls back to the general kernel otherwise (CZ_TRI_GENERAL=1 forces it).
__device__ __forceinline__ uint32_t csr_count(const uint32_t *__restrict__ tgt, uint32_t lo, uint32_t hi, uint32_t x) {
    uint32_t l = lo, h = hi;
    while (l < h) {  // lower bound
        const uint32_t mid = l + ((h - l) >> 1);
        if (tgt[mid] < x) l = mid + 1;
        else h = mid;
    }
    uint32_t c = 0;
    while (l + c < hi && tgt[l + c] == x) c++;  // multiplicities are 1 or 2 in practice
    return c;
}

// EXACT (round 4; the sampled form of round 3 could miss a violation): a 16-lane group per node u walks its list; every entry
// v > u that is the first of its run must occur in v's list as often as v does in u's; and the graph must hold as many entries
// below their node as above -- a pair (u < v) that only v lists is never met from u, but it adds to `below` alone.
// bad: violations; updown[0] / [1]: entries above / below their node.
__global__ void __launch_bounds__(256)
tri_symmetry_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, uint32_t *__restrict__ bad,
                    unsigned long long *__restrict__ updown) {
    const uint32_t gl = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4, ngroups = ((uint64_t)gridDim.x * 256) >> 4;
    unsigned long long up = 0, down = 0;
    uint32_t wrong = 0;
    for (uint64_t u = group; u < N; u += ngroups) {
        const uint32_t a = off[u], z = off[u + 1];
        for (uint32_t e = a + gl; e < z; e += 16) {
            const uint32_t v = tgt[e];
            if (v >= N) { wrong++; continue; }
            if (v < (uint32_t)u) { down++; continue; }
            if (v == (uint32_t)u) continue;  // self loops are judged by tri_self_loop_kernel
            up++;
            if (e > a && tgt[e - 1] == v) continue;  // not the first of its run
            uint32_t mine = 1;
            while (e + mine < z && tgt[e + mine] == v) mine++;
            if (csr_count(tgt, off[v], off[v + 1], (uint32_t)u) != mine) wrong++;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    if (up) atomicAdd(&updown[0], up);
    if (down) atomicAdd(&updown[1], down);
}

// self loops make degenerate "triangles" (the entry v of A(v) pairs with every other neighbour: triangles.rs:84-101 has no
// rule against it); they do not have three distinct corners, so a graph that holds one takes the general kernel
__global__ void __launch_bounds__(256)
tri_self_loop_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, uint32_t *__restrict__ found) {
    for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < N; v += gridDim.x * 256)
        if (csr_count(tgt, off[v], off[v + 1], v)) atomicAdd(found, 1u);
}

// A 16-lane group per node.  Second form (a wave per node, every pair (c, b) a binary search in c's list): 14.4 ms -- a node of
// the 10M / 200M graph has ~10 neighbours above itself, i.e. ~45 pairs x ~5 probes, and 2 x 10^9 probes through the L1 / L2
// path are what that costs, wherever the lines come from.  Now c's list is loaded ONCE per (node, c), a lane per entry, and
// rotated past the lanes (DPP row_ror:1 inside the 16-lane row): every lane holds one neighbour b of the node and counts how
// often it meets it -- 2 loads and 48 register steps per list instead of 45 probe chains per node.  Nodes with more than
// kTriMerge neighbours above themselves go to triangles_hub_kernel (this form reads every c's list once per 16 of the node's
// neighbours: the square of the list), and a c whose own list is longer than kTriScan is searched instead of rotated past.
// (64 / 128: the uniform 10M / 200M graph never meets either bound -- 10.9 ms as before; R-MAT 10M / 200M: 1236 -> 50 ms here.)
constexpr int kTriLanes = 16;
constexpr uint32_t kTriMerge = 64, kTriScan = 128;

__device__ __forceinline__ void tri_credit(const uint32_t *__restrict__ tgt, uint32_t a, uint32_t b, uint32_t v, uint32_t pi, uint32_t pj,
                                           uint32_t m_bc, unsigned long long *__restrict__ n_tri) {
    // pi / pj: FIRST positions of the neighbours c > bb > v in v's list; multiplicities = run lengths (pairs linked both ways)
    const uint32_t c = tgt[pi], bb = tgt[pj];
    uint32_t m_ac = 1, m_ab = 1;
    while (pi + m_ac < b && tgt[pi + m_ac] == c) m_ac++;
    while (pj + m_ab < b && tgt[pj + m_ab] == bb) m_ab++;
    atomicAdd(&n_tri[v], (unsigned long long)m_ab * m_ac);
    atomicAdd(&n_tri[bb], (unsigned long long)m_ab * m_bc);
    atomicAdd(&n_tri[c], (unsigned long long)m_ac * m_bc);
    (void)a;
}

__global__ void __launch_bounds__(256)
triangles_oriented_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N,
                          unsigned long long *__restrict__ n_tri /* zeroed */, uint32_t *__restrict__ degree,
                          uint32_t *__restrict__ hubs /* [N] */, uint32_t *__restrict__ n_hubs /* [2]: count, longest; zeroed */, uint32_t merge_max,
                          uint32_t scan_max) {
    const uint32_t glane = threadIdx.x & (kTriLanes - 1);
    const uint32_t group = (blockIdx.x * 256 + threadIdx.x) / kTriLanes, n_groups = (gridDim.x * 256) / kTriLanes;
    for (uint32_t v = group; v < N; v += n_groups) {
        const uint32_t a = off[v], b = off[v + 1];
        if (glane == 0) degree[v] = b - a;
        // s = first position whose neighbour is above v (the list is ascending)
        uint32_t s = a, h = b;
        while (s < h) {
            const uint32_t mid = s + ((h - s) >> 1);
            if (tgt[mid] <= v) s = mid + 1;
            else h = mid;
        }
        const uint32_t m = b - s;
        if (m < 2) continue;
        if (m <= merge_max) {
            for (uint32_t ca = 0; ca < m; ca += kTriLanes) {  // this lane's neighbour bb (first occurrences only)
                const uint32_t pj = s + ca + glane;
                const bool have = ca + glane < m;
                const uint32_t bb = have ? tgt[pj] : CZ_NONE;
                const bool first_b = have && !(pj > a && tgt[pj - 1] == bb);
                // the larger corners c, 16 at a time: every lane fetches ONE c with the bounds of its list, so that the chain
                // neighbour -> offsets costs one round trip per 16 corners instead of one per corner; the loop over the corners
                // then only reads lanes.  Positions ascend with values: corners at or before the chunk's first position have
                // nothing of this chunk below them.
                for (uint32_t cc = ca & ~(uint32_t)(kTriLanes - 1); cc < m; cc += kTriLanes) {
                    const uint32_t pc = s + cc + glane;
                    const bool have_c = cc + glane < m;
                    const uint32_t my_c = have_c ? tgt[pc] : CZ_NONE;
                    const bool rep = have_c && pc > a && tgt[pc - 1] == my_c;  // a repeated neighbour: counted at its first position
                    const uint32_t my_oc = have_c && !rep ? off[my_c] : 0, my_oc1 = have_c && !rep ? off[my_c + 1] : 0;
                    const uint32_t c_end = min((uint32_t)kTriLanes, m - cc);
                    for (uint32_t t = 0; t < c_end; t++) {  // (group-uniform)
                        const uint32_t ci = cc + t;
                        if (ci <= ca) continue;
                        const uint32_t c = (uint32_t)__shfl((int)my_c, (int)t, kTriLanes);
                        const uint32_t oc = (uint32_t)__shfl((int)my_oc, (int)t, kTriLanes), oc1 = (uint32_t)__shfl((int)my_oc1, (int)t, kTriLanes);
                        if (oc1 == oc) continue;  // repeated (or a neighbour without a list: cannot be, the graph is symmetric)
                        const bool want = first_b && bb < c;
                        uint32_t cnt = 0;
                        if (oc1 - oc > scan_max) {  // c's list is long: a search per lane instead of the whole list past every lane
                                                    // (a hub with high ids had its list read once per NEIGHBOUR: degree^2 words)
                            if (want) cnt = csr_count(tgt, oc, oc1, bb);
                        } else
                        for (uint32_t k = oc; k < oc1; k += kTriLanes) {
                            uint32_t L = k + glane < oc1 ? tgt[k + glane] : CZ_NONE - 1u;
#pragma unroll
                            for (int r = 0; r < kTriLanes; r++) {
                                cnt += L == bb ? 1u : 0u;
                                L = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)L, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
                            }
                        }
                        if (want && cnt) tri_credit(tgt, a, b, v, s + ci, pj, cnt, n_tri);
                    }
                }
            }
            continue;
        }
        // more than merge_max neighbours above itself: left to triangles_hub_kernel.  Until round 6 one 16-lane group walked all
        // m (m - 1) / 2 pairs of its list here -- 10^9 pairs for ONE node of a 1M-node R-MAT graph, 114 s for the rule.
        if (glane == 0) {
            hubs[atomicAdd(n_hubs, 1u)] = v;
            atomicMax(n_hubs + 1, m);  // (the longest such list: whether any needs the map in global memory)
        }
    }
}

// The hubs.  A triangle {v < bb < c} is found from v through bb's list instead of through the pairs of v's own: for every
// neighbour bb above v, every c above bb in bb's list, "how often is c in v's list?" -- sum over bb of its (short) list instead of
// the square of v's (long) one.  A binary search in a 200 000-entry list is 17 dependent loads, and a wave pays them whenever ONE
// of its 64 lanes has to: the workgroup first writes v's list into a set of its own that answers with the multiplicity (0, 1, 2,
// "3 or more": only the last falls back to the search) in one access --
//   * up to kTriHubLds distinct neighbours: an open-addressing table in LDS (sized to the list, at most half full);
//   * more: two bits per node in a map of N entries in global memory, one map per workgroup (written with atomics, read past
//     the L1 -- both act at the L2 -- and wiped entry by entry afterwards).
// One 1024-thread workgroup per hub at a time (two on a CU), hubs handed out through a counter (they are listed roughly largest
// first); neighbours whose own list above themselves is long (other hubs) are queued in LDS and walked by a wave each
// afterwards instead of by one 16-lane group.  The credits of the two smaller corners are summed in registers: every triangle of
// a hub adds to n_tri[v], every triangle found through bb to n_tri[bb] -- only the largest corner (scattered over the nodes) is
// credited at once.  Same credit, same multiplicities as the merge form.
constexpr uint32_t kTriHubThreads = 1024, kTriHubQueue = 2048, kTriHubLong = 512, kTriHubSlots = 8192, kTriHubLds = kTriHubSlots / 2;

struct TriHubSet {
    const uint32_t *keys;  // LDS table (mask != 0) ...
    const uint8_t *vals;
    uint32_t mask, shift;
    const uint32_t *map;  // ... or the 2-bit map in global memory
    __device__ __forceinline__ uint32_t count(uint32_t c) const {
        if (mask) {
            uint32_t slot = (c * 2654435761u) >> shift;
            for (;;) {
                const uint32_t k = keys[slot];
                if (k == c) return vals[slot];
                if (k == CZ_NONE) return 0;
                slot = (slot + 1) & mask;
            }
        }
        return (__hip_atomic_load(&map[c >> 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (2u * (c & 15u))) & 3u;
    }
};

// One entry of bb's list: c with the entries before and after it, loaded together with it (the two are in c's cache line nearly
// always; a dependent load on the hit path costs every lane of the wave its latency).  Taking the neighbours from the lanes
// next door instead (DPP rotations, one load per span for the entry past its end) was tried and is SLOWER: 633 against 400 ms.
struct TriEntry {
    uint32_t c, prev, next;
};
__device__ __forceinline__ TriEntry tri_entry(const uint32_t *__restrict__ tgt, uint32_t ob, uint32_t ob1, uint32_t k) {
    TriEntry e;
    const bool in = k < ob1;
    e.c = in ? tgt[k] : CZ_NONE;
    e.prev = in && k > ob ? tgt[k - 1] : CZ_NONE;
    e.next = in && k + 1 < ob1 ? tgt[k + 1] : CZ_NONE;
    return e;
}

__device__ __forceinline__ void tri_hub_probe(const uint32_t *__restrict__ tgt, const TriHubSet &set, uint32_t s, uint32_t b, uint32_t m_ab,
                                              uint32_t ob1, uint32_t k, const TriEntry &e, unsigned long long &acc_v,
                                              unsigned long long &acc_b, unsigned long long *__restrict__ n_tri) {
    const uint32_t c = e.c;
    if (c == CZ_NONE || e.prev == c) return;  // past the list / counted at its first position
    uint32_t m_ac = set.count(c);
    if (!m_ac) return;
    uint32_t m_bc = 1;
    if (e.next == c) {
        m_bc = 2;
        while (k + m_bc < ob1 && tgt[k + m_bc] == c) m_bc++;
    }
    if (m_ac == 3) {  // "3 or more": the run in v's list is measured
        uint32_t l = s, r = b;
        while (l < r) {
            const uint32_t mid = l + ((r - l) >> 1);
            if (tgt[mid] < c) l = mid + 1;
            else r = mid;
        }
        while (l + m_ac < b && tgt[l + m_ac] == c) m_ac++;
    }
    acc_v += (unsigned long long)m_ab * m_ac;
    acc_b += (unsigned long long)m_ab * m_bc;
    atomicAdd(&n_tri[c], (unsigned long long)m_ac * m_bc);
}

__device__ __forceinline__ unsigned long long tri_wave_sum(unsigned long long x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += (unsigned long long)__shfl_xor((long long)x, d, 64);
    return x;
}

__global__ void __launch_bounds__(kTriHubThreads, 2)
triangles_hub_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ hubs,
                     const uint32_t *__restrict__ n_hubs, uint32_t *__restrict__ next_hub /* zeroed */,
                     uint32_t *maps /* gridDim.x x words, zeroed */, uint32_t words, unsigned long long *__restrict__ n_tri) {
    __shared__ uint32_t keys[kTriHubSlots];
    __shared__ uint8_t vals[kTriHubSlots];
    __shared__ uint32_t q_pj[kTriHubQueue], q_lo[kTriHubQueue];
    __shared__ uint32_t q_n, h_sh;
    const uint32_t glane = threadIdx.x & (kTriLanes - 1), gi = threadIdx.x / kTriLanes;
    constexpr uint32_t kGroups = kTriHubThreads / kTriLanes;
    uint32_t *map = maps + (size_t)blockIdx.x * words;
    const uint32_t nh = *n_hubs;
    for (;;) {
        if (threadIdx.x == 0) {
            h_sh = atomicAdd(next_hub, 1u);
            q_n = 0;
        }
        __syncthreads();
        const uint32_t h = h_sh;
        if (h >= nh) break;
        const uint32_t v = hubs[h];
        const uint32_t a = off[v], b = off[v + 1];
        uint32_t s = a, hh = b;
        while (s < hh) {  // first position whose neighbour is above v
            const uint32_t mid = s + ((hh - s) >> 1);
            if (tgt[mid] <= v) s = mid + 1;
            else hh = mid;
        }
        // ---- v's list into the set (first positions of the runs, multiplicity capped at 3) ----
        const uint32_t m = b - s;
        TriHubSet set{keys, vals, 0, 0, map};
        if (m <= kTriHubLds) {
            uint32_t bits = 6;
            while ((1u << bits) < 2 * m) bits++;
            set.mask = (1u << bits) - 1;
            set.shift = 32 - bits;
            for (uint32_t i = threadIdx.x; i <= set.mask; i += kTriHubThreads) keys[i] = CZ_NONE;
            __syncthreads();
        }
        for (uint32_t p = s + threadIdx.x; p < b; p += kTriHubThreads) {
            const uint32_t x = tgt[p];
            if (p > s && tgt[p - 1] == x) continue;
            uint32_t run = 1;
            while (run < 3 && p + run < b && tgt[p + run] == x) run++;
            if (set.mask) {
                uint32_t slot = (x * 2654435761u) >> set.shift;
                while (atomicCAS(&keys[slot], CZ_NONE, x) != CZ_NONE) slot = (slot + 1) & set.mask;  // (x occurs once here: first of its run)
                vals[slot] = (uint8_t)run;
            } else {
                atomicOr(&map[x >> 4], run << (2u * (x & 15u)));
            }
        }
        __syncthreads();  // (waits for the atomics as well: they have acted at the L2 by now)
        unsigned long long acc_v = 0;
        for (uint32_t pj = s + gi; pj < b; pj += kGroups) {
            const uint32_t bb = tgt[pj];
            if (pj > a && tgt[pj - 1] == bb) continue;  // counted at its first position
            uint32_t m_ab = 1;
            while (pj + m_ab < b && tgt[pj + m_ab] == bb) m_ab++;
            const uint32_t ob = off[bb], ob1 = off[bb + 1];
            uint32_t lo = ob, hi = ob1;
            while (lo < hi) {  // bb's neighbours above bb
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (tgt[mid] <= bb) lo = mid + 1;
                else hi = mid;
            }
            if (ob1 - lo > kTriHubLong) {  // (group-uniform) another hub: the whole workgroup walks it below
                uint32_t slot = 0;
                if (glane == 0) slot = atomicAdd(&q_n, 1u);
                slot = (uint32_t)__shfl((int)slot, 0, kTriLanes);
                if (slot < kTriHubQueue) {
                    if (glane == 0) {
                        q_pj[slot] = pj;
                        q_lo[slot] = lo;
                    }
                    continue;
                }
            }
            unsigned long long acc_b = 0;
            for (uint32_t k = lo + glane; k < ob1; k += 2 * kTriLanes) {  // two entries per lane in flight
                const TriEntry e0 = tri_entry(tgt, ob, ob1, k), e1 = tri_entry(tgt, ob, ob1, k + kTriLanes);
                tri_hub_probe(tgt, set, s, b, m_ab, ob1, k, e0, acc_v, acc_b, n_tri);
                tri_hub_probe(tgt, set, s, b, m_ab, ob1, k + kTriLanes, e1, acc_v, acc_b, n_tri);
            }
            if (acc_b) atomicAdd(&n_tri[bb], acc_b);
        }
        __syncthreads();
        const uint32_t nq = min(q_n, kTriHubQueue);
        for (uint32_t i = threadIdx.x >> 6; i < nq; i += kTriHubThreads / 64) {  // a wave per queued neighbour
            const uint32_t pj = q_pj[i], bb = tgt[pj];
            uint32_t m_ab = 1;
            while (pj + m_ab < b && tgt[pj + m_ab] == bb) m_ab++;
            const uint32_t ob = off[bb], ob1 = off[bb + 1];
            unsigned long long acc_b = 0;
            for (uint32_t k = q_lo[i] + (threadIdx.x & 63u); k < ob1; k += 128) {
                const TriEntry e0 = tri_entry(tgt, ob, ob1, k), e1 = tri_entry(tgt, ob, ob1, k + 64);
                tri_hub_probe(tgt, set, s, b, m_ab, ob1, k, e0, acc_v, acc_b, n_tri);
                tri_hub_probe(tgt, set, s, b, m_ab, ob1, k + 64, e1, acc_v, acc_b, n_tri);
            }
            acc_b = tri_wave_sum(acc_b);
            if ((threadIdx.x & 63u) == 0 && acc_b) atomicAdd(&n_tri[bb], acc_b);
        }
        acc_v = tri_wave_sum(acc_v);
        if ((threadIdx.x & 63u) == 0 && acc_v) atomicAdd(&n_tri[v], acc_v);
        __syncthreads();  // (nobody still asks the set; h_sh and q_n may be written again)
        if (!set.mask) {
            for (uint32_t p = s + threadIdx.x; p < b; p += kTriHubThreads) map[tgt[p] >> 4] = 0;  // the map goes back to all zero
            __syncthreads();
        }
    }
}

extern "C" int cz_clustering_coefficients(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E,
                                          uint64_t *n_triangles, uint32_t *degree, const volatile uint8_t *poison, uint32_t flags) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (N == 0) return CZ_OK;
    if (!n_triangles || !degree) return cz::set_error(CZ_E_INVALID, "null output");
    rc = check_csr(offsets, targets, N, E);
    if (rc) return rc;
    if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    cz::DevBuf<uint32_t> d_off, d_tgt, d_deg;
    cz::DevBuf<unsigned long long> d_tri;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_tgt.alloc(E));
    CZ_HIP(d_deg.alloc(N));
    CZ_HIP(d_tri.alloc(N));
    CZ_HIP(hipMemcpy(d_off.p, offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) CZ_HIP(hipMemcpy(d_tgt.p, targets, E * 4, hipMemcpyHostToDevice));
    const int blocks = (int)std::min<uint64_t>(256 * 16, ((uint64_t)N + 3) / 4);
    t_timing.lap(T_UPLOAD);
    bool general = getenv("CZ_TRI_GENERAL") && atoi(getenv("CZ_TRI_GENERAL")) != 0;
    if (!general && E > 0) {  // the oriented count needs a symmetric adjacency without self loops: both are checked EXACTLY, the former
                              // unless the caller vouches for it (CZ_ADJ_SYMMETRIC: the rule, which builds the adjacency that way)
        cz::DevBuf<uint32_t> d_bad;
        cz::DevBuf<unsigned long long> d_updown;
        CZ_HIP(d_bad.alloc(1));
        CZ_HIP(d_updown.alloc(2));
        CZ_HIP(hipMemsetAsync(d_bad.p, 0, 4, nullptr));
        CZ_HIP(hipMemsetAsync(d_updown.p, 0, 16, nullptr));
        if (!(flags & CZ_ADJ_SYMMETRIC)) {
            hipLaunchKernelGGL(tri_symmetry_kernel, dim3(grid_for((uint64_t)N * 16)), dim3(256), 0, nullptr, d_off.p, d_tgt.p, N, d_bad.p,
                               d_updown.p);
        } else {  // vouched for: the range of the targets is still checked (a wrong one is an error, not a reason for the general kernel)
            cz::DevBuf<uint32_t> d_oor;
            CZ_HIP(d_oor.alloc(1));
            CZ_HIP(hipMemsetAsync(d_oor.p, 0, 4, nullptr));
            hipLaunchKernelGGL(targets_in_range_kernel, dim3(grid_for(E)), dim3(256), 0, nullptr, d_tgt.p, E, N, d_oor.p);
            uint32_t oor = 0;
            CZ_HIP(hipMemcpy(&oor, d_oor.p, 4, hipMemcpyDeviceToHost));
            if (oor) return cz::set_error(CZ_E_INVALID, "a target is out of range");
        }
        hipLaunchKernelGGL(tri_self_loop_kernel, dim3(grid_for(N)), dim3(256), 0, nullptr, d_off.p, d_tgt.p, N, d_bad.p);
        uint32_t bad = 0;
        unsigned long long updown[2] = {0, 0};
        CZ_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
        CZ_HIP(hipMemcpy(updown, d_updown.p, 16, hipMemcpyDeviceToHost));
        general = bad != 0 || updown[0] != updown[1];
    }
    if (general) {
        hipLaunchKernelGGL(triangles_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, nullptr, d_off.p, d_tgt.p, N, d_tri.p, d_deg.p);
    } else {
        cz::DevBuf<uint32_t> d_hubs, d_nhubs;
        CZ_HIP(d_hubs.alloc(N));
        CZ_HIP(d_nhubs.alloc(2));
        CZ_HIP(hipMemsetAsync(d_nhubs.p, 0, 8, nullptr));
        CZ_HIP(hipMemsetAsync(d_tri.p, 0, (size_t)N * 8, nullptr));
        const uint32_t merge_max = getenv("CZ_TRI_MERGE") ? (uint32_t)atoi(getenv("CZ_TRI_MERGE")) : kTriMerge;
        const uint32_t scan_max = getenv("CZ_TRI_SCAN") ? (uint32_t)atoi(getenv("CZ_TRI_SCAN")) : kTriScan;
        hipLaunchKernelGGL(triangles_oriented_kernel, dim3(grid_for((uint64_t)N * kTriLanes)), dim3(256), 0, nullptr, d_off.p, d_tgt.p, N,
                           d_tri.p, d_deg.p, d_hubs.p, d_nhubs.p, merge_max, scan_max);
        uint32_t nh[2] = {0, 0};  // how many nodes were left to the hub kernel, and the longest of their lists
        CZ_HIP(hipMemcpy(nh, d_nhubs.p, 8, hipMemcpyDeviceToHost));
        if (nh[0]) {
            // one 2-bit map of N entries per hub workgroup, if any list is too long for the LDS table: two workgroups per CU, fewer when
            // N is so large that the maps would pass 2 GiB together
            const uint32_t words = nh[1] > kTriHubLds ? (N + 15u) / 16u : 0u;
            const uint32_t hub_wgs = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(512, nh[0]), ((uint64_t)1 << 29) / std::max(words, 1u)));
            cz::DevBuf<uint32_t> d_maps, d_next;
            CZ_HIP(d_maps.alloc(std::max<size_t>((size_t)hub_wgs * words, 1)));
            CZ_HIP(d_next.alloc(1));
            CZ_HIP(hipMemsetAsync(d_maps.p, 0, std::max<size_t>((size_t)hub_wgs * words, 1) * 4, nullptr));
            CZ_HIP(hipMemsetAsync(d_next.p, 0, 4, nullptr));
            hipLaunchKernelGGL(triangles_hub_kernel, dim3(hub_wgs), dim3(kTriHubThreads), 0, nullptr, d_off.p, d_tgt.p, d_hubs.p, d_nhubs.p,
                               d_next.p, d_maps.p, words, d_tri.p);
            CZ_HIP(hipDeviceSynchronize());  // (the maps die with this scope)
        }
        CZ_HIP(hipDeviceSynchronize());  // (the hub list dies with this scope)
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "triangles launch: %s", hipGetErrorString(e));
    t_timing.lap(T_DEVICE);
    CZ_HIP(hipMemcpy(n_triangles, d_tri.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(degree, d_deg.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    t_timing.lap(T_DOWNLOAD);
    if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    return CZ_OK;
}

namespace {

// the multi-source near-far SSSP on device-resident state: `run` leaves the packed (cost, parent) words of `ns` sources in
// dp [ns][N]; shared by cz_sssp (which unpacks them) and cz_betweenness (which goes on to count paths over them)
struct SsspBatch {
    uint32_t N = 0, S = 0;
    uint64_t E = 0;
    float delta = 0.f;
    bool one_pile = false;
    hipStream_t s = nullptr;
    template <class T>
    struct View {  // the graph's arrays are somebody else's (a cz_graph: the call's own upload or a resident one)
        const T *p = nullptr;
    };
    View<uint32_t> d_off, d_tgt;
    View<float> d_w;
    // (from the stream-ordered pool: 0.56 GB per source batch, allocated and freed by every call -- common.h PoolBuf)
    cz::PoolBuf<uint32_t> d_qtag, d_ftag, d_misc, d_starts, d_lq;
    uint32_t lq_cap = 0;  // stretch items the long-list queue holds (0: no queue -- no long lists, or too many for one: the workgroups walk them)
    cz::PoolBuf<unsigned long long> d_dp, d_q[4];
    uint32_t *h_pin = nullptr;  // the counters come back through pinned memory: a round is a launch and one 32-byte copy
    ~SsspBatch() {
        if (h_pin) (void)hipHostFree(h_pin);
    }
    static constexpr uint32_t kMisc = 16;  // words of the counter block
    int read_counters(uint32_t *h) {
        CZ_HIP(hipMemcpyAsync(h_pin, d_misc.p, kMisc * 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        memcpy(h, h_pin, kMisc * 4);
        return CZ_OK;
    }

    int attach(const cz_graph &G, uint32_t n_starts, uint64_t pairs_budget) {
        N = G.N;
        E = G.E;
        if (N >= 0x80000000u) return cz::set_error(CZ_E_UNSUPPORTED, "node ids must stay below 2^31");
        if (!G.has_w) return cz::set_error(CZ_E_INVALID, "the graph was uploaded without weights");
        if (G.bad != ~0ull)  // BadEdgeWeightError, fixed_rule/mod.rs:258-286
            return cz::set_error(CZ_E_INVALID, "edge %llu has weight %g: weights must be non-negative numbers", G.bad, (double)G.bad_value);
        d_off.p = G.off.p;
        d_tgt.p = G.tgt.p;
        d_w.p = G.w.p;
        if (d_misc.n != kMisc) CZ_HIP(d_misc.alloc(kMisc));
        if (!h_pin) CZ_HIP(hipHostMalloc((void **)&h_pin, kMisc * 4));
        const double wsum = G.wsum;
        // bucket width of the near-far schedule: the mean edge weight (CZ_SSSP_DELTA overrides; <= 0 or "inf" = one pile,
        // i.e. plain frontier Bellman-Ford).  Only the schedule depends on it, never the result.
        delta = E ? (float)(wsum / (double)E) : 0.f;
        if (const char *de = getenv("CZ_SSSP_DELTA")) delta = (float)atof(de);
        one_pile = !(delta > 0.f) || !std::isfinite(delta);
        // sources per launch: 16 bytes of state + 32 bytes of queue space per (source, node)
        S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_starts, pairs_budget / std::max<uint32_t>(N, 1)));
        const uint64_t SN = (uint64_t)S * N;
        if (SN >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "too many (source, node) pairs per launch");
        // the long-list queue (SsspLongQ): a (source, node) pair is in a pile once per round, so S x the graph's stretches is all a
        // round can queue; 20 bytes an item, at most 1 GiB (beyond: no queue)
        {
            const uint64_t want = (uint64_t)S * G.long_stretches;
            const char *lq_env = getenv("CZ_SSSP_LONG_QUEUE");  // (read per call, like CZ_SSSP_DELTA: tests switch the queue off)
            const bool lq_off = lq_env && atoi(lq_env) == 0;
            const uint32_t cap = (want == 0 || want > (1ull << 30) / 20 || lq_off) ? 0u : (uint32_t)want;
            if (cap != lq_cap || d_lq.n != (size_t)cap * 5) {
                if (cap) CZ_HIP(d_lq.alloc((size_t)cap * 5));
                lq_cap = cap;
            }
        }
        // a kept state of the same shape (a resident graph's repeated call): every array, not just the first ones
        if (d_dp.n == SN && d_starts.n == S && d_qtag.n == SN && d_ftag.n == SN && d_q[0].n == SN && d_q[1].n == SN && d_q[2].n == SN &&
            d_q[3].n == SN)
            return CZ_OK;
        CZ_HIP(d_qtag.alloc(SN));
        CZ_HIP(d_ftag.alloc(SN));
        CZ_HIP(d_dp.alloc(SN));
        CZ_HIP(d_starts.alloc(S));
        for (auto &q : d_q) CZ_HIP(q.alloc(SN));
        return CZ_OK;
    }

    // d_misc: [0] seed / split near count, [1] far count, [2] far-next count, [3] min far cost bits, [6] goals not yet settled,
    // [7] the threshold sssp_threshold_kernel worked out, [8..10] the near counters: round r appends under [8 + r % 3], the round
    // after it reads its pile's size there -- by value, or from the device when it was launched ahead -- and every round zeroes
    // [8 + (r + 1) % 3] for the round after it (no memset per round); [11], [12] a launched-ahead round that found its pile beyond
    // its grid: its tag and the pile's size (sssp_relax_kernel); [13], [14] the long-list queue's counters (SsspLongQ), by round parity.
    // Host round trips: a round used to be a launch and one 64-byte copy back (~22 us for a pile of ten nodes: 35 of the 59
    // rounds of the 10M / 100M bench graph are that small, and a threshold move was three such trips).  Now a SMALL pile starts a
    // burst of kBurst rounds, launched back to back, each on the pile size the one before it left on the device (an empty round
    // costs a launch), and the threshold move is four launches and ONE trip.  Results cannot change: the same relaxations, and
    // the packed CAS makes their order irrelevant.  CZ_SSSP_BURST = 0 | 1 .. 16 (default 6; 0: a trip per round, as before).
    // goals (device ids, optional): stop as soon as every goal of every source is settled -- dijkstra()'s early exit when its goal
    // set is exhausted (shortest_path_dijkstra.rs:300-306); settled_bits then holds the threshold below which costs are final
    const uint32_t *d_goals = nullptr;
    uint32_t n_goals = 0, settled_bits = 0x7F800000u;
    int run(const uint32_t *starts, uint32_t ns, const volatile uint8_t *poison) {
        const uint64_t nsN = (uint64_t)ns * N;
        settled_bits = 0x7F800000u;
        trace_mark("run: entry");
        hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(nsN)), dim3(kT), 0, s, d_dp.p, nsN, kInfPacked);
        trace_mark("run: fill dp");
        CZ_HIP(hipMemsetAsync(d_qtag.p, 0, nsN * 4, s));
        CZ_HIP(hipMemsetAsync(d_ftag.p, 0, nsN * 4, s));
        CZ_HIP(hipMemsetAsync(d_misc.p, 0, kMisc * 4, s));
        trace_mark("run: memsets");
        CZ_HIP(hipMemcpyAsync(d_starts.p, starts, (size_t)ns * 4, hipMemcpyHostToDevice, s));
        unsigned long long *near_cur = d_q[0].p, *near_next = d_q[1].p, *far_cur = d_q[2].p, *far_next = d_q[3].p;
        hipLaunchKernelGGL(sssp_seed_kernel, dim3((ns + kT - 1) / kT), dim3(kT), 0, s, d_starts.p, ns, N, d_dp.p, near_cur, d_misc.p);
        uint32_t h[kMisc];
        int rc = read_counters(h);
        if (rc) return rc;
        uint32_t n_near = h[0], n_far = 0, round = 1, phase = 1;
        float thr = one_pile ? INFINITY : delta;
        static const bool trace = getenv("CZ_SSSP_TRACE") != nullptr;  // per-round pile sizes on stderr (scratch/ experiments)
        static const uint32_t burst = [] {
            const char *e = getenv("CZ_SSSP_BURST");
            return e ? (uint32_t)std::min(16, std::max(0, atoi(e))) : 6u;
        }();
        constexpr uint32_t kBurstPile = 2048;  // a pile of at most this many entries starts a burst ...
        constexpr uint32_t kBurstGrid = 8192;  // ... whose later rounds are launched for piles of up to this many (a larger one: left to the host)
        static const uint32_t kFarSearchBelow = [] {  // (CZ_SSSP_FAR_SEARCH_BELOW: piles from this size on move the threshold unasked)
            const char *e = getenv("CZ_SSSP_FAR_SEARCH_BELOW");
            return e ? (uint32_t)strtoul(e, nullptr, 10) : (1u << 16);
        }();
        bool empty_split = false;
        auto near_counter = [&](uint32_t r) { return d_misc.p + 8 + r % 3; };
        auto long_queue = [&](uint32_t r) {  // (counters [13], [14] by round parity: a round zeroes the next one's)
            return SsspLongQ{d_lq.p, lq_cap, lq_cap ? d_misc.p + 13 + (r & 1u) : (uint32_t *)nullptr, d_misc.p + 13 + ((r + 1u) & 1u)};
        };
        for (;;) {
            while (n_near > 0) {
                if (trace) {
                    static thread_local std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
                    const auto t_now = std::chrono::steady_clock::now();
                    fprintf(stderr, "sssp phase %u round %u thr %g near %u far %u  (+%.1f us since the previous round's line)\n", phase, round,
                            (double)thr, n_near, n_far, std::chrono::duration<double, std::micro>(t_now - t_prev).count());
                    t_prev = t_now;
                }
                if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
                uint32_t thr_bits;
                memcpy(&thr_bits, &thr, 4);
                const uint32_t rounds_now = (burst > 1 && n_near <= kBurstPile) ? burst : 1u;
                const uint32_t round0 = round;
                unsigned long long *const cur0 = near_cur, *const next0 = near_next;
                for (uint32_t j = 0; j < rounds_now; j++, round++) {
                    const uint64_t pile = j == 0 ? n_near : kBurstGrid;
                    hipLaunchKernelGGL(sssp_relax_kernel, dim3(grid_for(pile * kSsspLanes)), dim3(kT), 0, s, d_off.p, d_tgt.p, d_w.p, N,
                                       near_cur, n_near, d_dp.p, d_qtag.p, round, d_ftag.p, phase, thr_bits,
                                       SsspQueue{near_next, near_counter(round)}, SsspQueue{far_cur, d_misc.p + 1}, near_counter(round + 1),
                                       j == 0 ? (const uint32_t *)nullptr : (const uint32_t *)near_counter(round - 1), d_misc.p + 11,
                                       long_queue(round));
                    if (lq_cap)  // what the round queued of its long lists, by the whole grid (the count stays on the device)
                        hipLaunchKernelGGL(sssp_relax_long_kernel, dim3(1024), dim3(kT), 0, s, d_tgt.p, d_w.p, N, d_dp.p, d_qtag.p, round,
                                           d_ftag.p, phase, thr_bits, SsspQueue{near_next, near_counter(round)},
                                           SsspQueue{far_cur, d_misc.p + 1}, long_queue(round), (const uint32_t *)(d_misc.p + 11));
                    std::swap(near_cur, near_next);
                }
                if ((rc = read_counters(h))) return rc;
                n_far = h[1];
                if (h[11]) {  // round h[11] found its pile (h[12] entries) beyond its grid and left it alone, like the rounds behind it
                    round = h[11];
                    n_near = h[12];
                    const bool even = ((round - round0) & 1u) == 0;  // (the piles alternate between the two arrays)
                    near_cur = even ? cur0 : next0;
                    near_next = even ? next0 : cur0;
                    CZ_HIP(hipMemsetAsync(d_misc.p + 11, 0, 8, s));
                } else {
                    n_near = h[8 + (round - 1) % 3];
                }
            }
            if (n_far == 0) break;
            if (d_goals && n_goals && !one_pile) {  // every cost below thr is final now: are the goals among them?
                uint32_t thr_now;
                memcpy(&thr_now, &thr, 4);
                CZ_HIP(hipMemsetAsync(d_misc.p + 6, 0, 4, s));
                hipLaunchKernelGGL(sssp_goals_left_kernel, dim3(grid_for((uint64_t)ns * n_goals)), dim3(kT), 0, s, d_dp.p, N, ns, d_goals, n_goals,
                                   thr_now, d_misc.p + 6);
                if ((rc = read_counters(h))) return rc;
                if (h[6] == 0) {
                    settled_bits = thr_now;
                    break;
                }
            }
            // move the threshold to the bucket of the nearest waiting node, then split the far pile: all on the device, one trip.
            // Looking for the nearest of MILLIONS of waiting nodes is a random 8-byte read per entry (0.15 ms at 5M) to learn that
            // the next bucket is not empty: a large pile moves the threshold by one bucket unasked -- and if that bucket did turn
            // out empty (the split below found nothing; costs far apart), the next move searches.
            const bool search_min = n_far < kFarSearchBelow || empty_split;
            if (search_min) {
                CZ_HIP(hipMemsetAsync(d_misc.p + 3, 0xFF, 4, s));
                hipLaunchKernelGGL(sssp_far_min_kernel, dim3(grid_for(n_far)), dim3(kT), 0, s, far_cur, n_far, N, d_dp.p, d_misc.p + 3);
            }
            hipLaunchKernelGGL(sssp_threshold_kernel, dim3(1), dim3(1), 0, s, d_misc.p, thr, delta, search_min ? 1 : 0);
            phase++;
            hipLaunchKernelGGL(sssp_split_kernel, dim3(grid_for(n_far)), dim3(kT), 0, s, far_cur, n_far, N, d_dp.p, d_qtag.p, round,
                               d_ftag.p, phase, 0u, SsspQueue{near_cur, d_misc.p}, SsspQueue{far_next, d_misc.p + 2},
                               (const uint32_t *)(d_misc.p + 7));
            hipLaunchKernelGGL(sssp_far_carry_kernel, dim3(1), dim3(1), 0, s, d_misc.p);
            if ((rc = read_counters(h))) return rc;
            n_near = h[0];
            n_far = h[2];
            empty_split = n_near == 0;
            memcpy(&thr, &h[7], 4);
            std::swap(far_cur, far_next);
            round++;
            if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        }
        trace_mark("run: rounds done");
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sssp launch: %s", hipGetErrorString(e));
        return CZ_OK;
    }
};

int sssp_run(const cz_graph &G, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison,
             bool keep_state, const uint32_t *goals = nullptr, uint32_t n_goals = 0);

}  // namespace

extern "C" int cz_sssp(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N,
                       uint64_t E, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
                       const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, weights, N, E))) return rc;
    return sssp_run(G, starts, n_starts, dist, parent, poison, false);
}

extern "C" int cz_sssp_goals(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                             const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals, float *dist,
                             uint32_t *parent, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    if (n_goals && !goals) return cz::set_error(CZ_E_INVALID, "null goals");
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, weights, N, E))) return rc;
    return sssp_run(G, starts, n_starts, dist, parent, poison, false, goals, n_goals);
}

extern "C" int cz_sssp_goals_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals,
                                float *dist, uint32_t *parent, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (!g) return cz::set_error(CZ_E_INVALID, "null graph");
    if (n_starts == 0 || g->N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    if (n_goals && !goals) return cz::set_error(CZ_E_INVALID, "null goals");
    return sssp_run(*g, starts, n_starts, dist, parent, poison, true, goals, n_goals);
}

extern "C" int cz_sssp_on(const cz_graph *g, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
                          const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (!g) return cz::set_error(CZ_E_INVALID, "null graph");
    if (n_starts == 0 || g->N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    return sssp_run(*g, starts, n_starts, dist, parent, poison, true);
}

namespace {

struct SsspCallState {
    SsspBatch sb;
    cz::PoolBuf<uint32_t> d_parent;
    cz::PoolBuf<float> d_dist;
};

int sssp_run(const cz_graph &G, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison,
             bool keep_state, const uint32_t *goals, uint32_t n_goals) {
    const uint32_t N = G.N;
    int rc = CZ_OK;
    // a resident graph (cz_sssp_on) keeps its state arrays between calls; a one-shot call owns them for its own duration
    std::shared_ptr<SsspCallState> own;
    std::unique_lock<std::mutex> held;
    if (keep_state) held = std::unique_lock<std::mutex>(G.sssp_mu, std::try_to_lock);
    const bool kept = keep_state && held.owns_lock();  // (another thread is inside cz_sssp_on on this handle: own arrays)
    if (kept && G.sssp_state) own = std::static_pointer_cast<SsspCallState>(G.sssp_state);
    else own = std::make_shared<SsspCallState>();
    if (kept) G.sssp_state.reset();  // published again below, once the call has gone through
    SsspBatch &sb = own->sb;
    cz::PoolBuf<uint32_t> &d_parent = own->d_parent;
    cz::PoolBuf<float> &d_dist = own->d_dist;
    auto body = [&]() -> int {
        trace_mark("sssp_run: entry");
        if ((rc = sb.attach(G, n_starts, 80ull << 20))) return rc;  // about 4 GB in all
        const uint64_t SN = (uint64_t)sb.S * N;
        if (d_parent.n != SN) CZ_HIP(d_parent.alloc(SN));
        if (d_dist.n != SN) CZ_HIP(d_dist.alloc(SN));
        hipStream_t s = sb.s;
        cz::PoolBuf<uint32_t> d_goals;
        sb.d_goals = nullptr;
        sb.n_goals = 0;
        if (goals && n_goals) {
            CZ_HIP(d_goals.alloc(n_goals));
            CZ_HIP(hipMemcpyAsync(d_goals.p, goals, (size_t)n_goals * 4, hipMemcpyHostToDevice, s));
            sb.d_goals = d_goals.p;
            sb.n_goals = n_goals;
        }
        struct GoalsOff {  // (the state may be kept for the next call: it must not remember this call's goal array)
            SsspBatch &b;
            ~GoalsOff() {
                b.d_goals = nullptr;
                b.n_goals = 0;
            }
        } goals_off{sb};
        trace_mark("sssp_run: attach + allocs");
        t_timing.lap(T_UPLOAD);
        for (uint32_t s0 = 0; s0 < n_starts; s0 += sb.S) {
            const uint32_t ns = std::min<uint32_t>(sb.S, n_starts - s0);
            const uint64_t nsN = (uint64_t)ns * N;
            if ((rc = sb.run(starts + s0, ns, poison))) return rc;
            hipLaunchKernelGGL(sssp_unpack_flagged_kernel, dim3(grid_for(nsN)), dim3(kT), 0, s, sb.d_dp.p, nsN, d_dist.p, d_parent.p,
                               sb.settled_bits);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sssp launch: %s", hipGetErrorString(e));
            trace_mark("sssp_run: unpack");
            t_timing.lap(T_DEVICE);
            CZ_HIP(hipMemcpy(dist + (size_t)s0 * N, d_dist.p, nsN * 4, hipMemcpyDeviceToHost));
            CZ_HIP(hipMemcpy(parent + (size_t)s0 * N, d_parent.p, nsN * 4, hipMemcpyDeviceToHost));
            trace_mark("sssp_run: download");
            t_timing.lap(T_DOWNLOAD);
        }
        return CZ_OK;
    };
    rc = body();
    // a failed allocation leaves a half-built state behind: it is dropped, not handed to the next call
    if (kept && (rc == CZ_OK || rc == CZ_E_CANCELLED)) G.sssp_state = own;
    return rc;
}

}  // namespace

// ---- ClosenessCentrality (fixed_rule/algos/all_pairs_shortest_path.rs:97-176) ----------------------------------------------
// One cost-only Dijkstra per node (`dijkstra_cost_only`, :146-176: the same strict-`<` f32 relaxation as `dijkstra`), then
// per start, f32 throughout (:118-122): total = the finite distances summed one after the other in node order, nc = their
// count, centrality = nc * nc / total / (n - 1).  The all-sources SSSP leaves a batch's costs on the device; one lane per
// source walks its row in node order (the sum is one f32 chain: its order is the result), so neither the [starts][n]
// distances nor a per-source host loop cross the bus.
namespace {

__global__ void __launch_bounds__(kT)
closeness_kernel(const unsigned long long *__restrict__ dp, uint32_t N, uint32_t ns, double *__restrict__ out) {
    for (uint32_t si = blockIdx.x * blockDim.x + threadIdx.x; si < ns; si += gridDim.x * blockDim.x) {
        const unsigned long long *row = dp + (size_t)si * N;
        float total = 0.0f, nc = 0.0f;
        for (uint32_t v = 0; v < N; v++) {
            const uint32_t c = (uint32_t)(row[v] >> 32);
            if (c != 0x7F800000u) {  // `.filter(|d| d.is_finite())` (the costs are never NaN; +inf = unreached)
                total = total + __uint_as_float(c);
                nc = nc + 1.0f;
            }
        }
        out[si] = (double)(nc * nc / total / (float)(N - 1));
    }
}

}  // namespace

extern "C" int cz_closeness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N, uint64_t E,
                            double *centrality, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (N == 0) return CZ_OK;
    if (!centrality) return cz::set_error(CZ_E_INVALID, "null centrality");
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, weights, N, E))) return rc;
    SsspBatch sb;
    uint64_t pairs = 80ull << 20;
    if (const char *b = getenv("CZ_BC_BATCH")) pairs = std::max<uint64_t>(1, strtoull(b, nullptr, 10)) * N;  // sources per batch (tests)
    if ((rc = sb.attach(G, N, pairs))) return rc;
    cz::DevBuf<double> d_out;
    CZ_HIP(d_out.alloc(sb.S));
    std::vector<uint32_t> all(N);
    for (uint32_t i = 0; i < N; i++) all[i] = i;
    t_timing.lap(T_UPLOAD);
    for (uint32_t s0 = 0; s0 < N; s0 += sb.S) {
        const uint32_t ns = std::min<uint32_t>(sb.S, N - s0);
        if ((rc = sb.run(all.data() + s0, ns, poison))) return rc;
        hipLaunchKernelGGL(closeness_kernel, dim3(grid_for(ns, 64)), dim3(64), 0, sb.s, sb.d_dp.p, N, ns, d_out.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "closeness launch: %s", hipGetErrorString(e));
        t_timing.lap(T_DEVICE);
        CZ_HIP(hipMemcpy(centrality + s0, d_out.p, (size_t)ns * 8, hipMemcpyDeviceToHost));
        t_timing.lap(T_DOWNLOAD);
    }
    return CZ_OK;
}

// ---- BetweennessCentrality (fixed_rule/algos/all_pairs_shortest_path.rs:31-95) --------------------------------------------
// The reference runs dijkstra_keep_ties from every node, enumerates ALL shortest paths to every target and adds 1 / (number
// of shortest paths to that target) to every inner node of every path.  Its back pointers are exactly the "tight" edges
// dist[u] + w == dist[v] (f32), so the same sums come out of path COUNTS over the tight-edge DAG (Brandes):
//   sigma[v] = number of shortest paths s -> v = sum over tight (u, v) of sigma[u]                      (sigma[s] = 1)
//   delta[u] = sum over tight (u, v) of sigma[u] / sigma[v] * (1 + delta[v]);   centrality[u] += delta[u]   (u != s)
// Both recurrences are evaluated level by level of the tight-edge DAG (level = longest tight path from the source): a pair
// enters the next level when the last of its tight predecessors has been counted (a pending count per pair), every pair is
// visited once per recurrence by a 16-lane group pulling over its in- (sigma) or out-adjacency (delta) in a fixed order,
// f64 throughout (the reference adds f32 terms; the parity bar of this rule is 1e-5 against its literal enumeration).
// Round 2's first form swept ALL pairs once per level (3.2 of 3.5 s on a 20 000-node graph).  Sources run in batches on
// the distances the multi-source SSSP above leaves on the device.
namespace {

// number of tight in-edges of every (source, node) pair (what has to be final before the pair's path count is)
__global__ void __launch_bounds__(kT)
bc_tight_in_kernel(const uint32_t *__restrict__ in_off, const uint32_t *__restrict__ in_src, const float *__restrict__ in_w, uint32_t N,
                   uint32_t ns, const unsigned long long *__restrict__ dp, const uint32_t *__restrict__ starts,
                   uint32_t *__restrict__ tin, uint32_t *__restrict__ absorbed) {
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes;
    const uint64_t ngroups = (uint64_t)gridDim.x * blockDim.x / kSsspLanes, total = (uint64_t)ns * N;
    const uint64_t rounds = (total + ngroups - 1) / ngroups;  // every group of a wave runs the same trip count (shuffles)
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t i = group + r * ngroups;
        const bool live = i < total;
        const uint32_t si = live ? (uint32_t)(i / N) : 0, v = live ? (uint32_t)(i % N) : 0;
        const unsigned long long *dps = dp + (size_t)si * N;
        uint32_t c = 0;
        if (live && v != starts[si]) {
            const uint32_t cv = (uint32_t)(dps[v] >> 32);
            if (cv != 0x7F800000u) {
                const uint32_t e1 = in_off[v + 1];
                for (uint32_t e = in_off[v] + glane; e < e1; e += kSsspLanes) {
                    const uint32_t cu = (uint32_t)(dps[in_src[e]] >> 32);
                    if (cu != 0x7F800000u && __float_as_uint(__uint_as_float(cu) + in_w[e]) == cv) {
                        if (cu == cv) atomicAdd(absorbed, 1u);  // dist[u] + w == dist[u]: the tight edges are no DAG, see the host
                        c++;
                    }
                }
            }
        }
#pragma unroll
        for (int o = kSsspLanes / 2; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
        if (live && glane == 0) tin[i] = c;
    }
}

__global__ void __launch_bounds__(kT)
bc_seed_kernel(const uint32_t *__restrict__ starts, uint32_t ns, uint32_t N, unsigned long long *__restrict__ order, double *__restrict__ sigma) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    order[i] = ((unsigned long long)i << 32) | starts[i];
    sigma[(size_t)i * N + starts[i]] = 1.0;
}

// one level of the tight-edge DAG (level = longest tight path from the source): the pairs order[lo, lo + fsize) have all their
// tight predecessors in earlier levels.  A 16-lane group per pair: path count by a pull over the tight in-edges (fixed order;
// the counts are integers, exact in f64 up to 2^53), then every tight out-edge takes one off its target's pending count, and the
// pair that takes the last one appends the target to the next level.
__global__ void __launch_bounds__(kT)
bc_sigma_level_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                      const uint32_t *__restrict__ in_off, const uint32_t *__restrict__ in_src, const float *__restrict__ in_w, uint32_t N,
                      const unsigned long long *__restrict__ dp, const uint32_t *__restrict__ starts, uint32_t lo, uint32_t fsize,
                      double *__restrict__ sigma, uint32_t *__restrict__ tin, SsspQueue order) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;
    __shared__ StagedPile st;
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < fsize;
        const unsigned long long ent = live ? order.items[lo + i] : 0ull;
        const uint32_t si = (uint32_t)(ent >> 32), v = (uint32_t)ent;
        const unsigned long long *dps = dp + (size_t)si * N;
        double *sg = sigma + (size_t)si * N;
        const uint32_t cv = live ? (uint32_t)(dps[v] >> 32) : 0;
        double sum = 0.0;
        const bool is_start = live && v == starts[si];
        if (live && !is_start) {
            const uint32_t e1 = in_off[v + 1];
            for (uint32_t e = in_off[v] + glane; e < e1; e += kSsspLanes) {
                const uint32_t u = in_src[e];
                const uint32_t cu = (uint32_t)(dps[u] >> 32);
                if (cu != 0x7F800000u && __float_as_uint(__uint_as_float(cu) + in_w[e]) == cv) sum += sg[u];
            }
        }
#pragma unroll
        for (int o = kSsspLanes / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (live && !is_start && glane == 0) sg[v] = sum;
        const uint32_t e0 = live ? off[v] : 0, e1 = live ? off[v + 1] : 0;
        const float dv = __uint_as_float(cv);
        uint32_t maxlen = e1 - e0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
        for (uint32_t b = 0; b < maxlen; b += kSsspLanes) {
            const uint32_t e = e0 + b + glane;
            bool ready = false;
            uint32_t x = 0;
            if (e < e1) {
                x = tgt[e];
                if (__float_as_uint(dv + w[e]) == (uint32_t)(dps[x] >> 32)) ready = atomicSub(&tin[(size_t)si * N + x], 1u) == 1u;
            }
            staged_push(order, st, ready, ((unsigned long long)si << 32) | x, lane);
        }
        if ((r & 7) == 7 || r + 1 == rounds) staged_flush(order, st);
    }
}

// the levels again, last one first: the dependency of a pair by a pull over its tight out-edges (all in later levels: final)
__global__ void __launch_bounds__(kT)
bc_delta_level_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t N,
                      const unsigned long long *__restrict__ dp, const unsigned long long *__restrict__ order, uint32_t fsize,
                      const double *__restrict__ sigma, double *__restrict__ delta) {
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < fsize;
        const unsigned long long ent = live ? order[i] : 0ull;
        const uint32_t si = (uint32_t)(ent >> 32), u = (uint32_t)ent;
        const unsigned long long *dps = dp + (size_t)si * N;
        const double *sg = sigma + (size_t)si * N;
        double *dl = delta + (size_t)si * N;
        double sum = 0.0;
        if (live) {
            const double su = sg[u];
            const float du = __uint_as_float((uint32_t)(dps[u] >> 32));
            const uint32_t e1 = off[u + 1];
            for (uint32_t e = off[u] + glane; e < e1; e += kSsspLanes) {
                const uint32_t v = tgt[e];
                if (__float_as_uint(du + w[e]) == (uint32_t)(dps[v] >> 32)) sum += su / sg[v] * (1.0 + dl[v]);
            }
        }
#pragma unroll
        for (int o = kSsspLanes / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (live && glane == 0) dl[u] = sum;
    }
}

__global__ void __launch_bounds__(kT)
bc_accumulate_kernel(uint32_t N, uint32_t ns, const double *__restrict__ delta, const uint32_t *__restrict__ starts,
                     double *__restrict__ cent) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        double c = cent[v];
        for (uint32_t si = 0; si < ns; si++)
            if (starts[si] != v) c += delta[(size_t)si * N + v];
        cent[v] = c;
    }
}

}  // namespace

extern "C" int cz_betweenness(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N,
                              uint64_t E, double *centrality, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (N == 0) return CZ_OK;
    if (!centrality) return cz::set_error(CZ_E_INVALID, "null centrality");
    rc = check_csr(out_offsets, out_targets, N, E);
    if (rc) return rc;
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    for (uint64_t e = 0; e < E; e++)
        if (!(weights[e] > 0.0f))
            return cz::set_error(CZ_E_UNSUPPORTED, "edge %llu has weight %g: the device path of BetweennessCentrality needs positive weights",
                                 (unsigned long long)e, (double)weights[e]);
    // the transposed graph (sigma pulls over in-edges), in-lists in ascending source order (a counting sort)
    std::vector<uint32_t> in_off((size_t)N + 1, 0), in_src(E);
    std::vector<float> in_w(E);
    for (uint64_t e = 0; e < E; e++) {
        if (out_targets[e] >= N) return cz::set_error(CZ_E_INVALID, "target %u out of range", out_targets[e]);
        in_off[out_targets[e] + 1]++;
    }
    for (uint32_t v = 0; v < N; v++) in_off[v + 1] += in_off[v];
    {
        std::vector<uint32_t> cur(in_off.begin(), in_off.end() - 1);
        for (uint32_t u = 0; u < N; u++)
            for (uint32_t e = out_offsets[u]; e < out_offsets[u + 1]; e++) {
                const uint32_t at = cur[out_targets[e]]++;
                in_src[at] = u;
                in_w[at] = weights[e];
            }
    }
    SsspBatch sb;
    // 64 bytes per (source, node) here: 48 of the SSSP + sigma and delta -> a smaller batch than cz_sssp's
    std::vector<uint32_t> all(N);
    for (uint32_t i = 0; i < N; i++) all[i] = i;
    uint64_t pairs = 48ull << 20;
    if (const char *b = getenv("CZ_BC_BATCH")) pairs = std::max<uint64_t>(1, strtoull(b, nullptr, 10)) * N;  // sources per batch (tests)
    cz_graph G;
    if ((rc = graph_fill(G, out_offsets, out_targets, weights, N, E))) return rc;
    if ((rc = sb.attach(G, N, pairs))) return rc;
    const uint64_t SN = (uint64_t)sb.S * N;
    cz::DevBuf<uint32_t> d_ioff, d_isrc, d_flags;
    cz::DevBuf<float> d_iw;
    cz::DevBuf<double> d_sig, d_del, d_cent;
    CZ_HIP(d_ioff.alloc((size_t)N + 1));
    CZ_HIP(d_isrc.alloc(E));
    CZ_HIP(d_iw.alloc(E));
    CZ_HIP(d_flags.alloc(2));
    CZ_HIP(d_cent.alloc(N));
    CZ_HIP(d_sig.alloc(SN));
    CZ_HIP(d_del.alloc(SN));
    CZ_HIP(hipMemcpy(d_ioff.p, in_off.data(), ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) {
        CZ_HIP(hipMemcpy(d_isrc.p, in_src.data(), E * 4, hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(d_iw.p, in_w.data(), E * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = sb.s;
    t_timing.lap(T_UPLOAD);
    CZ_HIP(hipMemsetAsync(d_cent.p, 0, (size_t)N * 8, s));
    std::vector<uint32_t> level_lo;
    for (uint32_t s0 = 0; s0 < N; s0 += sb.S) {
        const uint32_t ns = std::min<uint32_t>(sb.S, N - s0);
        const uint64_t nsN = (uint64_t)ns * N;
        if ((rc = sb.run(all.data() + s0, ns, poison))) return rc;
        // the SSSP is done with its round tags and queues: the tags hold the pending tight in-edges of every pair, the first
        // queue the pairs in level order
        uint32_t *tin = sb.d_qtag.p;
        SsspQueue order{sb.d_q[0].p, d_flags.p};
        uint32_t h[2] = {ns, 0};
        CZ_HIP(hipMemcpy(d_flags.p, h, 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(bc_tight_in_kernel, dim3(grid_for(nsN * kSsspLanes)), dim3(kT), 0, s, d_ioff.p, d_isrc.p, d_iw.p, N, ns, sb.d_dp.p,
                           sb.d_starts.p, tin, d_flags.p + 1);
        CZ_HIP(hipMemsetAsync(d_sig.p, 0, nsN * 8, s));
        CZ_HIP(hipMemsetAsync(d_del.p, 0, nsN * 8, s));
        hipLaunchKernelGGL(bc_seed_kernel, dim3((ns + kT - 1) / kT), dim3(kT), 0, s, sb.d_starts.p, ns, N, order.items, d_sig.p);
        CZ_HIP(hipMemcpy(h, d_flags.p, 8, hipMemcpyDeviceToHost));
        if (h[1])
            return cz::set_error(CZ_E_UNSUPPORTED, "BetweennessCentrality: an edge weight is absorbed by the f32 path cost "
                                                   "(dist[u] + w == dist[u]); shortest-path counts are not defined on such a graph");
        level_lo.clear();
        uint32_t lo = 0, fsize = ns;
        while (fsize > 0) {
            if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
            level_lo.push_back(lo);
            hipLaunchKernelGGL(bc_sigma_level_kernel, dim3(grid_for((uint64_t)fsize * kSsspLanes)), dim3(kT), 0, s, sb.d_off.p, sb.d_tgt.p,
                               sb.d_w.p, d_ioff.p, d_isrc.p, d_iw.p, N, sb.d_dp.p, sb.d_starts.p, lo, fsize, d_sig.p, tin, order);
            uint32_t end = 0;
            CZ_HIP(hipMemcpy(&end, d_flags.p, 4, hipMemcpyDeviceToHost));
            lo += fsize;
            fsize = end - lo;
            if (level_lo.size() > (size_t)N + 1) return cz::set_error(CZ_E_HIP, "internal: the tight edges have no level order");
        }
        level_lo.push_back(lo);
        for (size_t k = level_lo.size() - 1; k-- > 0;) {
            const uint32_t l0 = level_lo[k], n = level_lo[k + 1] - l0;
            hipLaunchKernelGGL(bc_delta_level_kernel, dim3(grid_for((uint64_t)n * kSsspLanes)), dim3(kT), 0, s, sb.d_off.p, sb.d_tgt.p, sb.d_w.p,
                               N, sb.d_dp.p, order.items + l0, n, d_sig.p, d_del.p);
        }
        hipLaunchKernelGGL(bc_accumulate_kernel, dim3(grid_for(N)), dim3(kT), 0, s, N, ns, d_del.p, sb.d_starts.p, d_cent.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "betweenness launch: %s", hipGetErrorString(e));
    }
    t_timing.lap(T_DEVICE);
    CZ_HIP(hipMemcpy(centrality, d_cent.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    t_timing.lap(T_DOWNLOAD);
    return CZ_OK;
}

// ---- LabelPropagation (fixed_rule/algos/label_propagation.rs:56-109), one FIXED execution of it --------------------------------
// The reference visits the nodes in a freshly shuffled order every iteration (:63-66), updates labels in place, and picks a
// random label among the best-scored ones (:85): no two runs agree, so there is no result to be identical to.  This rule
// fixes both choices so that the run is (a) one the reference itself can produce and (b) parallel:
//   * order: the colour classes of a deterministic colouring, ascending, ids ascending inside a class.  In round r every
//     still uncoloured node whose key (hash(id) << 32 | id) is the largest among its uncoloured neighbours -- edges in EITHER
//     direction -- takes colour r.  Nodes of one class share no edge, so updating a class at once is the sequential loop over
//     its nodes in any order: one kernel launch per class.
//   * tie: the smallest label among those whose score == the largest score (f32::total_cmp order, :81-84).
// Scores are what :70-73 computes: per label, the f32 sum of the edge values in adjacency order, starting from 0.0 -- a wave
// per node walks the list 64 entries at a time and adds each label's entries one after the other.
namespace {

constexpr uint32_t kLpTable = 512;        // per-wave LDS table; nodes of degree <= kLpSmall never fill it beyond 3/4
constexpr uint32_t kLpSmall = 384;
constexpr uint32_t kLpColourLong = 1024;  // the colouring: list entries past this one are walked by the whole grid
constexpr uint32_t kLpTiny = 32;          // up to here a 16-lane group keeps a node's whole list in registers (two entries a lane)

__device__ __forceinline__ unsigned long long lp_priority(uint32_t v) {
    uint32_t x = v;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return ((unsigned long long)x << 32) | v;
}

// The colouring in O(edges): a node takes its colour in the first round in which every neighbour of higher key has one, i.e.
// colour(v) = 1 + the largest colour among its higher neighbours (0 without any).  pending[v] counts those neighbours (one per
// edge occurrence, either direction); a node that gets its colour takes one off every lower neighbour, and whoever takes the
// last one off gives that neighbour the next colour and appends it to the next round's list.  (The first device form re-scanned
// the lists of every still uncoloured node in every round: 110 of the rule's 209 ms on the 10M / 200M graph.)
__global__ void __launch_bounds__(kT)
lp_pending_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ in_off,
                  const uint32_t *__restrict__ in_src, uint32_t N, uint32_t *__restrict__ pending, uint32_t *__restrict__ colour,
                  QueueT<uint32_t> first) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    const uint32_t rounds = (N + ngroups - 1) / ngroups;  // every group of the grid runs the same trip count (shuffles, barriers)
    __shared__ StagedPileT<uint32_t> st;
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t v = group + r * ngroups;
        const bool live = v < N;
        uint32_t c = 0;
        if (live) {
            const unsigned long long kv = lp_priority(v);
            for (int side = 0; side < (in_off ? 2 : 1); side++) {  // (no transposed adjacency: the graph is symmetric, one side is all)
                const uint32_t *o = side ? in_off : off, *t = side ? in_src : tgt;
                const uint32_t e1 = o[v + 1];
                for (uint32_t e = o[v] + glane; e < e1; e += kSsspLanes) {
                    const uint32_t u = t[e];
                    if (u != v && lp_priority(u) > kv) c++;
                }
            }
        }
#pragma unroll
        for (int o = kSsspLanes / 2; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
        if (live && glane == 0) {
            pending[v] = c;
            colour[v] = c ? CZ_NONE : 0u;
        }
        staged_push(first, st, live && glane == 0 && c == 0, v, lane);
        if ((r & 7) == 7 || r + 1 == rounds) staged_flush(first, st);
    }
}

__global__ void __launch_bounds__(kT)
lp_colour_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ in_off,
                 const uint32_t *__restrict__ in_src, const uint32_t *__restrict__ list, uint32_t n_list, uint32_t next_colour,
                 uint32_t *__restrict__ pending, uint32_t *__restrict__ colour, QueueT<uint32_t> next, QueueT<uint32_t> long_nodes) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    const uint32_t rounds = (n_list + ngroups - 1) / ngroups;
    __shared__ StagedPileT<uint32_t> st;
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < n_list;
        const uint32_t v = live ? list[i] : 0;
        const unsigned long long kv = lp_priority(v);
        uint32_t len[2] = {0, 0}, beg[2] = {0, 0};
        if (live) {
            beg[0] = off[v];
            len[0] = off[v + 1] - beg[0];
            if (in_off) {
                beg[1] = in_off[v];
                len[1] = in_off[v + 1] - beg[1];
            }
            // a long list: its first kLpColourLong entries here, the rest by every workgroup of lp_colour_long_kernel (an R-MAT hub
            // of 215 000 entries kept one 16-lane group busy for 5 ms, and the hubs take their colours one round after the other)
            if (glane == 0 && max(len[0], len[1]) > kLpColourLong) long_nodes.items[atomicAdd(long_nodes.count, 1u)] = v;
            len[0] = min(len[0], kLpColourLong);
            len[1] = min(len[1], kLpColourLong);
        }
        for (int side = 0; side < (in_off ? 2 : 1); side++) {
            const uint32_t *t = side ? in_src : tgt;
            uint32_t maxlen = len[side];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
            for (uint32_t b = 0; b < maxlen; b += kSsspLanes) {
                bool ready = false;
                uint32_t u = 0;
                if (b + glane < len[side]) {
                    u = t[beg[side] + b + glane];
                    if (u != v && lp_priority(u) < kv && atomicSub(&pending[u], 1u) == 1u) {
                        ready = true;
                        colour[u] = next_colour;
                    }
                }
                staged_push(next, st, ready, u, lane);
            }
        }
        if ((r & 7) == 7 || r + 1 == rounds) staged_flush(next, st);
    }
}

// the lists lp_colour_kernel set aside, from entry kLpColourLong on: every workgroup takes stretches of every one of them
__global__ void __launch_bounds__(kT)
lp_colour_long_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ in_off,
                      const uint32_t *__restrict__ in_src, QueueT<uint32_t> long_nodes, uint32_t next_colour, uint32_t *__restrict__ pending,
                      uint32_t *__restrict__ colour, QueueT<uint32_t> next) {
    const int lane = threadIdx.x & 63;
    __shared__ StagedPileT<uint32_t> st;
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
    const uint32_t n_long = *long_nodes.count;
    uint32_t since_flush = 0;
    for (uint32_t q = 0; q < n_long; q++) {
        const uint32_t v = long_nodes.items[q];
        const unsigned long long kv = lp_priority(v);
        for (int side = 0; side < (in_off ? 2 : 1); side++) {
            const uint32_t *t = side ? in_src : tgt;
            const uint32_t beg = side ? in_off[v] : off[v], len = (side ? in_off[v + 1] : off[v + 1]) - beg;
            const uint32_t first_wg = (q * 37u + (uint32_t)side * 11u) % gridDim.x;  // (the stretches of different lists start at different workgroups)
            const uint32_t my = (blockIdx.x + gridDim.x - first_wg) % gridDim.x;
            for (uint32_t b = kLpColourLong + my * kT; b < len; b += gridDim.x * kT) {  // (uniform over the workgroup)
                bool ready = false;
                uint32_t u = 0;
                if (b + threadIdx.x < len) {
                    u = t[beg + b + threadIdx.x];
                    if (u != v && lp_priority(u) < kv && atomicSub(&pending[u], 1u) == 1u) {
                        ready = true;
                        colour[u] = next_colour;
                    }
                }
                staged_push(next, st, ready, u, lane);
                if ((++since_flush & 7u) == 0) staged_flush(next, st);
            }
        }
    }
    staged_flush(next, st);
}

__global__ void __launch_bounds__(kT)
lp_indegree_kernel(const uint32_t *__restrict__ tgt, uint64_t E, uint32_t N, uint32_t *__restrict__ cnt, uint32_t *__restrict__ bad) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = tgt[e];
        if (v < N) atomicAdd(&cnt[v], 1u);
        else *bad = 1;
    }
}

__global__ void __launch_bounds__(kT)
lp_transpose_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, const uint32_t *__restrict__ in_off,
                    uint32_t *__restrict__ cursor, uint32_t *__restrict__ in_src) {
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    for (uint32_t u = group; u < N; u += ngroups) {
        const uint32_t e1 = off[u + 1];
        for (uint32_t e = off[u] + glane; e < e1; e += kSsspLanes) {
            const uint32_t v = tgt[e];
            if (v < N) in_src[in_off[v] + atomicAdd(&cursor[v], 1u)] = u;
        }
    }
}

// bucket = 3 * colour + (0: degree <= kLpTiny, 1: <= kLpSmall, 2: above).  SCATTER = false: counts per bucket; true: order[] filled, `cursor` holding the
// buckets' first positions.  Per chunk of blockDim nodes: LDS counts (which also give every node its rank inside the chunk),
// then one global atomicAdd per bucket the chunk touched.
constexpr uint32_t kLpBuckets = 2048;

template <bool SCATTER>
__global__ void __launch_bounds__(kT)
lp_bucket_kernel(const uint32_t *__restrict__ colour, const uint32_t *__restrict__ off, uint32_t N, uint32_t nb, uint32_t *__restrict__ cursor,
                 uint32_t *__restrict__ order) {
    __shared__ uint32_t lcnt[kLpBuckets], lbase[kLpBuckets];
    const uint32_t total = gridDim.x * blockDim.x;
    const uint32_t rounds = (N + total - 1) / total;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lcnt[i] = 0;
        __syncthreads();
        const uint32_t v = r * total + blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t b = 0, mine = 0;
        if (v < N) {
            const uint32_t deg = off[v + 1] - off[v];
            b = 3 * colour[v] + (deg > kLpSmall ? 2u : deg > kLpTiny ? 1u : 0u);
            mine = atomicAdd(&lcnt[b], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x)
            if (lcnt[i]) lbase[i] = atomicAdd(&cursor[i], lcnt[i]);
        __syncthreads();
        if (SCATTER && v < N) order[lbase[b] + mine] = v;
        __syncthreads();
    }
}

struct LpTab {
    uint32_t *keys;   // CZ_NONE = empty
    float *vals;
    uint32_t *slots;  // the slots in use, in order of first use
    uint32_t bits;
};

// Which nodes an iteration has to look at.  A node's new label is a function of its neighbours' labels, so a node none of whose
// neighbours changed since it was last evaluated keeps its label: evaluating it again is the reference's work, not its result.
//   sparse == 0 (the first iterations: nearly everything changes): every node is evaluated and notes in chg[v] whether its label
//     changed; once an iteration changed few nodes, lp_mark_kernel turns those notes into dirty[] bytes for their dependants.
//   sparse == 1: a node is evaluated only if dirty[v]; it clears its byte first and, when its label changes, sets the byte of
//     every node that has it in its list (moff / mtgt: the out-lists themselves on a symmetric adjacency, else the transposed
//     lists) -- classes later in this iteration and everything in the next see it.
// The nodes of a class share no edge, so nobody writes dirty[v] while v's own group is at work.
struct LpActive {
    uint8_t *dirty, *chg;
    const uint32_t *moff, *mtgt;
    int sparse;
};

// one node, by one wave (all lanes in the same control flow; table accesses are wave-uniform: every lane reads and writes
// the same words with the same values)
__device__ __forceinline__ void lp_update_node(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                                               uint32_t v, uint32_t *__restrict__ labels, const LpTab &tab, int lane,
                                               uint32_t *__restrict__ flags, const LpActive &act) {
    const uint32_t a = off[v], b = off[v + 1];
    if (a == b) return;  // :74-76
    if (act.sparse) {
        if (!act.dirty[v]) return;
        if (lane == 0) act.dirty[v] = 0;
    }
    const uint32_t tmask = (1u << tab.bits) - 1u;
    uint32_t used = 0;
    for (uint32_t base = a; base < b; base += 64) {
        const uint32_t e = base + lane;
        const bool valid = e < b;
        const uint32_t l = valid ? labels[tgt[e]] : CZ_NONE;
        const float wv = valid ? w[e] : 0.f;
        unsigned long long remaining = __ballot(valid);
        while (remaining) {
            const int leader = __ffsll((long long)remaining) - 1;
            const uint32_t L = (uint32_t)__shfl((int)l, leader, 64);
            unsigned long long m = __ballot(valid && l == L);
            remaining &= ~m;
            uint32_t h = (L * 0x9E3779B1u) >> (32 - tab.bits);
            float sum = 0.0f;  // `entry(label).or_default()`
            for (;;) {
                const uint32_t k = tab.keys[h];
                if (k == L) {
                    sum = tab.vals[h];
                    break;
                }
                if (k == CZ_NONE) {
                    tab.keys[h] = L;
                    tab.slots[used++] = h;
                    break;
                }
                h = (h + 1) & tmask;
            }
            while (m) {  // `+= edge.value`, one entry after the other in adjacency order (:72)
                const int j = __ffsll((long long)m) - 1;
                sum += __shfl(wv, j, 64);
                m &= m - 1;
            }
            tab.vals[h] = sum;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
    // :77-85: the largest score under total_cmp; among the labels whose score == it, the smallest
    uint32_t best_key = 0;
    for (uint32_t i = lane; i < used; i += 64) {
        const uint32_t bts = __float_as_uint(tab.vals[tab.slots[i]]);
        const uint32_t key = (bts & 0x80000000u) ? ~bts : (bts | 0x80000000u);
        best_key = max(best_key, key);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best_key = max(best_key, (uint32_t)__shfl_xor((int)best_key, o, 64));
    const float max_score = __uint_as_float((best_key & 0x80000000u) ? (best_key & 0x7FFFFFFFu) : ~best_key);
    uint32_t new_label = CZ_NONE;
    for (uint32_t i = lane; i < used; i += 64) {
        const uint32_t sl = tab.slots[i];
        if (tab.vals[sl] == max_score) new_label = min(new_label, tab.keys[sl]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) new_label = min(new_label, (uint32_t)__shfl_xor((int)new_label, o, 64));
    for (uint32_t i = lane; i < used; i += 64) tab.keys[tab.slots[i]] = CZ_NONE;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const bool changed = new_label != CZ_NONE && new_label != labels[v];  // (every lane reads the same word)
    if (lane == 0) {
        if (new_label == CZ_NONE) flags[1] = 1;  // the best score is NaN: `choose` on an empty list, the reference panics
        else if (changed) {
            labels[v] = new_label;
            flags[0] = 1;
        }
        if (!act.sparse) act.chg[v] = changed;
    }
    if (act.sparse && changed)
        for (uint32_t e = act.moff[v] + lane; e < act.moff[v + 1]; e += 64) act.dirty[act.mtgt[e]] = 1;
}

// the nodes of one colour class with at most kLpTiny neighbours: a 16-lane group per node, the list in registers.  A label's
// score is the sum of its edges' weights IN ADJACENCY ORDER starting from 0.0 (`*entry(label).or_default() += weight`, :72):
// every lane walks the whole list in that order and adds up the entries that carry its own label -- the same additions in
// the same order as the table form, without the table (which serialises a wave per distinct label: 10 M nodes of ~10
// neighbours spent 50 of the rule's 80 ms there).
// One node of the tiny kernel as its 16-lane group holds it.  The loads of a node form a chain of four dependent accesses
// (order -> offsets -> targets -> labels), every link at a random place: a group can work on U nodes at once so that the
// chains overlap.  It does not pay (see tiny_u at the launch): the rule sits at the request rate of the memory system.
struct LpTinyNode {
    uint32_t v, a, b, l0, l1;
    float w0, w1;
    bool live;
};
template <int U>
__global__ void __launch_bounds__(kT)
lp_update_tiny_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                      const uint32_t *__restrict__ order, uint32_t count, uint32_t *__restrict__ labels, uint32_t *__restrict__ flags,
                      LpActive act) {
    constexpr int GL = 16;
    const int glane = threadIdx.x & (GL - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / GL, ngroups = gridDim.x * blockDim.x / GL;
    const uint32_t rounds = (count + ngroups * U - 1) / (ngroups * U);  // every group of the grid runs the same trip count (shuffles)
    for (uint32_t r = 0; r < rounds; r++) {
        LpTinyNode nd[U];
#pragma unroll
        for (int u = 0; u < U; u++) {  // (the loads of the U nodes: nothing here waits for another node's data)
            const uint32_t i = group + (r * U + u) * ngroups;
            LpTinyNode &q = nd[u];
            q.v = i < count ? order[i] : 0;
            q.live = i < count && (!act.sparse || act.dirty[q.v]);  // (the same for the 16 lanes of a group)
            q.a = q.live ? off[q.v] : 0;
            q.b = q.live ? off[q.v + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            LpTinyNode &q = nd[u];
            const bool v0 = q.a + glane < q.b, v1 = q.a + GL + glane < q.b;
            const uint32_t t0 = v0 ? tgt[q.a + glane] : 0, t1 = v1 ? tgt[q.a + GL + glane] : 0;
            q.w0 = v0 ? w[q.a + glane] : 0.f;
            q.w1 = v1 ? w[q.a + GL + glane] : 0.f;
            q.l0 = t0;
            q.l1 = t1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            LpTinyNode &q = nd[u];
            const bool v0 = q.a + glane < q.b, v1 = q.a + GL + glane < q.b;
            q.l0 = v0 ? labels[q.l0] : CZ_NONE;
            q.l1 = v1 ? labels[q.l1] : CZ_NONE;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const LpTinyNode &q = nd[u];
            const uint32_t v = q.v, a = q.a, b = q.b, l0 = q.l0, l1 = q.l1;
            const float w0 = q.w0, w1 = q.w1;
            const bool live = q.live, v0 = a + glane < b, v1 = a + GL + glane < b;
            float s0 = 0.0f, s1 = 0.0f;
            const uint32_t n0 = min(b - a, (uint32_t)GL);
#pragma unroll
            for (int j = 0; j < GL; j++) {
                const uint32_t lj = (uint32_t)__shfl((int)l0, j, GL);
                const float wj = __shfl(w0, j, GL);
                if ((uint32_t)j < n0) {
                    if (lj == l0) s0 += wj;
                    if (lj == l1) s1 += wj;
                }
            }
            if (__any(b - a > (uint32_t)GL)) {
                const uint32_t n1 = b - a > (uint32_t)GL ? b - a - GL : 0;
#pragma unroll
                for (int j = 0; j < GL; j++) {
                    const uint32_t lj = (uint32_t)__shfl((int)l1, j, GL);
                    const float wj = __shfl(w1, j, GL);
                    if ((uint32_t)j < n1) {
                        if (lj == l0) s0 += wj;
                        if (lj == l1) s1 += wj;
                    }
                }
            }
            // :77-85: the largest score under total_cmp; among the labels whose score == it, the smallest
            auto key_of = [](float f) {
                const uint32_t bts = __float_as_uint(f);
                return (bts & 0x80000000u) ? ~bts : (bts | 0x80000000u);
            };
            uint32_t best_key = max(v0 ? key_of(s0) : 0u, v1 ? key_of(s1) : 0u);
#pragma unroll
            for (int o = GL / 2; o >= 1; o >>= 1) best_key = max(best_key, (uint32_t)__shfl_xor((int)best_key, o, GL));
            const float max_score = __uint_as_float((best_key & 0x80000000u) ? (best_key & 0x7FFFFFFFu) : ~best_key);
            uint32_t new_label = CZ_NONE;
            if (v0 && s0 == max_score) new_label = l0;
            if (v1 && s1 == max_score) new_label = min(new_label, l1);
#pragma unroll
            for (int o = GL / 2; o >= 1; o >>= 1) new_label = min(new_label, (uint32_t)__shfl_xor((int)new_label, o, GL));
            const bool changed = live && a != b && new_label != CZ_NONE && new_label != labels[v];
            if (live && glane == 0) {
                if (act.sparse) act.dirty[v] = 0;
                if (a != b) {  // (no neighbours: the node keeps its label, :74-76)
                    if (new_label == CZ_NONE) flags[1] = 1;  // the best score is NaN: `choose` on an empty list, the reference panics
                    else if (changed) {
                        labels[v] = new_label;
                        flags[0] = 1;
                    }
                    if (!act.sparse) act.chg[v] = changed;
                }
            }
            if (act.sparse && changed)
                for (uint32_t e = act.moff[v] + glane; e < act.moff[v + 1]; e += GL) act.dirty[act.mtgt[e]] = 1;
        }
    }
}

// how many nodes the iteration changed (its chg[] notes)
__global__ void __launch_bounds__(kT)
lp_count_changed_kernel(const uint8_t *__restrict__ chg, uint32_t N, uint32_t *__restrict__ out) {
    uint32_t mine = 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) mine += chg[v];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// the dependants of every node the iteration changed become dirty (a 16-lane group per node)
__global__ void __launch_bounds__(kT)
lp_mark_kernel(const uint8_t *__restrict__ chg, uint32_t N, const uint32_t *__restrict__ moff, const uint32_t *__restrict__ mtgt,
               uint8_t *__restrict__ dirty) {
    constexpr uint32_t GL = 16;
    const uint32_t glane = threadIdx.x & (GL - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / GL, ngroups = gridDim.x * blockDim.x / GL;
    for (uint32_t v = group; v < N; v += ngroups) {
        if (!chg[v]) continue;
        for (uint32_t e = moff[v] + glane; e < moff[v + 1]; e += GL) dirty[mtgt[e]] = 1;
    }
}

// the nodes order[0 .. count) of one colour class, degree <= kLpSmall: tables in LDS
__global__ void __launch_bounds__(kT)
lp_update_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                 const uint32_t *__restrict__ order, uint32_t count, uint32_t *__restrict__ labels, uint32_t *__restrict__ flags,
                 LpActive act) {
    __shared__ uint32_t keys[kT / 64][kLpTable];
    __shared__ float vals[kT / 64][kLpTable];
    __shared__ uint32_t slots[kT / 64][kLpTable];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = lane; i < kLpTable; i += 64) keys[wv][i] = CZ_NONE;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    LpTab tab{keys[wv], vals[wv], slots[wv], 9};
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < count; i += n_waves) lp_update_node(off, tgt, w, order[i], labels, tab, lane, flags, act);
}

// The class's nodes of larger degree: a WORKGROUP per node, a table per workgroup in global memory sized for the largest degree
// of the graph.  (Until round 6 a wave per node took the list 64 entries at a time and served the labels among them one after the
// other, every one a chain of table accesses at wave-uniform addresses: ~2 us per label, 0.4 s for ONE 215 000-entry list whose
// neighbours all carry different labels -- and on an R-MAT graph the large nodes sit alone in their colour classes, one after the
// other: 13.4 s for 5 iterations over 10M / 200M.)
// A label's score is still the f32 sum of its entries IN ADJACENCY ORDER: the list is taken kLpHubChunk entries at a time, a
// chunk is sorted by label (rocPRIM's block radix sort: stable, so equal labels stay in list order), and the thread that holds
// the first entry of a run of equal labels fetches the label's running score from the table (or enters the label with 0.0), adds
// the run entry by entry, and writes it back.  Runs are disjoint labels: they go side by side; a chunk costs its longest run.
// Per chunk (thread 0's clock, R-MAT 10M / 200M): load + label gather 3 us, sort 10 us, table + sums 9..14 us; the final pick
// 3.5 us per node.  The rule on that graph: 13.4 s -> 1.87 s (this kernel 1.06 s of it, in 13 815 launches: the colour
// classes of an R-MAT graph are 2 763 and the large nodes come one or two per class).
constexpr uint32_t kLpHubThreads = 1024, kLpHubItems = 4, kLpHubChunk = kLpHubThreads * kLpHubItems;

__global__ void __launch_bounds__(kLpHubThreads)
lp_update_hub_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                     const uint32_t *__restrict__ order, uint32_t count, uint32_t *__restrict__ labels, uint32_t *__restrict__ flags,
                     uint32_t *__restrict__ tkeys, float *__restrict__ tvals, uint32_t *__restrict__ tslots, uint32_t bits,
                     uint32_t label_bits, LpActive act) {
    using Sort = rocprim::block_radix_sort<uint32_t, kLpHubThreads, kLpHubItems, float>;
    __shared__ typename Sort::storage_type sort_storage;
    __shared__ uint32_t sl[kLpHubChunk + 1];
    __shared__ float sw[kLpHubChunk];
    __shared__ uint32_t red[kLpHubThreads / 64];
    __shared__ uint32_t used_sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const size_t at = (size_t)blockIdx.x << bits;
    uint32_t *keys = tkeys + at, *slots = tslots + at;
    float *vals = tvals + at;
    const uint32_t tmask = (1u << bits) - 1u;
    auto block_reduce = [&](uint32_t x, bool is_max) {  // every thread gets the maximum / minimum over the workgroup
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t y = (uint32_t)__shfl_xor((int)x, o, 64);
            x = is_max ? max(x, y) : min(x, y);
        }
        __syncthreads();  // (red[] may still be read from the last call)
        if (lane == 0) red[wv] = x;
        __syncthreads();
        uint32_t r = red[0];
        for (uint32_t i = 1; i < kLpHubThreads / 64; i++) r = is_max ? max(r, red[i]) : min(r, red[i]);
        return r;
    };
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t v = order[i];
        const uint32_t a = off[v], b = off[v + 1];
        const bool live = a != b && (!act.sparse || act.dirty[v]);  // (:74-76: no neighbours, the node keeps its label)
        if (tid == 0) used_sh = 0;
        __syncthreads();
        if (!live) continue;
        if (act.sparse && tid == 0) act.dirty[v] = 0;
        for (uint32_t base = a; base < b; base += kLpHubChunk) {
            uint32_t k[kLpHubItems];
            float x[kLpHubItems];
#pragma unroll
            for (uint32_t j = 0; j < kLpHubItems; j++) {  // blocked: thread t holds the entries 4 t .. 4 t + 3 of the chunk
                const uint32_t e = base + tid * kLpHubItems + j;
                const bool valid = e < b;
                const uint32_t t = valid ? tgt[e] : 0;
                x[j] = valid ? w[e] : 0.f;
                k[j] = valid ? labels[t] : CZ_NONE;  // (past the list: the largest key, and behind every entry that shares it)
            }
            Sort().sort(k, x, sort_storage, 0, label_bits);
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < kLpHubItems; j++) {
                sl[tid * kLpHubItems + j] = k[j];
                sw[tid * kLpHubItems + j] = x[j];
            }
            __syncthreads();
            const uint32_t n = min(b - base, kLpHubChunk);
#pragma unroll
            for (uint32_t j = 0; j < kLpHubItems; j++) {
                const uint32_t p = tid * kLpHubItems + j;
                if (p >= n) continue;
                const uint32_t L = sl[p];
                if (p > 0 && sl[p - 1] == L) continue;  // not the first of its run
                uint32_t h = (L * 0x9E3779B1u) >> (32 - bits);
                float sum = 0.0f;  // `entry(label).or_default()`
                for (;;) {
                    const uint32_t was = atomicCAS(&keys[h], CZ_NONE, L);
                    if (was == CZ_NONE) {
                        slots[atomicAdd(&used_sh, 1u)] = h;
                        break;
                    }
                    if (was == L) {
                        sum = vals[h];
                        break;
                    }
                    h = (h + 1) & tmask;
                }
                uint32_t lo = p + 1, hi = n;  // the run's end: the first entry past p with another label (the chunk is sorted)
                while (lo < hi) {
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if (sl[mid] == L) lo = mid + 1;
                    else hi = mid;
                }
#pragma unroll 8
                for (uint32_t q = p; q < lo; q++) sum += sw[q];  // `+= edge.value`, in adjacency order (:72)
                vals[h] = sum;
            }
            __syncthreads();  // (the table and sl / sw are the next chunk's)
        }
        // :77-85: the largest score under total_cmp; among the labels whose score == it, the smallest
        const uint32_t used = used_sh;
        uint32_t best_key = 0;
        for (uint32_t q = tid; q < used; q += kLpHubThreads) {
            const uint32_t bts = __float_as_uint(vals[slots[q]]);
            best_key = max(best_key, (bts & 0x80000000u) ? ~bts : (bts | 0x80000000u));
        }
        best_key = block_reduce(best_key, true);
        const float max_score = __uint_as_float((best_key & 0x80000000u) ? (best_key & 0x7FFFFFFFu) : ~best_key);
        uint32_t new_label = CZ_NONE;
        for (uint32_t q = tid; q < used; q += kLpHubThreads) {
            const uint32_t slot = slots[q];
            if (vals[slot] == max_score) new_label = min(new_label, keys[slot]);
        }
        new_label = block_reduce(new_label, false);
        for (uint32_t q = tid; q < used; q += kLpHubThreads) keys[slots[q]] = CZ_NONE;
        const bool changed = new_label != CZ_NONE && new_label != labels[v];  // (every thread reads the same word)
        __syncthreads();  // (labels[v] read by everybody before it is written; the table is clean for the next node)
        if (tid == 0) {
            if (new_label == CZ_NONE) flags[1] = 1;  // the best score is NaN: `choose` on an empty list, the reference panics
            else if (changed) {
                labels[v] = new_label;
                flags[0] = 1;
            }
            if (!act.sparse) act.chg[v] = changed;
        }
        if (act.sparse && changed)
            for (uint32_t e = act.moff[v] + tid; e < act.moff[v + 1]; e += kLpHubThreads) act.dirty[act.mtgt[e]] = 1;
    }
}

}  // namespace

extern "C" int cz_label_propagation(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N,
                                    uint64_t E, uint32_t max_iter, uint32_t *labels, uint32_t *iters_run, uint32_t *n_colours,
                                    const volatile uint8_t *poison, uint32_t flags) {
    if (iters_run) *iters_run = 0;
    if (n_colours) *n_colours = 0;
    int rc = cz::ensure_device();
    if (rc) return rc;
    t_timing.start();
    if (N == 0) return CZ_OK;
    if (!labels) return cz::set_error(CZ_E_INVALID, "null labels");
    rc = check_csr(out_offsets, out_targets, N, E);
    if (rc) return rc;
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    uint32_t max_deg = 0;
    size_t n_long_nodes = 0;  // (lists the colouring hands to lp_colour_long_kernel)
    for (uint32_t v = 0; v < N; v++) {
        const uint32_t deg = out_offsets[v + 1] - out_offsets[v];
        max_deg = std::max(max_deg, deg);
        n_long_nodes += deg > kLpColourLong;
    }
    cz::DevBuf<uint32_t> d_off, d_tgt, d_ioff, d_isrc, d_colour, d_labels, d_order, d_flags, d_cnt, d_scratch;
    cz::DevBuf<float> d_w;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_tgt.alloc(E));
    CZ_HIP(d_w.alloc(E));
    CZ_HIP(d_colour.alloc(N));
    CZ_HIP(d_labels.alloc(N));
    CZ_HIP(d_order.alloc(N));
    CZ_HIP(d_flags.alloc(4));
    CZ_HIP(d_cnt.alloc((size_t)N + 1));
    CZ_HIP(d_scratch.alloc(scan_scratch_words((size_t)N + 1)));
    CZ_HIP(hipMemcpy(d_off.p, out_offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) {
        CZ_HIP(hipMemcpy(d_tgt.p, out_targets, E * 4, hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(d_w.p, weights, E * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = nullptr;
    t_timing.lap(T_UPLOAD);
    // ---- the colouring looks at edges in EITHER direction.  On a symmetric adjacency (symmetric multiplicities: what
    // as_directed_weighted_graph(undirected = true) builds) the in-lists ARE the out-lists, and one side is enough: a node's
    // pending count is its higher out-entries, and every higher neighbour takes off one per entry of its own list, which is as
    // many.  The caller may vouch for that (CZ_ADJ_SYMMETRIC); otherwise it is verified exactly (tri_symmetry_kernel: ~E log d
    // reads) and only an adjacency that is NOT symmetric pays for the transposed one (in-degree histogram + scatter by random
    // atomics: 16 of the rule's 55 ms on the 10M / 200M graph, round 3).  The labels are the same either way.
    CZ_HIP(hipMemsetAsync(d_flags.p, 0, 16, s));
    bool symmetric = (flags & CZ_ADJ_SYMMETRIC) != 0;
    if (!symmetric && E) {
        cz::DevBuf<unsigned long long> d_updown;
        CZ_HIP(d_updown.alloc(2));
        CZ_HIP(hipMemsetAsync(d_updown.p, 0, 16, s));
        hipLaunchKernelGGL(tri_symmetry_kernel, dim3(grid_for((uint64_t)N * 16)), dim3(256), 0, s, d_off.p, d_tgt.p, N, d_flags.p + 2, d_updown.p);
        uint32_t bad = 0;
        unsigned long long updown[2] = {0, 0};
        CZ_HIP(hipMemcpy(&bad, d_flags.p + 2, 4, hipMemcpyDeviceToHost));
        CZ_HIP(hipMemcpy(updown, d_updown.p, 16, hipMemcpyDeviceToHost));
        symmetric = bad == 0 && updown[0] == updown[1];
        CZ_HIP(hipMemsetAsync(d_flags.p, 0, 16, s));
    }
    if ((flags & CZ_ADJ_SYMMETRIC) != 0 && E) {  // vouched for: the range of the targets is still checked
        hipLaunchKernelGGL(targets_in_range_kernel, dim3(grid_for(E)), dim3(kT), 0, s, d_tgt.p, E, N, d_flags.p + 2);
        uint32_t oor = 0;
        CZ_HIP(hipMemcpy(&oor, d_flags.p + 2, 4, hipMemcpyDeviceToHost));
        if (oor) return cz::set_error(CZ_E_INVALID, "a target is out of range");
        CZ_HIP(hipMemsetAsync(d_flags.p, 0, 16, s));
    }
    const uint32_t *c_ioff = nullptr, *c_isrc = nullptr;  // (null: one side)
    if (!symmetric) {
        // the transposed adjacency (sources only, in any order inside a list)
        CZ_HIP(d_ioff.alloc((size_t)N + 1));
        CZ_HIP(d_isrc.alloc(E));
        CZ_HIP(hipMemsetAsync(d_cnt.p, 0, ((size_t)N + 1) * 4, s));
        if (E) hipLaunchKernelGGL(lp_indegree_kernel, dim3(grid_for(E)), dim3(kT), 0, s, d_tgt.p, E, N, d_cnt.p, d_flags.p + 2);
        rc = exclusive_scan(d_cnt.p, d_ioff.p, N + 1, d_flags.p + 3, d_scratch.p, s);
        if (rc) return rc;
        CZ_HIP(hipMemsetAsync(d_cnt.p, 0, ((size_t)N + 1) * 4, s));
        if (E)
            hipLaunchKernelGGL(lp_transpose_kernel, dim3(grid_for((uint64_t)N * kSsspLanes)), dim3(kT), 0, s, d_off.p, d_tgt.p, N, d_ioff.p,
                               d_cnt.p, d_isrc.p);
        uint32_t bad = 0;
        CZ_HIP(hipMemcpy(&bad, d_flags.p + 2, 4, hipMemcpyDeviceToHost));
        if (bad) return cz::set_error(CZ_E_INVALID, "a target is out of range");
        c_ioff = d_ioff.p;
        c_isrc = d_isrc.p;
    }
    // ---- colouring
    // (the label and order arrays are not in use yet: they hold this round's and the next round's list of uncoloured nodes)
    uint32_t n_col = 0, coloured = 0;
    uint32_t *list = d_labels.p, *list_next = d_order.p, *pending = d_cnt.p;  // (d_cnt: the transpose is done with its cursors)
    CZ_HIP(hipMemsetAsync(d_flags.p, 0, 4, s));
    hipLaunchKernelGGL(lp_pending_kernel, dim3(grid_for((uint64_t)N * kSsspLanes)), dim3(kT), 0, s, d_off.p, d_tgt.p, c_ioff, c_isrc, N,
                       pending, d_colour.p, QueueT<uint32_t>{list, d_flags.p});
    uint32_t n_list = 0;
    CZ_HIP(hipMemcpy(&n_list, d_flags.p, 4, hipMemcpyDeviceToHost));
    // the nodes a round sets aside for lp_colour_long_kernel: at most the nodes with a long list on either side
    uint32_t max_list = max_deg;
    if (!symmetric) {  // (the in-lists live on the device only; their longest is not known here)
        max_list = 0xFFFFFFFFu;
        n_long_nodes = N;
    }
    cz::DevBuf<uint32_t> d_long;
    CZ_HIP(d_long.alloc(std::max<size_t>(n_long_nodes, 1)));
    while (n_list > 0) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        coloured += n_list;
        n_col++;
        CZ_HIP(hipMemsetAsync(d_flags.p, 0, 8, s));
        hipLaunchKernelGGL(lp_colour_kernel, dim3(grid_for((uint64_t)n_list * kSsspLanes)), dim3(kT), 0, s, d_off.p, d_tgt.p, c_ioff, c_isrc,
                           list, n_list, n_col, pending, d_colour.p, QueueT<uint32_t>{list_next, d_flags.p}, QueueT<uint32_t>{d_long.p, d_flags.p + 1});
        if (max_list > kLpColourLong)  // (reads the count on the device: nothing to do in most rounds)
            hipLaunchKernelGGL(lp_colour_long_kernel, dim3(256), dim3(kT), 0, s, d_off.p, d_tgt.p, c_ioff, c_isrc,
                               QueueT<uint32_t>{d_long.p, d_flags.p + 1}, n_col, pending, d_colour.p, QueueT<uint32_t>{list_next, d_flags.p});
        CZ_HIP(hipMemcpy(&n_list, d_flags.p, 4, hipMemcpyDeviceToHost));
        std::swap(list, list_next);
    }
    if (coloured != N) return cz::set_error(CZ_E_HIP, "internal: the colouring reached %u of %u nodes", coloured, N);
    // ---- the classes as lists, the nodes of larger degree at the end of each.  (The order inside a class does not matter: its
    // nodes share no edge.)  Two buckets per class, filled on the device: per-workgroup counts in LDS, one reservation per
    // (workgroup, bucket); the host only sees the 2 x n_col totals.  More classes than the LDS counters hold: on the host.
    std::vector<uint32_t> small_end((size_t)n_col + 1, 0), class_off((size_t)n_col + 1, 0), tiny_end((size_t)n_col + 1, 0);
    if (3 * (size_t)n_col <= kLpBuckets) {
        const uint32_t nb = 3 * n_col;
        cz::DevBuf<uint32_t> d_bcnt;
        CZ_HIP(d_bcnt.alloc(nb));
        CZ_HIP(hipMemsetAsync(d_bcnt.p, 0, (size_t)nb * 4, s));
        hipLaunchKernelGGL(lp_bucket_kernel<false>, dim3(grid_for(N)), dim3(kT), 0, s, d_colour.p, d_off.p, N, nb, d_bcnt.p, (uint32_t *)nullptr);
        std::vector<uint32_t> cnt(nb), cur(nb);
        CZ_HIP(hipMemcpy(cnt.data(), d_bcnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
        uint32_t at = 0;
        for (uint32_t c = 0; c < n_col; c++) {
            class_off[c] = at;
            cur[3 * c] = at;
            at += cnt[3 * c];
            tiny_end[c] = at;
            cur[3 * c + 1] = at;
            at += cnt[3 * c + 1];
            small_end[c] = at;
            cur[3 * c + 2] = at;
            at += cnt[3 * c + 2];
        }
        class_off[n_col] = small_end[n_col] = tiny_end[n_col] = at;
        if (at != N) return cz::set_error(CZ_E_HIP, "internal: the classes hold %u of %u nodes", at, N);
        CZ_HIP(hipMemcpy(d_bcnt.p, cur.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(lp_bucket_kernel<true>, dim3(grid_for(N)), dim3(kT), 0, s, d_colour.p, d_off.p, N, nb, d_bcnt.p, d_order.p);
    } else {
        std::vector<uint32_t> colour(N), order(N);
        CZ_HIP(hipMemcpy(colour.data(), d_colour.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (uint32_t v = 0; v < N; v++) class_off[colour[v] + 1]++;
        for (uint32_t c = 0; c < n_col; c++) class_off[c + 1] += class_off[c];
        std::vector<uint32_t> cur_small(class_off.begin(), class_off.end() - 1), n_small(n_col, 0);
        for (uint32_t v = 0; v < N; v++)
            if (out_offsets[v + 1] - out_offsets[v] <= kLpSmall) n_small[colour[v]]++;
        std::vector<uint32_t> cur_hub(n_col);
        for (uint32_t c = 0; c < n_col; c++) {
            small_end[c] = class_off[c] + n_small[c];
            tiny_end[c] = class_off[c];  // (this many classes: no separate list of the short nodes)
            cur_hub[c] = small_end[c];
        }
        tiny_end[n_col] = class_off[n_col];
        for (uint32_t v = 0; v < N; v++) {
            const uint32_t c = colour[v];
            if (out_offsets[v + 1] - out_offsets[v] <= kLpSmall) order[cur_small[c]++] = v;
            else order[cur_hub[c]++] = v;
        }
        CZ_HIP(hipMemcpy(d_order.p, order.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    }
    // tables of the hub waves
    uint32_t hub_bits = 10;
    while ((1ull << hub_bits) * 3 / 4 < max_deg) hub_bits++;
    const bool any_hub = max_deg > kLpSmall;
    // (a table per workgroup; a workgroup per CU, fewer when the tables would pass 2 GiB together)
    const uint32_t hub_blocks = any_hub ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(256, (2048ull << 20) / (12ull << hub_bits))) : 0;
    uint32_t label_bits = 1;
    while (label_bits < 32 && (1ull << label_bits) < N) label_bits++;
    cz::DevBuf<uint32_t> d_tkeys, d_tslots;
    cz::DevBuf<float> d_tvals;
    if (any_hub) {
        const size_t words = (size_t)hub_blocks << hub_bits;
        CZ_HIP(d_tkeys.alloc(words));
        CZ_HIP(d_tvals.alloc(words));
        CZ_HIP(d_tslots.alloc(words));
        CZ_HIP(hipMemsetAsync(d_tkeys.p, 0xFF, words * 4, s));
    }
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(N)), dim3(kT), 0, s, d_labels.p, N);  // :61
    // the active set (LpActive): an iteration that changed at most this share of the nodes switches the rest of the run to
    // evaluating only the dependants of changed nodes.  CZ_LP_SPARSE_FRAC: 0 = never, 1 = from the second iteration on.
    double sparse_frac = 0.125;
    if (const char *sf = getenv("CZ_LP_SPARSE_FRAC")) sparse_frac = atof(sf);
    cz::DevBuf<uint8_t> d_dirty, d_chg;
    CZ_HIP(d_dirty.alloc(N));
    CZ_HIP(d_chg.alloc(N));
    CZ_HIP(hipMemsetAsync(d_chg.p, 0, N, s));
    // nodes a 16-lane group of the tiny kernel has in flight (CZ_LP_TINY_U = 1 | 2 | 4).  Measured on the 10M / 200M graph
    // (scratch/r5_lp.sh): 35.3 / 36.1 / 37.0 ms -- the kernel is at the random-REQUEST rate, not waiting on a chain, so 1.
    uint32_t tiny_u = 1;
    if (const char *tu = getenv("CZ_LP_TINY_U")) tiny_u = atoi(tu) == 2 ? 2 : atoi(tu) == 4 ? 4 : 1;
    LpActive act{d_dirty.p, d_chg.p, symmetric ? d_off.p : c_ioff, symmetric ? d_tgt.p : c_isrc, 0};
    uint32_t iters = 0;
    for (uint32_t it = 0; it < max_iter; it++) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        CZ_HIP(hipMemsetAsync(d_flags.p, 0, 12, s));
        iters++;
        for (uint32_t c = 0; c < n_col; c++) {
            const uint32_t nt = tiny_end[c] - class_off[c], ns = small_end[c] - tiny_end[c], nh = class_off[c + 1] - small_end[c];
            if (nt) {
                const dim3 gt(grid_for(((uint64_t)nt + tiny_u - 1) / tiny_u * 16));
                auto tiny = tiny_u == 1 ? lp_update_tiny_kernel<1> : tiny_u == 2 ? lp_update_tiny_kernel<2> : lp_update_tiny_kernel<4>;
                hipLaunchKernelGGL(tiny, gt, dim3(kT), 0, s, d_off.p, d_tgt.p, d_w.p, d_order.p + class_off[c], nt, d_labels.p, d_flags.p, act);
            }
            if (ns)
                hipLaunchKernelGGL(lp_update_kernel, dim3(grid_for((uint64_t)ns * 64)), dim3(kT), 0, s, d_off.p, d_tgt.p, d_w.p,
                                   d_order.p + tiny_end[c], ns, d_labels.p, d_flags.p, act);
            if (nh)
                hipLaunchKernelGGL(lp_update_hub_kernel, dim3(std::min<uint32_t>(hub_blocks, nh)), dim3(kLpHubThreads), 0, s, d_off.p, d_tgt.p,
                                   d_w.p, d_order.p + small_end[c], nh, d_labels.p, d_flags.p, d_tkeys.p, d_tvals.p, d_tslots.p, hub_bits,
                                   label_bits, act);
        }
        const bool count_changes = !act.sparse && sparse_frac > 0 && it + 1 < max_iter;
        if (count_changes) hipLaunchKernelGGL(lp_count_changed_kernel, dim3(grid_for(N)), dim3(kT), 0, s, d_chg.p, N, d_flags.p + 2);
        uint32_t h[3];
        CZ_HIP(hipMemcpy(h, d_flags.p, 12, hipMemcpyDeviceToHost));
        if (h[1]) return cz::set_error(CZ_E_INVALID, "LabelPropagation: a node's best label score is NaN (the reference panics there)");
        if (!h[0]) break;  // :92-94
        if (count_changes && (double)h[2] <= sparse_frac * (double)N) {
            CZ_HIP(hipMemsetAsync(d_dirty.p, 0, N, s));
            hipLaunchKernelGGL(lp_mark_kernel, dim3(grid_for((uint64_t)N * 16)), dim3(kT), 0, s, d_chg.p, N, act.moff, act.mtgt, d_dirty.p);
            act.sparse = 1;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "label propagation launch: %s", hipGetErrorString(e));
    t_timing.lap(T_DEVICE);
    CZ_HIP(hipMemcpy(labels, d_labels.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    t_timing.lap(T_DOWNLOAD);
    if (iters_run) *iters_run = iters;
    if (n_colours) *n_colours = n_col;
    return CZ_OK;
}

// =====================================================================================================================
// ONE traversal over a vertex-partitioned graph (third row): the loops of sharded_traversal.hpp
// over these kernels + the communicator's all-reduces.  Rank r holds the out-adjacency of [row_begin, row_end) (offsets
// relative to row_begin); every per-node array is full length on every rank.
// =====================================================================================================================
#include "sharded_traversal.hpp"

namespace {

__global__ void __launch_bounds__(kT)
bfs_commit_kernel(const uint32_t *__restrict__ buf, uint32_t total, uint32_t *__restrict__ order_at, uint32_t *__restrict__ depth,
                  uint32_t *__restrict__ vis, uint32_t next_depth) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const uint32_t v = buf[j] - 1u;  // written as id + 1 by exactly one rank, 0 by the others: the sum is id + 1
        order_at[j] = v;
        depth[v] = next_depth;
        atomicOr(&vis[v >> 5], 1u << (v & 31));  // (the rank that emitted v has set the bit already)
    }
}

struct HipShardedBfs {
    cz_comm *comm;
    hipStream_t s;
    uint32_t N, rb, re;
    const uint32_t *goals_host;
    uint32_t n_goals;
    cz::DevBuf<uint32_t> off, tgt, depth, parent, claim, order, cnt, pos, scratch, buf, misc, goals, vis;
    cz::DevBuf<uint8_t> won;
    size_t vis_words = 0;

    int alloc(const uint32_t *h_off, const uint32_t *h_tgt, uint64_t e_local) {
        const uint32_t rows = re - rb;
        vis_words = ((size_t)N + 31) / 32;
        CZ_HIP(vis.alloc(vis_words));  // the visited bits and the won bytes of the one-GPU rule (section BFS above)
        CZ_HIP(won.alloc(e_local));
        CZ_HIP(hipMemsetAsync(vis.p, 0, vis_words * 4, s));
        CZ_HIP(off.alloc((size_t)rows + 1));
        CZ_HIP(tgt.alloc(e_local));
        CZ_HIP(depth.alloc(N));
        CZ_HIP(parent.alloc(N));
        CZ_HIP(claim.alloc(N));
        CZ_HIP(order.alloc((size_t)N + 1));
        CZ_HIP(cnt.alloc(N));
        CZ_HIP(pos.alloc(N));
        CZ_HIP(buf.alloc(N));
        CZ_HIP(scratch.alloc(scan_scratch_words(N)));
        CZ_HIP(misc.alloc(4));
        CZ_HIP(hipMemcpy(off.p, h_off, ((size_t)rows + 1) * 4, hipMemcpyHostToDevice));
        if (e_local) CZ_HIP(hipMemcpy(tgt.p, h_tgt, e_local * 4, hipMemcpyHostToDevice));
        if (goals_host && n_goals) {
            CZ_HIP(goals.alloc(n_goals));
            CZ_HIP(hipMemcpy(goals.p, goals_host, (size_t)n_goals * 4, hipMemcpyHostToDevice));
        }
        CZ_HIP(hipMemsetAsync(depth.p, 0xFF, (size_t)N * 4, s));
        CZ_HIP(hipMemsetAsync(claim.p, 0xFF, (size_t)N * 4, s));
        return CZ_OK;
    }
    int any_poisoned(bool mine, bool *any) {
        const uint32_t v = mine ? 1u : 0u;
        CZ_HIP(hipMemcpyAsync(misc.p + 2, &v, 4, hipMemcpyHostToDevice, s));
        int rc = cz::comm_all_reduce(comm, misc.p + 2, 1, cz::COMM_U32, cz::COMM_SUM, s);
        if (rc) return rc;
        uint32_t h = 0;
        CZ_HIP(hipMemcpyAsync(&h, misc.p + 2, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        *any = h != 0;
        return CZ_OK;
    }
    int bfs_reset(bool keep_visited) {
        CZ_HIP(hipMemsetAsync(parent.p, 0xFF, (size_t)N * 4, s));
        if (!keep_visited) {
            CZ_HIP(hipMemsetAsync(depth.p, 0xFF, (size_t)N * 4, s));
            CZ_HIP(hipMemsetAsync(claim.p, 0xFF, (size_t)N * 4, s));
            CZ_HIP(hipMemsetAsync(vis.p, 0, vis_words * 4, s));
        }
        return CZ_OK;
    }
    int bfs_seed(uint32_t start, bool *already) {
        uint32_t d = 0;
        CZ_HIP(hipMemcpyAsync(&d, depth.p + start, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        *already = d != CZ_NONE;
        if (*already) return CZ_OK;
        hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, depth.p, start, 0u);
        hipLaunchKernelGGL(or_u32_kernel, dim3(1), dim3(1), 0, s, vis.p, start >> 5, 1u << (start & 31));
        hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, order.p, 0u, start);
        return CZ_OK;
    }
    int bfs_claim(uint32_t lo, uint32_t fsize) {
        hipLaunchKernelGGL(bfs_claim_kernel, dim3(grid_for((uint64_t)fsize * kBfsLanes)), dim3(kT), 0, s, off.p, tgt.p, order.p + lo,
                           fsize, depth.p, (const uint32_t *)vis.p, claim.p, rb, re);
        return CZ_OK;
    }
    int reduce_claim() { return cz::comm_all_reduce(comm, claim.p, N, cz::COMM_U32, cz::COMM_MIN, s); }
    int bfs_count(uint32_t lo, uint32_t fsize) {
        CZ_HIP(hipMemsetAsync(cnt.p, 0, (size_t)fsize * 4, s));
        hipLaunchKernelGGL(bfs_count_kernel, dim3(grid_for((uint64_t)fsize * kBfsLanes)), dim3(kT), 0, s, off.p, tgt.p, order.p + lo,
                           fsize, depth.p, (const uint32_t *)vis.p, claim.p, cnt.p, won.p, rb, re);
        return CZ_OK;
    }
    int reduce_counts(uint32_t fsize) { return cz::comm_all_reduce(comm, cnt.p, fsize, cz::COMM_U32, cz::COMM_SUM, s); }
    int bfs_scan(uint32_t fsize, uint32_t *total) {
        int rc = exclusive_scan(cnt.p, pos.p, fsize, misc.p, scratch.p, s);
        if (rc) return rc;
        CZ_HIP(hipMemcpyAsync(total, misc.p, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        return CZ_OK;
    }
    int bfs_emit(uint32_t lo, uint32_t fsize, uint32_t total, uint32_t next_depth) {
        if (total) CZ_HIP(hipMemsetAsync(buf.p, 0, (size_t)total * 4, s));
        hipLaunchKernelGGL(bfs_emit_kernel, dim3(grid_for((uint64_t)fsize * kBfsLanes)), dim3(kT), 0, s, off.p, tgt.p, order.p + lo,
                           fsize, depth.p, vis.p, claim.p, (const uint8_t *)won.p, pos.p, buf.p, parent.p, next_depth, rb, re, 1u);
        return CZ_OK;
    }
    int reduce_next(uint32_t total) { return cz::comm_all_reduce(comm, buf.p, total, cz::COMM_U32, cz::COMM_SUM, s); }
    int bfs_commit(uint32_t at, uint32_t total, uint32_t next_depth) {
        if (total)
            hipLaunchKernelGGL(bfs_commit_kernel, dim3(grid_for(total)), dim3(kT), 0, s, buf.p, total, order.p + at, depth.p, vis.p, next_depth);
        return CZ_OK;
    }
    int goals_left(uint32_t start, uint32_t *left) {
        CZ_HIP(hipMemsetAsync(misc.p + 1, 0, 4, s));
        hipLaunchKernelGGL(bfs_goals_left_kernel, dim3(grid_for(n_goals)), dim3(kT), 0, s, goals.p, n_goals, N, depth.p, start, misc.p + 1);
        CZ_HIP(hipMemcpyAsync(left, misc.p + 1, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        return CZ_OK;
    }
    int reduce_parents() { return cz::comm_all_reduce(comm, parent.p, N, cz::COMM_U32, cz::COMM_MIN, s); }
};

// ---- SSSP: near-far piles, sparse exchange (sharded_traversal.hpp says what each step means) ----
constexpr unsigned long long kIdleProp = ~0ull;  // a proposal slot nobody touched this round
constexpr unsigned long long kPadNode = 0xFFFFFFFFull;

// relax: a 16-lane group per near entry this rank owns; per target the best strictly improving word lands in prop[] (atomicMin),
// and whoever finds the slot idle lists the target
__global__ void __launch_bounds__(kT)
sssp_sp_relax_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t rb,
                     uint32_t re, const uint32_t *__restrict__ near, uint32_t n_near, const unsigned long long *__restrict__ dp,
                     unsigned long long *__restrict__ prop, QueueT<uint32_t> touched) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kSsspLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kSsspLanes, ngroups = gridDim.x * blockDim.x / kSsspLanes;
    const uint32_t rounds = (n_near + ngroups - 1) / ngroups;  // every group of the GRID runs the same trip count (ballots, barriers)
    __shared__ StagedPileT<uint32_t> st;
    if (threadIdx.x == 0) st.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        uint32_t u = i < n_near ? near[i] : CZ_NONE;
        const bool live = u != CZ_NONE && u >= rb && u < re;
        const uint32_t e0 = live ? off[u - rb] : 0, e1 = live ? off[u - rb + 1] : 0;
        const float du = live ? __uint_as_float((uint32_t)(dp[u] >> 32)) : 0.f;
        uint32_t maxlen = e1 - e0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o, 64));
        for (uint32_t b = 0; b < maxlen; b += kSsspLanes) {
            const uint32_t e = e0 + b + glane;
            bool first = false;
            uint32_t v = 0;
            if (e < e1) {
                v = tgt[e];
                const uint32_t nb = __float_as_uint(du + w[e]);  // `cost + path_weight` in f32 (shortest_path_dijkstra.rs:303)
                if (nb < (uint32_t)(dp[v] >> 32))                // strict `<` against the round's starting cost (:304)
                    first = atomicMin(&prop[v], ((unsigned long long)nb << 32) | u) == kIdleProp;
            }
            staged_push(touched, st, first, v, lane);
        }
        if ((r & 7) == 7 || r + 1 == rounds) staged_flush(touched, st);
    }
}

// this rank's slot of the exchange buffer: its pairs (word, target), the proposal slots idle again, padding behind them
__global__ void __launch_bounds__(kT)
sssp_sp_collect_kernel(const uint32_t *__restrict__ touched, uint32_t n, uint32_t longest, unsigned long long *__restrict__ prop,
                       unsigned long long *__restrict__ slot) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < longest; i += gridDim.x * blockDim.x) {
        unsigned long long word = kIdleProp, node = kPadNode;
        if (i < n) {
            node = touched[i];
            word = prop[node];
            prop[node] = kIdleProp;
        }
        slot[2 * (size_t)i] = word;
        slot[2 * (size_t)i + 1] = node;
    }
}

// counts[r] = (pairs of rank r) | (its cancellation flag << 40): every rank writes its own word, the sum gathers them
__global__ void sssp_sp_count_kernel(unsigned long long *__restrict__ counts, uint32_t world, uint32_t rank, const uint32_t *__restrict__ n_touched,
                                     uint32_t poisoned) {
    const uint32_t i = threadIdx.x;
    if (i < world) counts[i] = i == rank ? ((unsigned long long)*n_touched | ((unsigned long long)poisoned << 40)) : 0ull;
}

__global__ void __launch_bounds__(kT)
sssp_sp_apply_min_kernel(const unsigned long long *__restrict__ pairs, uint64_t total, unsigned long long *__restrict__ dp,
                         uint8_t *__restrict__ lowered) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long word = pairs[2 * i], node = pairs[2 * i + 1];
        lowered[i] = node != kPadNode && word < atomicMin(&dp[node], word);
    }
}

// the winners: a pair that lowered its target's word and still IS that word (words of different ranks differ in the parent)
__global__ void __launch_bounds__(kT)
sssp_sp_apply_place_kernel(const unsigned long long *__restrict__ pairs, const uint8_t *__restrict__ lowered, uint64_t total,
                           const unsigned long long *__restrict__ dp, uint32_t thr_bits, QueueT<uint32_t> near, SsspQueue far) {
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (total + stride - 1) / stride;
    __shared__ StagedPileT<uint32_t> st_near;
    __shared__ StagedPile st_far;
    if (threadIdx.x == 0) st_near.count = st_far.count = 0;
    __syncthreads();
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + r * stride;
        bool to_near = false, to_far = false;
        uint32_t v = 0, cost = 0;
        if (i < total && lowered[i]) {
            const unsigned long long word = pairs[2 * i];
            v = (uint32_t)pairs[2 * i + 1];
            if (dp[v] == word) {
                cost = (uint32_t)(word >> 32);
                to_near = cost < thr_bits;
                to_far = !to_near;
            }
        }
        staged_push(near, st_near, to_near, v, lane);
        staged_push(far, st_far, to_far, ((unsigned long long)cost << 32) | v, lane);
        if ((r & 1) == 1 || r + 1 == rounds) {  // at most one entry per thread and iteration: two iterations fit the buffer
            staged_flush(near, st_near);
            staged_flush(far, st_far);
        }
    }
}

// the cheapest LIVE far entry (its node still has the cost it was filed under)
__global__ void __launch_bounds__(kT)
sssp_sp_far_min_kernel(const unsigned long long *__restrict__ far, uint32_t n_far, const unsigned long long *__restrict__ dp,
                       uint32_t *__restrict__ min_bits) {
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_far; i += gridDim.x * blockDim.x) {
        const unsigned long long ent = far[i];
        const uint32_t cost = (uint32_t)(ent >> 32);
        if ((uint32_t)(dp[(uint32_t)ent] >> 32) == cost) m = min(m, cost);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m != 0xFFFFFFFFu) atomicMin(min_bits, m);
}

// live far entries below the threshold -> the near list, the other live ones -> the next far pile; stale ones are dropped
__global__ void __launch_bounds__(kT)
sssp_sp_split_kernel(const unsigned long long *__restrict__ far, uint32_t n_far, const unsigned long long *__restrict__ dp,
                     uint32_t thr_bits, QueueT<uint32_t> near, SsspQueue far_next) {
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n_far + stride - 1) / stride;
    __shared__ StagedPileT<uint32_t> st_near;
    __shared__ StagedPile st_far;
    if (threadIdx.x == 0) st_near.count = st_far.count = 0;
    __syncthreads();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + r * stride;
        bool to_near = false, to_far = false;
        unsigned long long ent = 0;
        if (i < n_far) {
            ent = far[i];
            const uint32_t cost = (uint32_t)(ent >> 32);
            if ((uint32_t)(dp[(uint32_t)ent] >> 32) == cost) {
                to_near = cost < thr_bits;
                to_far = !to_near;
            }
        }
        staged_push(near, st_near, to_near, (uint32_t)ent, lane);
        staged_push(far_next, st_far, to_far, ent, lane);
        if ((r & 1) == 1 || r + 1 == rounds) {
            staged_flush(near, st_near);
            staged_flush(far_next, st_far);
        }
    }
}

struct HipShardedSssp {
    cz_comm *comm;
    hipStream_t s;
    uint32_t N, rb, re;
    cz::DevBuf<uint32_t> off, tgt, near, touched, canon, misc, parent;
    cz::DevBuf<float> w, dist;
    cz::DevBuf<unsigned long long> dpb, prop, far_a, far_b, pairs, counts;
    cz::DevBuf<uint8_t> lowered;
    unsigned long long *dp = nullptr, *far = nullptr, *far_next = nullptr;
    std::vector<unsigned long long> h_counts;
    size_t far_cap = 0, compact_at = 0;
    uint32_t n_far = 0, my_pairs = 0, thr_bits = 0;
    uint64_t rounds = 0, pairs_exchanged = 0, compactions = 0, buckets = 0;  // (over all the starts of a call)
    float delta = 0.f;
    bool one_pile = false;

    int alloc(const uint32_t *h_off, const uint32_t *h_tgt, const float *h_w, uint64_t e_local) {
        const uint32_t rows = re - rb, world = (uint32_t)cz::comm_world(comm);
        far_cap = 3 * (size_t)N + 1024;  // (a compaction leaves at most N live entries, a round adds at most N)
        compact_at = far_cap;
        if (const char *fc = getenv("CZ_SSSP_FAR_COMPACT")) compact_at = (size_t)atoll(fc);  // tests: drop the stale entries this often
        CZ_HIP(off.alloc((size_t)rows + 1));
        CZ_HIP(tgt.alloc(e_local));
        CZ_HIP(w.alloc(e_local));
        CZ_HIP(near.alloc(N));
        CZ_HIP(touched.alloc(N));
        CZ_HIP(canon.alloc(N));
        CZ_HIP(misc.alloc(8));
        CZ_HIP(parent.alloc(N));
        CZ_HIP(dist.alloc(N));
        CZ_HIP(dpb.alloc(N));
        CZ_HIP(prop.alloc(N));
        CZ_HIP(far_a.alloc(far_cap));
        CZ_HIP(far_b.alloc(far_cap));
        CZ_HIP(counts.alloc(world));
        h_counts.resize(world);
        CZ_HIP(hipMemcpy(off.p, h_off, ((size_t)rows + 1) * 4, hipMemcpyHostToDevice));
        if (e_local) {
            CZ_HIP(hipMemcpy(tgt.p, h_tgt, e_local * 4, hipMemcpyHostToDevice));
            CZ_HIP(hipMemcpy(w.p, h_w, e_local * 4, hipMemcpyHostToDevice));
        }
        dp = dpb.p;
        hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(N)), dim3(kT), 0, s, prop.p, (uint64_t)N, kIdleProp);
        return CZ_OK;
    }
    // the bucket width: the mean edge weight of the WHOLE graph (the one-GPU rule's choice; CZ_SSSP_DELTA overrides; <= 0 or "inf"
    // = one pile).  A collective: every rank calls it, with its own sums.
    int agree_on_delta(const float *h_w, uint64_t e_local) {
        double h[2] = {0.0, (double)e_local};
        for (uint64_t e = 0; e < e_local; e++) h[0] += (double)h_w[e];
        cz::DevBuf<double> d;
        CZ_HIP(d.alloc(2));
        CZ_HIP(hipMemcpyAsync(d.p, h, 16, hipMemcpyHostToDevice, s));
        int rc = cz::comm_all_reduce(comm, d.p, 2, cz::COMM_F64, cz::COMM_SUM, s);
        if (rc) return rc;
        CZ_HIP(hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        delta = h[1] > 0.0 ? (float)(h[0] / h[1]) : 0.f;
        if (const char *de = getenv("CZ_SSSP_DELTA")) delta = (float)atof(de);
        one_pile = !(delta > 0.f) || !std::isfinite(delta);
        return CZ_OK;
    }
    // threshold above a cheapest cost: that cost + the bucket width, and always above the cost itself
    uint32_t threshold_over(uint32_t min_bits) const {
        if (one_pile) return 0x7F800000u;
        const float m = __builtin_bit_cast(float, min_bits);
        float t = m + delta;
        if (!(t > m)) t = std::nextafter(m, std::numeric_limits<float>::infinity());
        return __builtin_bit_cast(uint32_t, t);
    }
    int read_misc(uint32_t *h, uint32_t n) {
        CZ_HIP(hipMemcpyAsync(h, misc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        return CZ_OK;
    }
    // misc: [0] near count, [1] far count, [2] touched count, [3] min far cost bits
    int sssp_seed(uint32_t start, uint32_t *n_near) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(N)), dim3(kT), 0, s, dp, (uint64_t)N, kInfPacked);
        far = far_a.p;
        far_next = far_b.p;
        n_far = 0;
        thr_bits = threshold_over(0u);
        *n_near = 0;
        if (start < N) {
            const unsigned long long zero = 0x00000000FFFFFFFFull;  // cost 0.0, no parent
            CZ_HIP(hipMemcpyAsync(dp + start, &zero, 8, hipMemcpyHostToDevice, s));
            CZ_HIP(hipMemcpyAsync(near.p, &start, 4, hipMemcpyHostToDevice, s));
            CZ_HIP(hipStreamSynchronize(s));
            *n_near = 1;
        }
        return CZ_OK;
    }
    int sssp_relax(uint32_t n_near) {
        CZ_HIP(hipMemsetAsync(misc.p + 2, 0, 4, s));
        hipLaunchKernelGGL(sssp_sp_relax_kernel, dim3(grid_for((uint64_t)n_near * kSsspLanes)), dim3(kT), 0, s, off.p, tgt.p, w.p, rb, re,
                           near.p, n_near, dp, prop.p, QueueT<uint32_t>{touched.p, misc.p + 2});
        return CZ_OK;
    }
    int exchange_counts(bool poisoned, uint32_t *longest, bool *any_poisoned) {
        const uint32_t world = (uint32_t)cz::comm_world(comm), rank = (uint32_t)cz::comm_rank(comm);
        if (world > 1024) return cz::set_error(CZ_E_UNSUPPORTED, "at most 1024 ranks");
        hipLaunchKernelGGL(sssp_sp_count_kernel, dim3(1), dim3(1024), 0, s, counts.p, world, rank, misc.p + 2, poisoned ? 1u : 0u);
        int rc = cz::comm_all_reduce(comm, counts.p, world, cz::COMM_U64, cz::COMM_SUM, s);
        if (rc) return rc;
        CZ_HIP(hipMemcpyAsync(h_counts.data(), counts.p, (size_t)world * 8, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        *longest = 0;
        *any_poisoned = false;
        for (uint32_t r = 0; r < world; r++) {
            *longest = std::max(*longest, (uint32_t)h_counts[r]);
            *any_poisoned = *any_poisoned || (h_counts[r] >> 40) != 0;
            pairs_exchanged += (uint32_t)h_counts[r];
        }
        my_pairs = (uint32_t)h_counts[rank];
        rounds++;
        return CZ_OK;
    }
    int exchange_pairs(uint32_t longest) {
        const uint32_t world = (uint32_t)cz::comm_world(comm), rank = (uint32_t)cz::comm_rank(comm);
        const size_t need = 2 * (size_t)world * longest;
        if (pairs.n < need) CZ_HIP(pairs.alloc(need + need / 4));  // (grows with the busiest round so far)
        if (lowered.n < (size_t)world * longest) CZ_HIP(lowered.alloc((size_t)world * longest + (size_t)world * longest / 4));
        unsigned long long *slot = pairs.p + 2 * (size_t)rank * longest;
        hipLaunchKernelGGL(sssp_sp_collect_kernel, dim3(grid_for(longest)), dim3(kT), 0, s, touched.p, my_pairs, longest, prop.p, slot);
        return cz::comm_all_gather(comm, slot, pairs.p, 2 * (size_t)longest, cz::COMM_U64, s);
    }
    int compact_far() {  // drop the stale entries: at most one live entry per node is left
        CZ_HIP(hipMemsetAsync(misc.p, 0, 8, s));
        hipLaunchKernelGGL(sssp_sp_split_kernel, dim3(grid_for(n_far)), dim3(kT), 0, s, far, n_far, dp, 0u, QueueT<uint32_t>{near.p, misc.p},
                           SsspQueue{far_next, (uint32_t *)(misc.p + 1)});
        uint32_t h[2];
        int rc = read_misc(h, 2);
        if (rc) return rc;
        std::swap(far, far_next);
        n_far = h[1];
        compactions++;
        return CZ_OK;
    }
    int sssp_apply(uint32_t longest, uint32_t *n_near, uint32_t *n_far_out) {
        const uint64_t total = (uint64_t)cz::comm_world(comm) * longest;
        if (n_far + std::min<uint64_t>(N, total) > far_cap || n_far > compact_at) {
            int rc = compact_far();
            if (rc) return rc;
        }
        const uint32_t init[2] = {0, n_far};  // near starts empty, the far pile is appended to
        uint32_t h[2];
        CZ_HIP(hipMemcpyAsync(misc.p, init, 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(sssp_sp_apply_min_kernel, dim3(grid_for(total)), dim3(kT), 0, s, pairs.p, total, dp, lowered.p);
        hipLaunchKernelGGL(sssp_sp_apply_place_kernel, dim3(grid_for(total)), dim3(kT), 0, s, pairs.p, lowered.p, total, dp, thr_bits,
                           QueueT<uint32_t>{near.p, misc.p}, SsspQueue{far, (uint32_t *)(misc.p + 1)});
        int rc = read_misc(h, 2);
        if (rc) return rc;
        *n_near = h[0];
        *n_far_out = n_far = h[1];
        return CZ_OK;
    }
    int sssp_next_bucket(uint32_t *n_near, uint32_t *n_far_out) {
        const uint32_t none = 0xFFFFFFFFu;
        CZ_HIP(hipMemcpyAsync(misc.p + 3, &none, 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(sssp_sp_far_min_kernel, dim3(grid_for(n_far)), dim3(kT), 0, s, far, n_far, dp, misc.p + 3);
        uint32_t h[4];
        int rc = read_misc(h, 4);
        if (rc) return rc;
        *n_near = 0;
        if (h[3] == none) {  // every entry was stale
            *n_far_out = n_far = 0;
            return CZ_OK;
        }
        thr_bits = threshold_over(h[3]);
        buckets++;
        CZ_HIP(hipMemsetAsync(misc.p, 0, 8, s));
        hipLaunchKernelGGL(sssp_sp_split_kernel, dim3(grid_for(n_far)), dim3(kT), 0, s, far, n_far, dp, thr_bits,
                           QueueT<uint32_t>{near.p, misc.p}, SsspQueue{far_next, (uint32_t *)(misc.p + 1)});
        if ((rc = read_misc(h, 2))) return rc;
        std::swap(far, far_next);
        *n_near = h[0];
        *n_far_out = n_far = h[1];
        return CZ_OK;
    }
    int sssp_canonical_parents() {
        CZ_HIP(hipMemsetAsync(canon.p, 0xFF, (size_t)N * 4, s));
        hipLaunchKernelGGL(sssp_canon_kernel, dim3(grid_for((uint64_t)N * kSsspLanes)), dim3(kT), 0, s, off.p, tgt.p, w.p, N, 1u, dp,
                           canon.p, rb, re);
        return CZ_OK;
    }
    int reduce_canonical() { return cz::comm_all_reduce(comm, canon.p, N, cz::COMM_U32, cz::COMM_MIN, s); }
};

thread_local uint64_t t_sssp_sharded_stats[4] = {0, 0, 0, 0};

int check_shard(const uint32_t *off, const uint32_t *tgt, uint32_t N, uint32_t rb, uint32_t re, uint64_t e_local) {
    if (rb > re || re > N) return cz::set_error(CZ_E_INVALID, "bad row range [%u,%u) of %u", rb, re, N);
    if (!off) return cz::set_error(CZ_E_INVALID, "null offsets");
    if (off[0] != 0 || off[re - rb] != e_local) return cz::set_error(CZ_E_INVALID, "offsets must be relative to the shard and end at E_local");
    if (e_local > 0 && !tgt) return cz::set_error(CZ_E_INVALID, "null targets");
    if (e_local >= 0xFFFFFFFFull) return cz::set_error(CZ_E_UNSUPPORTED, "a shard holds fewer than 2^32-1 edges");
    // the kernels index full-length per-node arrays by the targets (prop[v], claim[v], comp[v]) and walk [off[r], off[r+1]):
    // both are checked here, on the host arrays the caller handed over
    for (uint32_t r = 0; r < re - rb; r++)
        if (off[r + 1] < off[r]) return cz::set_error(CZ_E_INVALID, "offsets not monotone at local row %u", r);
    for (uint64_t e = 0; e < e_local; e++)
        if (tgt[e] >= N) return cz::set_error(CZ_E_INVALID, "target %u of local edge %llu is out of range (N = %u)", tgt[e], (unsigned long long)e, N);
    return CZ_OK;
}

// A rank that rejected its shard (or could not allocate) must not leave the others blocked in the loop's first collective:
// every rank reports its status, the sum is all-reduced, and all of them leave together.
int collective_status(cz_comm *comm, int mine) {
    if (cz::comm_world(comm) <= 1) return mine;
    const std::string my_msg = mine ? std::string(cz_last_error()) : std::string();
    cz::DevBuf<uint32_t> d;
    uint32_t h = mine ? 1u : 0u;
    if (d.alloc(1) != hipSuccess) return mine ? mine : cz::set_error(CZ_E_OOM, "out of device memory");
    if (hipMemcpy(d.p, &h, 4, hipMemcpyHostToDevice) != hipSuccess) return mine ? mine : cz::set_error(CZ_E_HIP, "status upload failed");
    int rc = cz::comm_all_reduce(comm, d.p, 1, cz::COMM_U32, cz::COMM_SUM, nullptr);
    if (rc) return mine ? mine : rc;
    if (hipMemcpy(&h, d.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return mine ? mine : cz::set_error(CZ_E_HIP, "status download failed");
    if (mine) return cz::set_error(mine, "%s", my_msg.c_str());
    if (h) return cz::set_error(CZ_E_INVALID, "%u other rank(s) rejected their shard of this call", h);
    return CZ_OK;
}

}  // namespace

extern "C" int cz_bfs_sharded(cz_comm *comm, const uint32_t *out_offsets_local, const uint32_t *out_targets, uint32_t N,
                              uint32_t row_begin, uint32_t row_end, uint64_t E_local, const uint32_t *starts, uint32_t n_starts,
                              const uint32_t *goals, uint32_t n_goals, int share_visited, uint32_t *parent, uint32_t *depth,
                              uint32_t *order, uint32_t *n_reached, const volatile uint8_t *poison) {
    if (!comm) return cz::set_error(CZ_E_INVALID, "null communicator");
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !parent) return cz::set_error(CZ_E_INVALID, "null starts/parent");
    rc = check_shard(out_offsets_local, out_targets, N, row_begin, row_end, E_local);
    HipShardedBfs b;
    b.comm = comm;
    b.s = nullptr;
    b.N = N;
    b.rb = row_begin;
    b.re = row_end;
    b.goals_host = goals;
    b.n_goals = goals ? n_goals : 0;
    if (!rc) rc = b.alloc(out_offsets_local, out_targets, E_local);
    if ((rc = collective_status(comm, rc))) return rc;
    for (uint32_t si = 0; si < n_starts; si++) {
        uint32_t reached = 0;
        const bool skip = goals && n_goals == 0;  // nothing pending: the reference discovers nothing useful
        rc = skip ? b.bfs_reset(share_visited != 0 && si > 0)
                  : czs::run_sharded_bfs(b, starts[si], N, goals != nullptr, share_visited != 0 && si > 0, poison, &reached);
        if (rc == czs::TRAVERSAL_CANCELLED) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        if (rc) return rc;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sharded bfs launch: %s", hipGetErrorString(e));
        CZ_HIP(hipMemcpy(parent + (size_t)si * N, b.parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        if (depth) CZ_HIP(hipMemcpy(depth + (size_t)si * N, b.depth.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        if (order) {
            for (uint32_t i = 0; i < N; i++) order[(size_t)si * N + i] = CZ_NONE;
            if (reached) CZ_HIP(hipMemcpy(order + (size_t)si * N, b.order.p + 1, (size_t)reached * 4, hipMemcpyDeviceToHost));
        }
        if (n_reached) n_reached[si] = reached;
    }
    return CZ_OK;
}

extern "C" int cz_sssp_sharded(cz_comm *comm, const uint32_t *out_offsets_local, const uint32_t *out_targets, const float *weights,
                               uint32_t N, uint32_t row_begin, uint32_t row_end, uint64_t E_local, const uint32_t *starts,
                               uint32_t n_starts, float *dist, uint32_t *parent, const volatile uint8_t *poison) {
    if (!comm) return cz::set_error(CZ_E_INVALID, "null communicator");
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    rc = check_shard(out_offsets_local, out_targets, N, row_begin, row_end, E_local);
    if (!rc && E_local > 0 && !weights) rc = cz::set_error(CZ_E_INVALID, "null weights");
    for (uint64_t e = 0; !rc && e < E_local; e++)
        if (!(weights[e] >= 0.0f))
            rc = cz::set_error(CZ_E_INVALID, "edge %llu has weight %g: weights must be non-negative numbers", (unsigned long long)e,
                               (double)weights[e]);
    HipShardedSssp b;
    b.comm = comm;
    b.s = nullptr;
    b.N = N;
    b.rb = row_begin;
    b.re = row_end;
    if (!rc) rc = b.alloc(out_offsets_local, out_targets, weights, E_local);
    if ((rc = collective_status(comm, rc))) return rc;
    if ((rc = b.agree_on_delta(weights, E_local))) return rc;
    for (uint32_t si = 0; si < n_starts; si++) {
        rc = czs::run_sharded_sssp(b, starts[si], N, poison);
        if (rc == czs::TRAVERSAL_CANCELLED) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        if (rc) return rc;
        hipLaunchKernelGGL(sssp_unpack_kernel, dim3(grid_for(N)), dim3(kT), 0, b.s, b.dp, b.canon.p, (uint64_t)N, b.dist.p, b.parent.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sharded sssp launch: %s", hipGetErrorString(e));
        CZ_HIP(hipMemcpy(dist + (size_t)si * N, b.dist.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        CZ_HIP(hipMemcpy(parent + (size_t)si * N, b.parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        t_sssp_sharded_stats[0] = b.rounds;
        t_sssp_sharded_stats[1] = b.pairs_exchanged;
        t_sssp_sharded_stats[2] = b.compactions;
        t_sssp_sharded_stats[3] = b.buckets;
    }
    return CZ_OK;
}

extern "C" int cz_sssp_sharded_last_stats(uint64_t *out4) {
    if (!out4) return cz::set_error(CZ_E_INVALID, "null out");
    for (int i = 0; i < 4; i++) out4[i] = t_sssp_sharded_stats[i];
    return CZ_OK;
}

// ---- ConnectedComponents over a vertex partition (czs::run_sharded_cc): an all-reduce(min) of the N pointers per round ----------
namespace {

__global__ void __launch_bounds__(kT)
cc_link_rows_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t rb, uint32_t re, uint32_t *__restrict__ comp) {
    const uint32_t glane = threadIdx.x & (kCcLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kCcLanes, ngroups = gridDim.x * blockDim.x / kCcLanes;
    for (uint32_t u = rb + group; u < re; u += ngroups) {
        const uint32_t e1 = off[u - rb + 1];
        for (uint32_t e = off[u - rb] + glane; e < e1; e += kCcLanes) cc_link(u, tgt[e], comp);
    }
}

__global__ void __launch_bounds__(kT)
cc_differs_kernel(const uint32_t *__restrict__ a, uint32_t *__restrict__ prev, uint32_t N, uint32_t *__restrict__ flag) {
    bool d = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const uint32_t x = a[i];
        if (x != prev[i]) {
            d = true;
            prev[i] = x;
        }
    }
    if (d) *flag = 1;
}

struct HipShardedCc {
    cz_comm *comm;
    hipStream_t s;
    uint32_t N, rb, re;
    cz::DevBuf<uint32_t> off, tgt, comp, prev, flag, rank, scratch, misc;

    int alloc(const uint32_t *h_off, const uint32_t *h_tgt, uint64_t e_local) {
        const uint32_t rows = re - rb;
        CZ_HIP(off.alloc((size_t)rows + 1));
        CZ_HIP(tgt.alloc(e_local));
        CZ_HIP(comp.alloc(N));
        CZ_HIP(prev.alloc(N));
        CZ_HIP(flag.alloc(N));
        CZ_HIP(rank.alloc(N));
        CZ_HIP(scratch.alloc(scan_scratch_words(N)));
        CZ_HIP(misc.alloc(4));
        CZ_HIP(hipMemcpy(off.p, h_off, ((size_t)rows + 1) * 4, hipMemcpyHostToDevice));
        if (e_local) CZ_HIP(hipMemcpy(tgt.p, h_tgt, e_local * 4, hipMemcpyHostToDevice));
        return CZ_OK;
    }
    int any_poisoned(bool mine, bool *any) {
        const uint32_t v = mine ? 1u : 0u;
        CZ_HIP(hipMemcpyAsync(misc.p + 2, &v, 4, hipMemcpyHostToDevice, s));
        int rc = cz::comm_all_reduce(comm, misc.p + 2, 1, cz::COMM_U32, cz::COMM_SUM, s);
        if (rc) return rc;
        uint32_t h = 0;
        CZ_HIP(hipMemcpyAsync(&h, misc.p + 2, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        *any = h != 0;
        return CZ_OK;
    }
    int cc_init() {
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(N)), dim3(kT), 0, s, comp.p, N);
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(N)), dim3(kT), 0, s, prev.p, N);
        return CZ_OK;
    }
    int cc_local_round() {
        if (re > rb)
            hipLaunchKernelGGL(cc_link_rows_kernel, dim3(grid_for((uint64_t)(re - rb) * kCcLanes)), dim3(kT), 0, s, off.p, tgt.p, rb, re, comp.p);
        hipLaunchKernelGGL(cc_compress_kernel, dim3(grid_for(N)), dim3(kT), 0, s, N, comp.p);
        return CZ_OK;
    }
    int reduce_labels() { return cz::comm_all_reduce(comm, comp.p, N, cz::COMM_U32, cz::COMM_MIN, s); }
    int cc_settle(bool *changed) {
        hipLaunchKernelGGL(cc_compress_kernel, dim3(grid_for(N)), dim3(kT), 0, s, N, comp.p);
        CZ_HIP(hipMemsetAsync(misc.p, 0, 4, s));
        hipLaunchKernelGGL(cc_differs_kernel, dim3(grid_for(N)), dim3(kT), 0, s, comp.p, prev.p, N, misc.p);
        uint32_t h = 0;
        CZ_HIP(hipMemcpyAsync(&h, misc.p, 4, hipMemcpyDeviceToHost, s));
        CZ_HIP(hipStreamSynchronize(s));
        *changed = h != 0;
        return CZ_OK;
    }
    int cc_number_groups() {
        hipLaunchKernelGGL(cc_rootflag_kernel, dim3(grid_for(N)), dim3(kT), 0, s, N, comp.p, flag.p);
        int rc = exclusive_scan(flag.p, rank.p, N, misc.p, scratch.p, s);
        if (rc) return rc;
        hipLaunchKernelGGL(cc_group_kernel, dim3(grid_for(N)), dim3(kT), 0, s, N, comp.p, rank.p, flag.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sharded cc launch: %s", hipGetErrorString(e));
        return CZ_OK;
    }
};

}  // namespace

extern "C" int cz_connected_components_sharded(cz_comm *comm, const uint32_t *offsets_local, const uint32_t *targets, uint32_t N,
                                               uint32_t row_begin, uint32_t row_end, uint64_t E_local, uint32_t *group,
                                               uint32_t *n_groups, uint32_t *rounds, const volatile uint8_t *poison) {
    if (n_groups) *n_groups = 0;
    if (rounds) *rounds = 0;
    if (!comm) return cz::set_error(CZ_E_INVALID, "null communicator");
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (N == 0) return CZ_OK;
    if (!group) return cz::set_error(CZ_E_INVALID, "null group");
    rc = check_shard(offsets_local, targets, N, row_begin, row_end, E_local);
    HipShardedCc b;
    b.comm = comm;
    b.s = nullptr;
    b.N = N;
    b.rb = row_begin;
    b.re = row_end;
    if (!rc) rc = b.alloc(offsets_local, targets, E_local);
    if ((rc = collective_status(comm, rc))) return rc;
    rc = czs::run_sharded_cc(b, poison, rounds);
    if (rc == czs::TRAVERSAL_CANCELLED) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    if (rc) return rc;
    CZ_HIP(hipMemcpy(group, b.flag.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    uint32_t total = 0;
    CZ_HIP(hipMemcpy(&total, b.misc.p, 4, hipMemcpyDeviceToHost));
    if (n_groups) *n_groups = total;
    return CZ_OK;
}

