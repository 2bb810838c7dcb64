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
//   unique fixpoint of monotone relaxation, so a frontier Bellman-Ford that relaxes with the same f32 add and
//   strict `<` reaches bit-identical costs.  (cost, parent) are packed in one u64 and updated by CAS only on a
//   strict improvement, which keeps every parent pointer tight and the predecessor graph a tree.
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "common.h"

namespace {

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
constexpr int kBfsLanes = 16;

__device__ __forceinline__ unsigned int bfs_group_mask(unsigned long long ballot, int lane) {
    return (unsigned int)(ballot >> (lane & ~(kBfsLanes - 1))) & ((1u << kBfsLanes) - 1u);
}

__global__ void __launch_bounds__(kT)
bfs_claim_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                 uint32_t fsize, const uint32_t *__restrict__ depth, uint32_t *__restrict__ claim) {
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    for (uint32_t i = group; i < fsize; i += ngroups) {
        const uint32_t u = frontier[i];
        const uint32_t e1 = off[u + 1];
        for (uint32_t e = off[u] + glane; e < e1; e += kBfsLanes) {
            const uint32_t v = tgt[e];
            if (depth[v] == CZ_NONE) atomicMin(&claim[v], i);
        }
    }
}

__global__ void __launch_bounds__(kT)
bfs_count_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                 uint32_t fsize, const uint32_t *__restrict__ depth, const uint32_t *__restrict__ claim,
                 uint32_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;  // every group of a wave runs the same trip count (ballots)
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < fsize;
        const uint32_t u = live ? frontier[i] : 0;
        const uint32_t e0 = live ? off[u] : 0, e1 = live ? off[u + 1] : 0;
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
                hit = (e == e0 || tgt[e - 1] != v) && depth[v] == CZ_NONE && claim[v] == i;
            }
            c += __popc(bfs_group_mask(__ballot(hit), lane));
        }
        if (live && glane == 0) cnt[i] = c;
    }
}

__global__ void __launch_bounds__(kT)
bfs_emit_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ frontier,
                uint32_t fsize, uint32_t *__restrict__ depth, const uint32_t *__restrict__ claim,
                const uint32_t *__restrict__ pos, uint32_t *__restrict__ next, uint32_t *__restrict__ parent,
                uint32_t next_depth) {
    const int lane = threadIdx.x & 63;
    const uint32_t glane = threadIdx.x & (kBfsLanes - 1);
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) / kBfsLanes, ngroups = gridDim.x * blockDim.x / kBfsLanes;
    const uint32_t rounds = (fsize + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = group + r * ngroups;
        const bool live = i < fsize;
        const uint32_t u = live ? frontier[i] : 0;
        const uint32_t e0 = live ? off[u] : 0, e1 = live ? off[u + 1] : 0;
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
                hit = (e == e0 || tgt[e - 1] != v) && claim[v] == i && depth[v] == CZ_NONE;
            }
            const unsigned int m = bfs_group_mask(__ballot(hit), lane);
            if (hit) {
                next[o + __popc(m & ((1u << glane) - 1u))] = v;
                parent[v] = u;
                depth[v] = next_depth;
            }
            o += __popc(m);
        }
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

// ---------------------------------------------------------------------------------------------
// connected components
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kT) iota_kernel(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

__global__ void __launch_bounds__(kT)
cc_relax_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t N, uint32_t *__restrict__ label,
                uint32_t *__restrict__ changed) {
    bool ch = false;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < N; u += gridDim.x * blockDim.x) {
        uint32_t lu = label[u], m = lu;
        for (uint32_t e = off[u]; e < off[u + 1]; e++) {
            const uint32_t v = tgt[e];
            const uint32_t lv = label[v];
            if (lv < m) m = lv;
            if (lu < lv) {
                if (atomicMin(&label[v], lu) > lu) ch = true;
            }
        }
        if (m < lu) {
            if (atomicMin(&label[u], m) > m) ch = true;
            // hook the old representative too, so whole trees move at once
            if (atomicMin(&label[lu], m) > m) ch = true;
        }
    }
    if (ch) *changed = 1;
}

__global__ void __launch_bounds__(kT) cc_jump_kernel(uint32_t N, uint32_t *__restrict__ label) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        uint32_t l = label[i];
        for (;;) {
            const uint32_t ll = label[l];
            if (ll == l) break;
            l = ll;
        }
        label[i] = l;
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

__global__ void __launch_bounds__(kT) fill_u64_kernel(unsigned long long *__restrict__ p, uint32_t n, unsigned long long v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

__global__ void __launch_bounds__(kT)
sssp_relax_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w,
                  const uint32_t *__restrict__ frontier, uint32_t fsize, unsigned long long *__restrict__ dp,
                  uint32_t *__restrict__ queued, uint32_t round_tag, uint32_t *__restrict__ next,
                  uint32_t *__restrict__ next_count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < fsize; i += gridDim.x * blockDim.x) {
        const uint32_t u = frontier[i];
        const unsigned long long cu = __hip_atomic_load(&dp[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float du = __uint_as_float((uint32_t)(cu >> 32));
        for (uint32_t e = off[u]; e < off[u + 1]; e++) {
            const uint32_t v = tgt[e];
            const float nd = du + w[e];  // `cost + path_weight` in f32 (shortest_path_dijkstra.rs:303)
            const uint32_t nb = __float_as_uint(nd);
            unsigned long long cur = __hip_atomic_load(&dp[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (nb < (uint32_t)(cur >> 32)) {  // strict `<` (:304); non-negative floats order as their bits
                const unsigned long long want = ((unsigned long long)nb << 32) | u;
                const unsigned long long seen = atomicCAS(&dp[v], cur, want);
                if (seen == cur) {
                    if (atomicExch(&queued[v], round_tag) != round_tag) next[atomicAdd(next_count, 1u)] = v;
                    break;
                }
                cur = seen;
            }
        }
    }
}

__global__ void __launch_bounds__(kT)
sssp_unpack_kernel(const unsigned long long *__restrict__ dp, uint32_t N, float *__restrict__ dist, uint32_t *__restrict__ parent) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const unsigned long long c = dp[i];
        dist[i] = __uint_as_float((uint32_t)(c >> 32));
        parent[i] = (uint32_t)c;
    }
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

extern "C" int cz_bfs(const uint32_t *out_offsets, const uint32_t *out_targets, uint32_t N, uint64_t E,
                      const uint32_t *starts, uint32_t n_starts, const uint32_t *goals, uint32_t n_goals,
                      int share_visited, uint32_t *parent, uint32_t *depth, uint32_t *order, uint32_t *n_reached,
                      const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !parent) return cz::set_error(CZ_E_INVALID, "null starts/parent");
    rc = check_csr(out_offsets, out_targets, N, E);
    if (rc) return rc;
    cz::DevBuf<uint32_t> d_off, d_tgt, d_depth, d_parent, d_claim, d_order, d_cnt, d_pos, d_scratch, d_goals, d_misc;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_tgt.alloc(E));
    CZ_HIP(d_depth.alloc(N));
    CZ_HIP(d_parent.alloc(N));
    CZ_HIP(d_claim.alloc(N));
    CZ_HIP(d_order.alloc((size_t)N + 1));
    CZ_HIP(d_cnt.alloc(N));
    CZ_HIP(d_pos.alloc(N));
    CZ_HIP(d_scratch.alloc(scan_scratch_words(N)));
    CZ_HIP(d_misc.alloc(4));
    CZ_HIP(hipMemcpy(d_off.p, out_offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) CZ_HIP(hipMemcpy(d_tgt.p, out_targets, E * 4, hipMemcpyHostToDevice));
    if (goals && n_goals) {
        CZ_HIP(d_goals.alloc(n_goals));
        CZ_HIP(hipMemcpy(d_goals.p, goals, (size_t)n_goals * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = nullptr;
    CZ_HIP(hipMemsetAsync(d_depth.p, 0xFF, (size_t)N * 4, s));
    CZ_HIP(hipMemsetAsync(d_claim.p, 0xFF, (size_t)N * 4, s));
    for (uint32_t si = 0; si < n_starts; si++) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        const uint32_t start = starts[si];
        uint32_t reached = 0;
        if (!share_visited && si > 0) {
            CZ_HIP(hipMemsetAsync(d_depth.p, 0xFF, (size_t)N * 4, s));
            CZ_HIP(hipMemsetAsync(d_claim.p, 0xFF, (size_t)N * 4, s));
        }
        CZ_HIP(hipMemsetAsync(d_parent.p, 0xFF, (size_t)N * 4, s));
        bool run = start < N;
        if (run && share_visited) {
            uint32_t dstart;
            CZ_HIP(hipMemcpy(&dstart, d_depth.p + start, 4, hipMemcpyDeviceToHost));
            run = dstart == CZ_NONE;  // algos/bfs.rs:52-54 already visited => skip
        }
        if (run && goals && n_goals == 0) run = false;  // nothing pending: the reference discovers nothing useful
        if (run) {
            hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, d_depth.p, start, 0u);
            hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, s, d_order.p, 0u, start);
            uint32_t lo = 0, fsize = 1, level = 0;
            while (fsize > 0) {
                if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
                const uint32_t *fr = d_order.p + lo;
                const int g = grid_for((uint64_t)fsize * kBfsLanes);  // a 16-lane group per frontier node
                hipLaunchKernelGGL(bfs_claim_kernel, dim3(g), dim3(kT), 0, s, d_off.p, d_tgt.p, fr, fsize, d_depth.p, d_claim.p);
                hipLaunchKernelGGL(bfs_count_kernel, dim3(g), dim3(kT), 0, s, d_off.p, d_tgt.p, fr, fsize, d_depth.p, d_claim.p,
                                   d_cnt.p);
                rc = exclusive_scan(d_cnt.p, d_pos.p, fsize, d_misc.p, d_scratch.p, s);
                if (rc) return rc;
                hipLaunchKernelGGL(bfs_emit_kernel, dim3(g), dim3(kT), 0, s, d_off.p, d_tgt.p, fr, fsize, d_depth.p, d_claim.p,
                                   d_pos.p, d_order.p + lo + fsize, d_parent.p, level + 1);
                CZ_HIP(hipMemsetAsync(d_misc.p + 1, 0, 4, s));
                if (goals)
                    hipLaunchKernelGGL(bfs_goals_left_kernel, dim3(grid_for(n_goals)), dim3(kT), 0, s, d_goals.p, n_goals, N,
                                       d_depth.p, start, d_misc.p + 1);
                uint32_t h[2];
                CZ_HIP(hipMemcpy(h, d_misc.p, 8, hipMemcpyDeviceToHost));
                lo += fsize;
                fsize = h[0];
                reached += fsize;
                level++;
                if (goals && h[1] == 0) break;
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "bfs launch: %s", hipGetErrorString(e));
        }
        CZ_HIP(hipMemcpy(parent + (size_t)si * N, d_parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        if (depth) {
            CZ_HIP(hipMemcpy(depth + (size_t)si * N, d_depth.p, (size_t)N * 4, hipMemcpyDeviceToHost));
            if (!run && !share_visited)
                for (uint32_t i = 0; i < N; i++) depth[(size_t)si * N + i] = CZ_NONE;
        }
        if (order && reached) CZ_HIP(hipMemcpy(order + (size_t)si * N, d_order.p + 1, (size_t)reached * 4, hipMemcpyDeviceToHost));
        if (n_reached) n_reached[si] = reached;
    }
    return CZ_OK;
}

extern "C" int cz_connected_components(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E,
                                       uint32_t *group, uint32_t *n_groups, const volatile uint8_t *poison) {
    if (n_groups) *n_groups = 0;
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (N == 0) return CZ_OK;
    if (!group) return cz::set_error(CZ_E_INVALID, "null group");
    rc = check_csr(offsets, targets, N, E);
    if (rc) return rc;
    cz::DevBuf<uint32_t> d_off, d_tgt, d_label, d_flag, d_rank, d_scratch, d_misc;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_tgt.alloc(E));
    CZ_HIP(d_label.alloc(N));
    CZ_HIP(d_flag.alloc(N));
    CZ_HIP(d_rank.alloc(N));
    CZ_HIP(d_scratch.alloc(scan_scratch_words(N)));
    CZ_HIP(d_misc.alloc(4));
    CZ_HIP(hipMemcpy(d_off.p, offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) CZ_HIP(hipMemcpy(d_tgt.p, targets, E * 4, hipMemcpyHostToDevice));
    hipStream_t s = nullptr;
    const int g = grid_for(N);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kT), 0, s, d_label.p, N);
    for (;;) {
        if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
        CZ_HIP(hipMemsetAsync(d_misc.p, 0, 4, s));
        hipLaunchKernelGGL(cc_relax_kernel, dim3(g), dim3(kT), 0, s, d_off.p, d_tgt.p, N, d_label.p, d_misc.p);
        hipLaunchKernelGGL(cc_jump_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p);
        uint32_t changed = 0;
        CZ_HIP(hipMemcpy(&changed, d_misc.p, 4, hipMemcpyDeviceToHost));
        if (!changed) break;
    }
    hipLaunchKernelGGL(cc_rootflag_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p, d_flag.p);
    rc = exclusive_scan(d_flag.p, d_rank.p, N, d_misc.p, d_scratch.p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cc_group_kernel, dim3(g), dim3(kT), 0, s, N, d_label.p, d_rank.p, d_flag.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "cc launch: %s", hipGetErrorString(e));
    CZ_HIP(hipMemcpy(group, d_flag.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    uint32_t total = 0;
    CZ_HIP(hipMemcpy(&total, d_misc.p, 4, hipMemcpyDeviceToHost));
    if (n_groups) *n_groups = total;
    return CZ_OK;
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
        if (d >= 2) {
            // pairs (i, j), j < i suffices: the list is ascending, so A[i] > A[j] implies j < i
            for (uint32_t i = 1; i < d; i++) {
                const uint32_t src = tgt[a + i];
                const uint32_t lo = off[src], hi = off[src + 1];
                for (uint32_t j = lane; j < i; j += 64) {
                    const uint32_t dst = tgt[a + j];
                    if (dst < src && csr_contains(tgt, lo, hi, dst)) cnt++;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) {
            n_tri[v] = cnt;
            degree[v] = d;
        }
    }
}

extern "C" int cz_clustering_coefficients(const uint32_t *offsets, const uint32_t *targets, uint32_t N, uint64_t E,
                                          uint64_t *n_triangles, uint32_t *degree, const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
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
    hipLaunchKernelGGL(triangles_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, nullptr, d_off.p, d_tgt.p, N, d_tri.p, d_deg.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "triangles launch: %s", hipGetErrorString(e));
    CZ_HIP(hipMemcpy(n_triangles, d_tri.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    CZ_HIP(hipMemcpy(degree, d_deg.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
    return CZ_OK;
}

extern "C" int cz_sssp(const uint32_t *out_offsets, const uint32_t *out_targets, const float *weights, uint32_t N,
                       uint64_t E, const uint32_t *starts, uint32_t n_starts, float *dist, uint32_t *parent,
                       const volatile uint8_t *poison) {
    int rc = cz::ensure_device();
    if (rc) return rc;
    if (n_starts == 0 || N == 0) return CZ_OK;
    if (!starts || !dist || !parent) return cz::set_error(CZ_E_INVALID, "null starts/dist/parent");
    rc = check_csr(out_offsets, out_targets, N, E);
    if (rc) return rc;
    if (E > 0 && !weights) return cz::set_error(CZ_E_INVALID, "null weights");
    for (uint64_t e = 0; e < E; e++)  // BadEdgeWeightError, fixed_rule/mod.rs:258-286
        if (!(weights[e] >= 0.0f) || !std::isfinite(weights[e]))
            return cz::set_error(CZ_E_INVALID, "edge %llu has weight %g: weights must be finite and non-negative",
                                 (unsigned long long)e, (double)weights[e]);
    cz::DevBuf<uint32_t> d_off, d_tgt, d_queued, d_f0, d_f1, d_misc, d_parent;
    cz::DevBuf<float> d_w, d_dist;
    cz::DevBuf<unsigned long long> d_dp;
    CZ_HIP(d_off.alloc((size_t)N + 1));
    CZ_HIP(d_tgt.alloc(E));
    CZ_HIP(d_w.alloc(E));
    CZ_HIP(d_queued.alloc(N));
    CZ_HIP(d_f0.alloc(N));
    CZ_HIP(d_f1.alloc(N));
    CZ_HIP(d_misc.alloc(4));
    CZ_HIP(d_parent.alloc(N));
    CZ_HIP(d_dist.alloc(N));
    CZ_HIP(d_dp.alloc(N));
    CZ_HIP(hipMemcpy(d_off.p, out_offsets, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    if (E) {
        CZ_HIP(hipMemcpy(d_tgt.p, out_targets, E * 4, hipMemcpyHostToDevice));
        CZ_HIP(hipMemcpy(d_w.p, weights, E * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = nullptr;
    const int gN = grid_for(N);
    for (uint32_t si = 0; si < n_starts; si++) {
        const uint32_t start = starts[si];
        hipLaunchKernelGGL(fill_u64_kernel, dim3(gN), dim3(kT), 0, s, d_dp.p, N, kInfPacked);
        CZ_HIP(hipMemsetAsync(d_queued.p, 0, (size_t)N * 4, s));
        if (start < N) {
            const unsigned long long zero = 0x00000000FFFFFFFFull;  // cost 0.0, no parent
            CZ_HIP(hipMemcpyAsync(d_dp.p + start, &zero, 8, hipMemcpyHostToDevice, s));
            CZ_HIP(hipMemcpyAsync(d_f0.p, &start, 4, hipMemcpyHostToDevice, s));
            CZ_HIP(hipStreamSynchronize(s));
            uint32_t fsize = 1, round = 1;
            uint32_t *cur = d_f0.p, *nxt = d_f1.p;
            while (fsize > 0) {
                if (cz::poisoned(poison)) return cz::set_error(CZ_E_CANCELLED, "cancelled");
                CZ_HIP(hipMemsetAsync(d_misc.p, 0, 4, s));
                hipLaunchKernelGGL(sssp_relax_kernel, dim3(grid_for(fsize)), dim3(kT), 0, s, d_off.p, d_tgt.p, d_w.p, cur, fsize,
                                   d_dp.p, d_queued.p, round, nxt, d_misc.p);
                CZ_HIP(hipMemcpy(&fsize, d_misc.p, 4, hipMemcpyDeviceToHost));
                std::swap(cur, nxt);
                round++;
            }
        }
        hipLaunchKernelGGL(sssp_unpack_kernel, dim3(gN), dim3(kT), 0, s, d_dp.p, N, d_dist.p, d_parent.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return cz::set_error(CZ_E_HIP, "sssp launch: %s", hipGetErrorString(e));
        CZ_HIP(hipMemcpy(dist + (size_t)si * N, d_dist.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        CZ_HIP(hipMemcpy(parent + (size_t)si * N, d_parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    }
    return CZ_OK;
}
