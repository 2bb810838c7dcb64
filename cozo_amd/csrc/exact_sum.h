// exact_sum.h -- the reference's SEQUENTIAL f32 row sum, evaluated by a whole wave, bit for bit.
//
// graph 0.3.1 `page_rank` adds a node's in-neighbour contributions one after the other in f32
// (oracle/cozo_oracle.c orc_pagerank).  fl(s + a) is not associative, so until round 3 one
// LANE walked each row (a 10^5-term hub row = a 10^5-long dependent v_add_f32 chain, ~14 cycles per term, and a skewed
// graph's sweep waited for it).  The chain is only sequential in the BINADE of the running sum, though:
//
//   while s stays inside one binade [2^E, 2^(E+1)) it is an integer multiple M * u of u = ulp(s) = 2^(E-23) with
//   2^23 <= M < 2^24, and round-to-nearest-even of s + a is  (M + rne(a / u)) * u :  adding a term adds an INTEGER to M.
//   rne(M + k) for real k = q + f (q integer, 0 <= f < 1) is M + q (f < 1/2), M + q + 1 (f > 1/2), and on an exact tie
//   (f == 1/2) the even one of the two -- which depends on M only through its PARITY.  So a term is a function
//   "parity of M  ->  integer increment", a pair (inc_even, inc_odd); running one run of terms after another composes
//   two such pairs into one, and composition is associative: a run of terms can be cut over the lanes of a wave, every
//   lane composes its own piece, and a prefix scan over the lanes gives each lane the value of M it starts from.
//
// The binade is an assumption a pass has to check: terms are non-negative here (contributions of PageRank), so M only
// grows and the check is "did any lane end at or above 2^24".  The FIRST lane that does re-adds its few terms with
// real f32 additions from its (exact) starting value -- always right, whatever happens inside -- and the pass after
// it starts behind that lane with the new binade.  A sum of n positive terms crosses ~log2(n) binades, so a row costs
// n / (64 T) + O(log n) passes instead of n dependent additions.  Anything the integer view does not cover (negative
// terms, inf, nan, s = 0, a term far larger than s) is mapped to "at or above 2^24", i.e. onto the true additions.
//
// The functions below are plain integer code shared by the device kernels (pagerank.hip) and the host unit test
// (tests/cpp/exact_sum_test.cpp, which runs the wave procedure lane by lane on the CPU against a plain float loop).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CZ_XS_FN __host__ __device__ __forceinline__
#else
#define CZ_XS_FN static inline
#endif

namespace cz_exact {

constexpr uint32_t kLimit = 1u << 24;  // M at or above this: the sum has left the binade
constexpr uint32_t kSat = 1u << 26;    // increments saturate here (only "at or above kLimit" matters beyond it)

struct Inc {
    uint32_t even, odd;  // what the run adds to an even / an odd M
};

// s = M * 2^(eb - 150), eb = the biased exponent with denormals folded onto 1
CZ_XS_FN void split(uint32_t sbits, uint32_t &M, uint32_t &eb) {
    eb = (sbits >> 23) & 0xffu;
    M = sbits & 0x7fffffu;
    if (eb) M |= 0x800000u;
    else eb = 1;
}

// bits of M * 2^(eb - 150) for M < 2^24 (M >= 2^23 carries into the exponent field by itself)
CZ_XS_FN uint32_t join(uint32_t M, uint32_t eb) { return ((eb - 1u) << 23) + M; }

// one term against a sum whose folded exponent is eb_s (and which is positive or zero); straight-line code.
// a / u = q + f with 0 <= f < 1 (u = the sum's ulp): `rup` rounds halves up, and on an exact tie the composition steps
// back by one when that makes M + increment even (`tie`).  Everything else is "past the binade" (rup = kSat), i.e. left
// to the true additions: a term at or above the sum's own exponent (it takes a normal sum out of its binade anyway),
// negative terms (-0.0 included: the sign bit makes ea0 > 255), inf and nan (ea0 = 255 >= any eb_s).
struct Term {
    uint32_t rup, tie;  // tie: 1 when a / u is exactly q + 1/2 (then rup = q + 1)
};
CZ_XS_FN uint32_t funnel_low(uint32_t ma, uint32_t sh) {  // the sh bits dropped by ma >> sh, left-aligned (1 <= sh <= 31)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(ma, 0u, sh);
#else
    return ma << (32u - sh);
#endif
}
CZ_XS_FN Term classify(uint32_t abits, uint32_t eb_s) {
    const uint32_t ea0 = abits >> 23;
    const uint32_t ea = ea0 ? ea0 : 1u;
    const int32_t d = (int32_t)eb_s - (int32_t)ea;
    const uint32_t ma = abits - ((ea - 1u) << 23);  // the mantissa with its implicit bit (none for a denormal)
    const uint32_t sh = (uint32_t)(d < 1 ? 1 : (d > 25 ? 25 : d));  // ma / 2^25 < 1/2: adds nothing
    const uint32_t low = funnel_low(ma, sh);
    Term t;
    t.rup = (ma >> sh) + (low >> 31);
    t.tie = low == 0x80000000u ? 1u : 0u;
    if (d <= 0) t.rup = kSat;  // kSat is even: a tie flag next to it changes nothing that matters
    return t;
}
// the term as an increment pair: M even -> M + increment must be even on a tie; M odd -> the increment must be odd
CZ_XS_FN Inc term_inc(uint32_t abits, uint32_t eb_s) {
    const Term t = classify(abits, eb_s);
    Inc r;
    r.even = t.rup - (t.tie & t.rup);
    r.odd = t.rup - (t.tie & (t.rup ^ 1u));
    return r;
}
// one more term on a run whose increment (for a start of known parity) is x: x + rup, stepped back to the even / odd
// neighbour on a tie (start_odd = 0: M + x must end even; 1: x must end odd)
CZ_XS_FN uint32_t add_term(uint32_t x, Term t, uint32_t start_odd) {
    const uint32_t y = x + t.rup;
    return y - (t.tie & (y ^ start_odd));
}

// the run f followed by the run g
CZ_XS_FN Inc then(Inc f, Inc g) {
    Inc h;
    const uint32_t e = f.even + ((f.even & 1u) ? g.odd : g.even);
    const uint32_t o = f.odd + ((f.odd & 1u) ? g.even : g.odd);  // an odd M plus an odd increment is even
    h.even = e < kSat ? e : kSat;
    h.odd = o < kSat ? o : kSat;
    return h;
}

CZ_XS_FN uint32_t apply(uint32_t M, Inc f) {
    const uint32_t r = M + ((M & 1u) ? f.odd : f.even);
    return r < kSat ? r : kSat;
}

}  // namespace cz_exact

#if defined(__HIPCC__)
namespace cz_exact {

// lane shifted reads for the scan: DPP row_shr:n inside a row of 16 lanes (lanes without a source read 0 = "adds
// nothing"), row_bcast:15 / row_bcast:31 to carry a row's total into the rows above it, wave_shr:1 for the exclusive form
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Inc dpp_fetch(Inc old, Inc v) {
    Inc r;
    r.even = __builtin_amdgcn_update_dpp(old.even, v.even, CTRL, ROW_MASK, 0xF, false);
    r.odd = __builtin_amdgcn_update_dpp(old.odd, v.odd, CTRL, ROW_MASK, 0xF, false);
    return r;
}

// inclusive scan of `then` over the lanes of a group (LANES = 16: one DPP row; 64: the wave), lower lanes first
template <int LANES>
__device__ __forceinline__ Inc scan_then(Inc g) {
    const Inc zero{0u, 0u};
    g = then(dpp_fetch<0x111, 0xF>(zero, g), g);  // row_shr:1
    g = then(dpp_fetch<0x112, 0xF>(zero, g), g);  // row_shr:2
    g = then(dpp_fetch<0x114, 0xF>(zero, g), g);  // row_shr:4
    g = then(dpp_fetch<0x118, 0xF>(zero, g), g);  // row_shr:8
    if (LANES == 64) {
        g = then(dpp_fetch<0x142, 0xA>(zero, g), g);  // row_bcast:15 into rows 1 and 3
        g = then(dpp_fetch<0x143, 0xC>(zero, g), g);  // row_bcast:31 into rows 2 and 3
    }
    return g;
}

// Adds t[0..n) to s one after the other in f32 -- the value of `for (i) s = s + t[i]` -- with a group of LANES lanes
// (16: four independent rows per wave, each group with its own t / n / s; 64: the whole wave on one row).  Every lane of
// the wave calls it (a group without a row passes n = 0); the result is uniform over the group.  t: 4-byte aligned.
// A pass takes about as many terms as the sum already holds -- the next binade is about that far away, and what lies
// behind a crossing is done again -- as 4, 8 or 16 (<= T) consecutive terms per lane, read as 16-byte vectors: a row of
// n terms costs ~n / LANES term steps plus ~2 log2(n) passes.
// PRO: a row that starts from nothing leaves its binade with almost every term at first, so the first PRO (+ 0..3, up to
// a 16-byte boundary of t) terms are simply added, by every lane of the group redundantly; rows too short for a pass
// after that are added that way entirely.
template <int LANES, int T, int PRO = 32>
__device__ __forceinline__ float group_seq_sum(const float *t, uint32_t n, float s) {
    static_assert(LANES == 16 || LANES == 64, "a DPP row or the wave");
    static_assert(T == 4 || T == 8 || T == 16, "increments of one lane must stay below 2^30; terms are read four at a time");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gl = lane & (LANES - 1);       // lane inside the group
    const uint32_t gbase = lane & ~(LANES - 1u);  // first lane of the group
    uint32_t p = PRO + ((4u - ((((uint32_t)(uintptr_t)t >> 2) + PRO) & 3u)) & 3u);  // t + p is 16-byte aligned
    if (n < 2u * PRO + 4u) p = n;
    {
        uint32_t pmax = p;
        if (LANES != 64) {
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) pmax = max(pmax, (uint32_t)__shfl_xor((int)pmax, o, 64));
        }
        for (uint32_t j = 0; j < pmax; j++)
            if (j < p) s = s + t[j];
    }
    while (__ballot(p < n) != 0ull) {  // groups whose row is done idle until the wave's longest row is
        const bool live = p < n;
        const uint32_t rem = live ? n - p : 0u;
        const uint32_t ext = min(min(rem, max(p, (uint32_t)LANES)), (uint32_t)(LANES * T));  // terms wanted from this pass
        const uint32_t per = (T >= 16 && ext > (uint32_t)LANES * 8u) ? 16u : (T >= 8 && ext > (uint32_t)LANES * 4u) ? 8u : 4u;
        const uint32_t used = (ext + per - 1u) / per;  // lanes of the group that take terms (the pass covers used * per)
        const uint32_t first = p + gl * per;
        uint32_t per_max = per;
        if (LANES != 64) {
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) per_max = max(per_max, (uint32_t)__shfl_xor((int)per_max, o, 64));
        }
        uint32_t a[T];
#pragma unroll
        for (int c = 0; c < T / 4; c++) {
            const uint32_t i = first + 4u * c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // +0.0 adds nothing
            if (gl < used && 4u * c < per && i < n) v = *(const float4 *)(t + i);
            a[4 * c + 0] = __float_as_uint(v.x);
            a[4 * c + 1] = i + 1u < n ? __float_as_uint(v.y) : 0u;  // the row's last vector may reach past its end
            a[4 * c + 2] = i + 2u < n ? __float_as_uint(v.z) : 0u;
            a[4 * c + 3] = i + 3u < n ? __float_as_uint(v.w) : 0u;
        }
        uint32_t M, eb;
        const uint32_t sbits = __float_as_uint(s);
        split(sbits, M, eb);
        uint32_t e = 0, o = 0;  // this lane's terms as ONE increment pair (composed in order; clamped once at the end)
#pragma unroll
        for (int c = 0; c < T / 4; c++)
            if (4u * c < per_max) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const Term x = classify(a[4 * c + u], eb);
                    e = add_term(e, x, 0u);
                    o = add_term(o, x, 1u);
                }
            }
        Inc f;
        f.even = e < kSat ? e : kSat;
        f.odd = o < kSat ? o : kSat;
        if (sbits > 0x7f7fffffu) f.even = f.odd = kSat;  // a negative / inf / nan sum: true additions only
        const Inc g = scan_then<LANES>(f);
        Inc ex = dpp_fetch<0x138, 0xF>(Inc{0u, 0u}, g);  // wave_shr:1: the lanes below
        if (gl == 0) ex.even = ex.odd = 0;
        const uint32_t m0 = apply(M, ex);  // this lane's starting M (exact while below kLimit)
        const uint32_t m1 = apply(m0, f);  // ... and where it ends
        const unsigned long long out_all = __ballot(live && m1 >= kLimit);
        const uint32_t out = LANES == 64 ? 0u : (uint32_t)(out_all >> gbase) & ((1u << (LANES & 31)) - 1u);
        const bool crossed = LANES == 64 ? out_all != 0ull : out != 0u;
        // no lane of the group left the binade: the last lane's end is the new sum
        const uint32_t m_last = LANES == 64 ? (uint32_t)__builtin_amdgcn_readlane((int)m1, 63) : (uint32_t)__shfl(m1, (int)(gbase + LANES - 1), 64);
        if (out_all != 0ull) {  // some group of the wave crossed: its first such lane re-adds its terms for real
            // lane 0 of a group starts from s itself (also when s is negative, inf or nan: then every lane is "out")
            float sl = gl == 0 ? s : __uint_as_float(join(m0 < kLimit ? m0 : 0u, eb));
#pragma unroll
            for (int c = 0; c < T / 4; c++)
                if (4u * c < per_max) {
#pragma unroll
                    for (int u = 0; u < 4; u++) sl = sl + __uint_as_float(a[4 * c + u]);  // padding is +0.0
                }
            const uint32_t L = LANES == 64 ? (uint32_t)__ffsll((long long)out_all) - 1u : (uint32_t)__ffs((int)out) - 1u;
            const float sx = LANES == 64 ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sl), (int)L))
                                         : __shfl(sl, (int)(gbase + (crossed ? L : 0u)), 64);
            if (live && crossed) {
                s = sx;
                p += (L + 1u) * per;
            }
        }
        if (live && !crossed) {
            s = __uint_as_float(join(m_last, eb));
            p += used * per;
        }
    }
    return s;
}

template <int T, int PRO = 32>
__device__ __forceinline__ float wave_seq_sum(const float *t, uint32_t n, float s) {
    return group_seq_sum<64, T, PRO>(t, n, s);
}

}  // namespace cz_exact
#endif
