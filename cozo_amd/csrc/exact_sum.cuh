// exact_sum.cuh -- the reference's SEQUENTIAL f32 row sum, evaluated by a whole wave, bit for bit.
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

// one term against a sum whose folded exponent is eb_s (and which is positive or zero)
CZ_XS_FN Inc term_inc(uint32_t abits, uint32_t eb_s) {
    Inc r;
    uint32_t ea = (abits >> 23) & 0xffu;
    uint32_t ma = abits & 0x7fffffu;
    const bool negative = (abits >> 31) != 0 && (abits << 1) != 0;  // -0.0 adds nothing to a positive sum
    if (negative || ea == 255u || eb_s == 255u) {
        r.even = r.odd = kSat;
        return r;
    }
    if (ea) ma |= 0x800000u;
    else ea = 1;
    if (ea >= eb_s) {
        const uint32_t sh = ea - eb_s;
        const uint32_t q = sh >= 3u ? kSat : (ma << sh);  // ma < 2^24: sh <= 2 stays below 2^26
        r.even = r.odd = q < kSat ? q : kSat;
        return r;
    }
    uint32_t sh = eb_s - ea;
    if (sh > 25u) sh = 25u;  // ma / 2^25 < 1/2: adds nothing
    const uint32_t q = ma >> sh;
    const uint32_t rem = ma & ((1u << sh) - 1u);
    const uint32_t half = 1u << (sh - 1u);
    if (rem > half) r.even = r.odd = q + 1u;
    else if (rem < half) r.even = r.odd = q;
    else {  // tie: to the even one of M + q, M + q + 1
        r.even = q + (q & 1u);
        r.odd = q + ((q & 1u) ^ 1u);
    }
    return r;
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

// Adds t[0..n) to s one after the other in f32 -- the value of `for (i) s = s + t[i]` -- with a whole wave.
// Every lane of the wave calls it with the same arguments (t in LDS or global memory); the result is uniform.
// T = terms per lane and pass.  PRO: a row that starts from nothing doubles its sum -- leaves its binade -- with almost
// every term at first, so the first PRO terms of a row of >= 2 PRO terms are simply added (by every lane, redundantly).
template <int T, int PRO = 32>
__device__ __forceinline__ float wave_seq_sum(const float *t, uint32_t n, float s) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t p = 0;
    if (PRO > 0 && n >= 2u * PRO) {
#pragma unroll
        for (int j = 0; j < PRO; j++) s = s + t[j];
        p = PRO;
    }
    while (p < n) {
        const uint32_t rem = n - p;
        // short remainders are spread over all lanes (per = terms per lane of this pass)
        const uint32_t per = rem >= 64u * T ? (uint32_t)T : (rem + 63u) / 64u;
        const uint32_t first = p + lane * per;
        uint32_t a[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            const uint32_t i = first + j;
            a[j] = ((uint32_t)j < per && i < n) ? __float_as_uint(t[i]) : 0u;  // +0.0 adds nothing
        }
        uint32_t M, eb;
        split(__float_as_uint(s), M, eb);
        const bool s_ok = (__float_as_uint(s) >> 31) == 0 || (__float_as_uint(s) << 1) == 0;  // a negative sum: true additions only
        Inc f = term_inc(a[0], eb);
#pragma unroll
        for (int j = 1; j < T; j++) f = then(f, term_inc(a[j], eb));
        if (!s_ok) f.even = f.odd = kSat;
        // inclusive scan over the lanes (lower lanes first), then the exclusive prefix
        Inc g = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Inc lo;
            lo.even = __shfl_up(g.even, o, 64);
            lo.odd = __shfl_up(g.odd, o, 64);
            if (lane >= (uint32_t)o) g = then(lo, g);
        }
        Inc ex;
        ex.even = __shfl_up(g.even, 1, 64);
        ex.odd = __shfl_up(g.odd, 1, 64);
        if (lane == 0) ex.even = ex.odd = 0;
        const uint32_t m0 = apply(M, ex);   // this lane's starting M (exact while below kLimit)
        const uint32_t m1 = apply(m0, f);   // ... and where it ends
        const unsigned long long out = __ballot(m1 >= kLimit);
        if (out == 0ull) {
            s = __uint_as_float(join(__shfl(m1, 63, 64), eb));
            p += 64u * per;
        } else {
            const uint32_t L = (uint32_t)__ffsll((long long)out) - 1u;
            // lane 0 starts from s itself (also when s is negative, inf or nan: then every lane is "out" and L = 0)
            float sl = lane == 0 ? s : __uint_as_float(join(m0 < kLimit ? m0 : 0u, eb));
#pragma unroll
            for (int j = 0; j < T; j++) sl = sl + __uint_as_float(a[j]);  // padding is +0.0
            s = __shfl(sl, (int)L, 64);
            p += (L + 1u) * per;
        }
    }
    return s;
}

}  // namespace cz_exact
#endif
