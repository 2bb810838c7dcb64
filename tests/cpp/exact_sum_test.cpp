// exact_sum_test.cpp -- host check of cozo_amd/csrc/exact_sum.h: the wave procedure (run here lane by lane, same
// primitives, same control flow) against the plain sequential f32 loop it must equal bit for bit.
//   g++ -O1 -ffp-contract=off tests/cpp/exact_sum_test.cpp -o tests/cpp/bin/exact_sum_test && tests/cpp/bin/exact_sum_test
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../cozo_amd/csrc/exact_sum.h"

using namespace cz_exact;

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float seq_sum(const std::vector<float> &t, float s) {
    for (float v : t) { volatile float x = s + v; s = x; }
    return s;
}

static long g_passes = 0, g_pair_mismatch = 0;

static uint32_t g_mis = 0;  // (address of t / 4) mod 4 the emulation pretends

template <int T>
static float emu_wave_seq_sum(const float *t, uint32_t n, float s) {
    constexpr uint32_t PRO = 32;
    uint32_t p = PRO + ((4u - ((g_mis + PRO) & 3u)) & 3u);
    if (n < 2u * PRO + 4u) p = n;
    for (uint32_t j = 0; j < p; j++) { volatile float x = s + t[j]; s = x; }
    while (p < n) {
        g_passes++;
        const uint32_t rem = n - p;
        const uint32_t ext = std::min(std::min(rem, std::max(p, 64u)), 64u * T);  // as the device code
        const uint32_t per = (T >= 16 && ext > 64u * 8u) ? 16u : (T >= 8 && ext > 64u * 4u) ? 8u : 4u;
        const uint32_t used = (ext + per - 1u) / per;
        uint32_t a[64][T];
        Inc f[64], g[64], ex[64];
        uint32_t M, eb;
        split(f2u(s), M, eb);
        const bool s_ok = f2u(s) <= 0x7f7fffffu;  // a negative / inf / nan sum: true additions only (as the device code)
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t first = p + lane * per;
            for (uint32_t j = 0; j < (uint32_t)T; j++) {
                const uint32_t i = first + j;
                a[lane][j] = (lane < used && j < per && i < n) ? f2u(t[i]) : 0u;
            }
            Inc x = term_inc(a[lane][0], eb);
            for (int j = 1; j < T; j++) x = then(x, term_inc(a[lane][j], eb));
            {  // the device's lane-local form (add_term on both parities, clamped once) must be the same pair
                uint32_t e = 0, o = 0;
                for (int j = 0; j < T; j++) {
                    const Term c = classify(a[lane][j], eb);
                    e = add_term(e, c, 0u);
                    o = add_term(o, c, 1u);
                }
                e = e < kSat ? e : kSat;
                o = o < kSat ? o : kSat;
                const bool sat = x.even >= kLimit || x.odd >= kLimit;  // past the binade only "at or above kLimit" matters
                if (sat ? (e < kLimit || o < kLimit) : (e != x.even || o != x.odd)) g_pair_mismatch++;
            }
            if (!s_ok) x.even = x.odd = kSat;
            f[lane] = g[lane] = x;
        }
        for (int o = 1; o < 64; o <<= 1) {
            Inc nx[64];
            for (int lane = 0; lane < 64; lane++) nx[lane] = lane >= o ? then(g[lane - o], g[lane]) : g[lane];
            memcpy(g, nx, sizeof(g));
        }
        uint32_t m0[64], m1[64];
        int L = -1;
        for (int lane = 0; lane < 64; lane++) {
            if (lane == 0) ex[lane].even = ex[lane].odd = 0;
            else ex[lane] = g[lane - 1];
            m0[lane] = apply(M, ex[lane]);
            m1[lane] = apply(m0[lane], f[lane]);
            if (L < 0 && m1[lane] >= kLimit) L = lane;
        }
        if (L < 0) {
            s = u2f(join(m1[63], eb));
            p += used * per;
        } else {
            float sl = L == 0 ? s : u2f(join(m0[L] < kLimit ? m0[L] : 0u, eb));
            for (int j = 0; j < T; j++) { volatile float x = sl + u2f(a[L][j]); sl = x; }
            s = sl;
            p += (uint32_t)(L + 1) * per;
        }
    }
    return s;
}

static int failures = 0, cases = 0;
static bool same(float a, float b) { return (std::isnan(a) && std::isnan(b)) || f2u(a) == f2u(b); }

template <int T>
static void check(const char *what, const std::vector<float> &t, float s0 = 0.0f) {
    const float want = seq_sum(t, s0);
    const float got = emu_wave_seq_sum<T>(t.data(), (uint32_t)t.size(), s0);
    cases++;
    if (!same(want, got)) {
        failures++;
        if (failures < 20) printf("MISMATCH %s T=%d n=%zu: want %.9g (%08x) got %.9g (%08x)\n", what, T, t.size(), want, f2u(want), got, f2u(got));
    }
}

static void both(const char *what, const std::vector<float> &t, float s0 = 0.0f) {
    g_mis = (g_mis + 1) & 3;
    check<4>(what, t, s0);
    check<8>(what, t, s0);
    check<16>(what, t, s0);
}

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const uint32_t sizes[] = {0, 1, 2, 3, 63, 64, 65, 127, 128, 500, 512, 513, 1000, 1024, 4097, 16384, 100003};
    for (uint32_t n : sizes) {
        std::vector<float> t(n);
        // PageRank-like: score / out-degree around 1e-7
        for (auto &v : t) v = 1e-7f * (0.05f + U(rng)) / (float)(1 + (rng() % 40));
        both("pagerank-like", t);
        // wide exponent range
        for (auto &v : t) v = std::ldexp(U(rng), (int)(rng() % 60) - 40);
        both("wide", t);
        // ties: terms that are exact half-ulps / ulps of a sum near 1 and near 2^k
        for (auto &v : t) { const int k = (int)(rng() % 4); v = std::ldexp(1.0f, -24 - k) * (float)(1 + (rng() % 3)); }
        both("ties@1", t, 1.0f);
        both("ties@1+ulp", t, u2f(f2u(1.0f) + 1));
        both("ties@0.75", t, 0.75f);
        // sparse: mostly zeros (also -0.0), some denormals
        for (auto &v : t) { const int k = (int)(rng() % 8); v = k == 0 ? U(rng) : k == 1 ? u2f((uint32_t)(rng() % 5000)) : k == 2 ? -0.0f : 0.0f; }
        both("sparse+denormal", t);
        both("sparse+denormal from denormal", t, u2f(77));
        // all denormal: the sum walks from denormals into the normals
        for (auto &v : t) v = u2f((uint32_t)(rng() % 0x7fffff));
        both("denormal only", t);
        // big jumps
        for (auto &v : t) v = (rng() % 97 == 0) ? std::ldexp(U(rng), 30) : U(rng);
        both("jumps", t);
        // equal terms (ties against powers of two all the time)
        for (auto &v : t) v = 1.0f;
        both("ones", t);
        for (auto &v : t) v = 3.0f * std::ldexp(1.0f, -20);
        both("threes", t, 1.0f);
    }
    // outside the integer view: negative terms, inf, nan, overflow, a negative start
    {
        std::vector<float> t(3000);
        for (auto &v : t) v = U(rng) - 0.3f;
        both("negative terms", t);
        both("negative start", t, -5.0f);
        for (auto &v : t) v = U(rng);
        t[1234] = INFINITY;
        both("inf", t);
        t[2000] = -INFINITY;
        both("inf-inf", t);
        for (auto &v : t) v = 3e38f * U(rng);
        both("overflow", t);
        for (auto &v : t) v = U(rng);
        t[17] = NAN;
        both("nan", t);
        both("start inf", std::vector<float>(100, 1.0f), INFINITY);
    }
    // randomised lengths
    for (int it = 0; it < 400; it++) {
        const uint32_t n = (uint32_t)(rng() % 3000);
        std::vector<float> t(n);
        const int mode = (int)(rng() % 3);
        for (auto &v : t)
            v = mode == 0 ? U(rng) * 1e-8f : mode == 1 ? std::ldexp(1.0f + (float)(rng() % 4) * 0.25f, -(int)(rng() % 30)) : std::ldexp(U(rng), (int)(rng() % 20) - 10);
        both("random", t, (rng() & 1) ? 0.0f : U(rng));
    }
    // pass count on a long row: n / (64 T) + O(log n)
    {
        std::vector<float> t(131072);
        for (auto &v : t) v = 1e-7f * (0.05f + U(rng));
        g_passes = 0;
        check<16>("long row", t);
        printf("passes for 131072 terms at T=16: %ld (128 full passes + what the binade crossings and the growing pass sizes add)\n", g_passes);
        if (g_passes > 128 + 96) { failures++; printf("too many passes\n"); }
    }
    if (g_pair_mismatch) { failures++; printf("lane-local composition differs from then(): %ld times\n", g_pair_mismatch); }
    printf("%d cases, %d failed\n", cases, failures);
    return failures ? 1 : 0;
}
