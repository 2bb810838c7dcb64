// inplace_plan_test.cpp -- CPU walk of cozo_amd/csrc/inplace_plan.hpp's arrays, element for element what the kernels of
// csrc/pagerank_inplace.hip do with them (phase A: val[pos] = staged slice[asrc[pos]]; phase B: tile[perm[e]] = stream[vpos[e]],
// tile[upos[j]] = contribution[usrc[j]], then every row added in order), driven by the same schedule (phase A of level l any time
// between phase B of level l and phase B of level l + urgent_gap + 1).  TEST CODE: it exists so that the layout and the schedule
// can be checked against the oracle where there is no GPU (tests/test_inplace_plan.py); the product never runs it.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../../cozo_amd/csrc/inplace_plan.hpp"

namespace {

struct State {
    const czgs::Plan &p;
    std::vector<float> c[2], X, Y[2], scores;
    explicit State(const czgs::Plan &plan) : p(plan) {
        const float nan = std::numeric_limits<float>::quiet_NaN();
        c[0].assign(p.N, nan);
        c[1].assign(p.N, nan);
        X.assign(p.n_pos[0] + 4, nan);
        Y[0].assign(p.n_pos[1] + 4, nan);
        Y[1].assign(p.n_pos[1] + 4, nan);
        scores.assign(p.N, nan);
    }
    void expand(const czgs::Item &it, const float *contrib, float *out) const {
        // the staged slice: aligned 16-byte vectors, LDS word 0 = node (node0 & ~3); words outside the slice are whatever memory holds
        const uint32_t mis = it.node0 & 3u;
        std::vector<float> sl(mis + it.n + 4, std::numeric_limits<float>::quiet_NaN());
        for (uint32_t i = 0; i < it.n; i++) sl[mis + i] = contrib[it.node0 + i];
        for (uint32_t i = it.begin; i < it.end; i++) out[i] = sl.at(p.asrc[it.cls][i]);
    }
};

}  // namespace

// schedule: 0 = phase A of a level as EARLY as allowed (right after the level's phase B), 1 = as LATE as allowed (right before
// phase B of level l + urgent_gap + 1, the rest at the end of the sweep).  Both must give the oracle's scores.
// info: [0] levels, [1] blocks, [2] items, [3] long rows, [4] X edges, [5] Y edges, [6] urgent edges, [7] long edges
extern "C" int ipt_emulate(const uint64_t *in_off, const uint32_t *in_src, const uint32_t *out_deg, uint32_t N, uint32_t tile,
                           uint32_t rows_per_block, uint32_t slice, uint32_t part, uint32_t urgent_gap, float damping, uint32_t sweeps,
                           int schedule, float *scores_out, double *err_out, uint64_t *info) {
    czgs::Params prm;
    prm.tile = tile;
    prm.rows_per_block = rows_per_block;
    prm.slice = slice;
    prm.part = part;
    prm.urgent_gap = urgent_gap & 0xffffu;
    prm.jacobi = (urgent_gap >> 16) != 0;  // (bit 16 of the argument: the Jacobi reading in this layout)
    urgent_gap &= 0xffffu;
    czgs::Plan p;
    if (!czgs::build_plan(in_off, in_src, out_deg, N, prm, p)) return -1;
    if (info) {
        info[0] = p.L;
        info[1] = p.blocks.size();
        info[2] = p.items.size();
        info[3] = p.long_rows.size();
        info[4] = p.n_edges[0];
        info[5] = p.n_edges[1];
        info[6] = p.n_edges[2];
        info[7] = p.n_long_edges;
    }
    if (N == 0) return 0;
    // structural checks the kernels rely on
    for (const czgs::Item &it : p.items)
        if ((it.begin & 3u) || (it.end & 3u) || it.end <= it.begin || it.end - it.begin > prm.part || it.n > prm.slice || it.node0 + it.n > N) return -2;
    for (const czgs::Block &b : p.blocks)
        if (b.row1 <= b.row0 || b.row1 - b.row0 > prm.rows_per_block || p.off2[b.row1] - b.e0 > prm.tile) return -3;
    State st(p);
    const float init = 1.0f / (float)N, base = (1.0f - damping) / (float)N;
    for (uint32_t i = 0; i < N; i++) {
        st.scores[i] = init;
        st.c[1][i] = init / (float)p.od[i];
    }
    for (const czgs::Item &it : p.items)
        if (it.cls == 1) st.expand(it, st.c[1].data(), st.Y[0].data());
    const float nan = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> tilebuf;
    double err = 0.0;
    for (uint32_t t = 0; t < sweeps; t++) {
        float *cn = st.c[t & 1].data();
        const float *co = st.c[(t + 1) & 1].data();
        const float *Ycur = st.Y[t & 1].data();
        float *Ynext = st.Y[(t + 1) & 1].data();
        err = 0.0;
        uint32_t next_a = 0;  // phase A of levels [0, next_a) has run
        auto run_a = [&](uint32_t upto) {
            for (; next_a < upto; next_a++)
                for (uint32_t i = p.item_first[next_a]; i < p.item_first[next_a + 1]; i++)
                    st.expand(p.items[i], cn, p.items[i].cls ? Ynext : st.X.data());
        };
        for (uint32_t l = 0; l < p.L; l++) {
            if (schedule == 1 && l > urgent_gap) run_a(l - urgent_gap);  // levels <= l - urgent_gap - 1
            for (uint32_t bi = p.blk_first[l]; bi < p.blk_first[l + 1]; bi++) {
                const czgs::Block &b = p.blocks[bi];
                const uint32_t nt = p.off2[b.row1] - b.e0;
                tilebuf.assign((size_t)prm.tile + 4, nan);
                for (uint32_t g = b.g0; g < b.g1; g++) {
                    const uint32_t gp = p.gpos[g];
                    if ((gp & 3u) != 0) return -4;  // groups are read as one aligned 16-byte vector
                    const float *src = (gp & czgs::kYBit) ? Ycur + (gp & ~czgs::kYBit) : st.X.data() + gp;
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t q = p.gperm[4 * (size_t)g + k];
                        if (q > prm.tile || (q < prm.tile && q >= nt)) return -5;
                        tilebuf[q] = src[k];  // q == tile: padding, lands in the spare words
                    }
                }
                for (uint32_t j = b.u0; j < b.u1; j++) tilebuf.at(p.upos[j]) = cn[p.usrc[j]];
                for (uint32_t r = b.row0; r < b.row1; r++) {
                    float s = 0.0f;
                    for (uint32_t e = p.off2[r] - b.e0; e < p.off2[r + 1] - b.e0; e++) s = s + tilebuf[e];
                    const float old = st.scores[r];
                    const float nw = base + damping * s;
                    st.scores[r] = nw;
                    cn[r] = nw / (float)p.od[r];
                    err += std::fabs((double)(nw - old));
                }
            }
            for (uint32_t k = p.long_first[l]; k < p.long_first[l + 1]; k++) {
                const uint32_t r = p.long_rows[k];
                float s = 0.0f;
                for (uint32_t e = p.long_off[k]; e < p.long_off[k + 1]; e++) {
                    const uint32_t v = p.long_src[e];
                    s = s + ((v & czgs::kOldBit) ? co[v & ~czgs::kOldBit] : cn[v]);
                }
                const float old = st.scores[r];
                const float nw = base + damping * s;
                st.scores[r] = nw;
                cn[r] = nw / (float)p.od[r];
                err += std::fabs((double)(nw - old));
            }
            if (schedule == 0) run_a(l + 1);
        }
        run_a(p.L);
    }
    for (uint32_t i = 0; i < N; i++) scores_out[p.order[i]] = st.scores[i];
    if (err_out) *err_out = err;
    return 0;
}
