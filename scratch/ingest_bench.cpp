// scratch: host ingest of a relation into the CSR the kernels take: first-appearance id assignment by
// hash map + two counting-sort passes (cozo_amd/host) against the reference's shape of the same step, a
// BTreeMap<DataValue, u32> lookup per endpoint and a comparison sort of the edge list (fixed_rule/mod.rs:144-195).
//   g++ -std=c++17 -O2 -Icozo_amd/host/include -Iinclude scratch/ingest_bench.cpp cozo_amd/host/src/fixed_rule.cpp \
//       cozo_amd/host/src/graph_rules.cpp -Lcozo_amd/lib -lcozo_gpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/cozo_amd/lib -o /tmp/ingest_bench
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <map>
#include <random>

#include "cozo_host/fixed_rule.hpp"

using namespace cozo;
using clk = std::chrono::steady_clock;
static double secs(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }

int main(int argc, char **argv) {
    const size_t E = argc > 1 ? (size_t)atoll(argv[1]) : 2000000;
    const uint32_t N = (uint32_t)(E / 10);
    for (int strings = 0; strings < 2; strings++) {
        std::mt19937_64 rng(1);
        std::vector<Tuple> rows;
        rows.reserve(E);
        for (size_t i = 0; i < E; i++) {
            const uint64_t a = rng() % N, b = rng() % N;
            if (strings) rows.push_back(Tuple{DataValue("node-" + std::to_string(a)), DataValue("node-" + std::to_string(b))});
            else rows.push_back(Tuple{DataValue((int64_t)a), DataValue((int64_t)b)});
        }
        auto t0 = clk::now();
        FixedRuleInputRelation rel(std::move(rows));  // the stored relation: sorted, de-duplicated (what a scan yields)
        const double t_scan = secs(t0);
        t0 = clk::now();
        GraphWithIndices g = rel.as_directed_graph(false);
        const double t_fast = secs(t0);
        // the reference's shape: ordered-map lookups + comparison sorts
        t0 = clk::now();
        std::map<DataValue, uint32_t> inv;
        std::vector<DataValue> indices;
        std::vector<std::pair<uint32_t, uint32_t>> el;
        el.reserve(rel.iter().size());
        for (const Tuple &t : rel.iter()) {
            uint32_t id[2];
            for (int c = 0; c < 2; c++) {
                auto it = inv.find(t[c]);
                if (it == inv.end()) {
                    it = inv.emplace(t[c], (uint32_t)indices.size()).first;
                    indices.push_back(t[c]);
                }
                id[c] = it->second;
            }
            el.push_back({id[0], id[1]});
        }
        std::vector<std::pair<uint32_t, uint32_t>> rev(el);
        for (auto &p : rev) std::swap(p.first, p.second);
        std::sort(el.begin(), el.end());
        std::sort(rev.begin(), rev.end());
        const double t_ref = secs(t0);
        bool same = indices.size() == g.indices.size();
        for (size_t e = 0; same && e < el.size(); e++) same = el[e].second == g.graph.out_targets[e] && rev[e].second == g.graph.in_sources[e];
        std::printf("%s keys: %zu rows, %zu nodes | scan+sort of the relation %.2f s | as_directed_graph: hash + counting sort %.3f s (%.1f M rows/s), "
                    "ordered map + comparison sort %.3f s (%.1f M rows/s), x%.1f, identical CSR: %s\n",
                    strings ? "string" : "int", rel.iter().size(), g.indices.size(), t_scan, t_fast, rel.iter().size() / t_fast / 1e6, t_ref,
                    rel.iter().size() / t_ref / 1e6, t_ref / t_fast, same ? "yes" : "NO");
    }
    return 0;
}
