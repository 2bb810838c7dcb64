// scratch: throughput of libcozo_ingest's czi_graph_ingest + czi_graph_csr on the stored bytes of an edge relation
// (SURVEY section 8 f1), one core.  Rows are (from, to) int keys or "node-<n>" string keys in key order.
//   g++ -std=c++17 -O2 -Iinclude scratch/ingest_bytes_bench.cpp -Lcozo_amd/lib -lcozo_ingest -Wl,-rpath,$PWD/cozo_amd/lib -o /tmp/ingest_bytes_bench
//   /tmp/ingest_bytes_bench 10000000
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "cozo_ingest.h"

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }

static void put_be64(std::string &o, uint64_t v) {
    for (int i = 7; i >= 0; i--) o.push_back((char)(v >> (8 * i)));
}
static void enc_int(std::string &o, int64_t i) {  // data/memcmp.rs:127-145, |i| < 2^53
    o.push_back(0x05);
    double f = (double)i;
    uint64_t u;
    memcpy(&u, &f, 8);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    put_be64(o, u);
    o.push_back(0x00);
}
static void enc_str(std::string &o, const std::string &s) {  // :147-163
    o.push_back(0x06);
    size_t index = 0;
    while (index <= s.size()) {
        const size_t remain = s.size() - index;
        if (remain > 8) {
            o.append(s, index, 8);
            o.push_back((char)0xFF);
        } else {
            o.append(s, index, remain);
            o.append(8 - remain, '\0');
            o.push_back((char)(0xFF - (8 - remain)));
        }
        index += 8;
    }
}

int main(int argc, char **argv) {
    const size_t E = argc > 1 ? (size_t)atoll(argv[1]) : 2000000;
    const uint64_t N = E / 10;
    for (int strings = 0; strings < 2; strings++) {
        std::mt19937_64 rng(1);
        std::vector<std::string> keys(E);
        for (size_t i = 0; i < E; i++) {
            const uint64_t a = rng() % N, b = rng() % N;
            std::string &k = keys[i];
            put_be64(k, 7);
            if (strings) {
                enc_str(k, "node-" + std::to_string(a));
                enc_str(k, "node-" + std::to_string(b));
            } else {
                enc_int(k, (int64_t)a);
                enc_int(k, (int64_t)b);
            }
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> off{0};
        for (const std::string &k : keys) {
            bytes.insert(bytes.end(), k.begin(), k.end());
            off.push_back(bytes.size());
        }
        czi_rows rel{bytes.data(), off.data(), nullptr, nullptr, keys.size(), 2};
        auto t0 = clk::now();
        czi_graph *g = nullptr;
        if (czi_graph_ingest(&rel, 0, &g)) {
            fprintf(stderr, "%s\n", czi_last_error());
            return 1;
        }
        const double t_ids = secs(t0);
        const uint32_t n = czi_graph_node_count(g);
        std::vector<uint32_t> ooff(n + 1), otgt(keys.size()), ioff(n + 1), isrc(keys.size());
        double t_csr = 1e30;
        for (int rep = 0; rep < 3; rep++) {  // this VM's memory system is noisy at GB footprints: best of three
            t0 = clk::now();
            czi_graph_csr(g, 0, ooff.data(), otgt.data(), nullptr);
            czi_graph_csr(g, 1, ioff.data(), isrc.data(), nullptr);
            t_csr = std::min(t_csr, secs(t0));
        }
        printf("%s keys: %zu rows (%.1f B/row), %u nodes: id assignment %.3f s (%.1f M rows/s), both CSR directions %.3f s, "
               "total %.1f M rows/s\n",
               strings ? "string" : "int", keys.size(), (double)bytes.size() / keys.size(), n, t_ids, keys.size() / t_ids / 1e6, t_csr,
               keys.size() / (t_ids + t_csr) / 1e6);
        czi_graph_free(g);
    }
    return 0;
}
