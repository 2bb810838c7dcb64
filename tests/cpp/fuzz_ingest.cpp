// fuzz_ingest.cpp -- TEST ONLY.  Robustness of libcozo_ingest's parsers: a valid set of stored rows (written by the test
// harness to a file) is mutated at random -- byte flips, truncations, offset-table damage -- and fed to the entry
// points.  Built together with cozo_amd/ingest/*.cpp under -fsanitize=address,undefined: every input must come back with CZI_OK or
// a negative status; a crash, an out-of-bounds read or an exception through the C ABI fails the test.
//   fuzz_ingest <graph.bin> <idx.bin> <base.bin> <iterations>
// file format: u32 n_key_cols, u64 n_rows, key_off[n_rows+1], val_off[n_rows+1], u64 key_bytes, keys, u64 val_bytes, vals
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "cozo_ingest.h"

struct Rows {
    uint32_t n_key_cols = 0;
    std::vector<uint64_t> key_off, val_off;
    std::vector<uint8_t> keys, vals;
    czi_rows view() const {
        return czi_rows{keys.data(), key_off.data(), vals.data(), val_off.data(), key_off.empty() ? 0 : key_off.size() - 1, n_key_cols};
    }
};

static Rows load(const char *path) {
    Rows r;
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        exit(2);
    }
    uint64_t n = 0, nb = 0;
    if (fread(&r.n_key_cols, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1) exit(2);
    r.key_off.resize(n + 1);
    r.val_off.resize(n + 1);
    if (fread(r.key_off.data(), 8, n + 1, f) != n + 1 || fread(r.val_off.data(), 8, n + 1, f) != n + 1) exit(2);
    if (fread(&nb, 8, 1, f) != 1) exit(2);
    r.keys.resize(nb);
    if (nb && fread(r.keys.data(), 1, nb, f) != nb) exit(2);
    if (fread(&nb, 8, 1, f) != 1) exit(2);
    r.vals.resize(nb);
    if (nb && fread(r.vals.data(), 1, nb, f) != nb) exit(2);
    fclose(f);
    return r;
}

// damage that keeps the offset tables inside their buffers (the caller's contract), but nothing else
static Rows mutate(const Rows &seed, std::mt19937_64 &rng) {
    Rows r = seed;
    const int kind = (int)(rng() % 6);
    auto flip = [&](std::vector<uint8_t> &b, int count) {
        for (int i = 0; i < count && !b.empty(); i++) b[rng() % b.size()] = (uint8_t)rng();
    };
    switch (kind) {
        case 0: flip(r.keys, 1 + (int)(rng() % 4)); break;
        case 1: flip(r.vals, 1 + (int)(rng() % 4)); break;
        case 2: flip(r.keys, 64); flip(r.vals, 64); break;
        case 3: {  // shift one row boundary (rows overlap / shrink)
            if (r.key_off.size() > 2) {
                const size_t i = 1 + rng() % (r.key_off.size() - 2);
                r.key_off[i] = rng() % (r.keys.size() + 1);
            }
            break;
        }
        case 4: {
            if (r.val_off.size() > 2) {
                const size_t i = 1 + rng() % (r.val_off.size() - 2);
                r.val_off[i] = rng() % (r.vals.size() + 1);
            }
            break;
        }
        default: r.n_key_cols = (uint32_t)(rng() % 12); break;
    }
    return r;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const Rows graph = load(argv[1]), idx = load(argv[2]), base = load(argv[3]);
    const int iters = atoi(argv[4]);
    std::mt19937_64 rng(12345);
    int ok = 0, rejected = 0;
    for (int it = 0; it < iters; it++) {
        {
            const Rows m = it ? mutate(graph, rng) : graph;
            const czi_rows v = m.view();
            czi_graph *g = nullptr;
            const int rc = czi_graph_ingest(&v, (uint32_t)(rng() % 16), &g);
            if (rc > 0) return 3;
            if (rc == CZI_OK) {
                const uint32_t n = czi_graph_node_count(g);
                std::vector<uint32_t> off(n + 1), tgt(czi_graph_edge_count(g));
                std::vector<float> w(tgt.size());
                if (czi_graph_csr(g, it & 1, off.data(), tgt.data(), w.data()) != CZI_OK) return 4;
                if (off[n] != tgt.size()) return 5;
                ok++;
            } else {
                if (!*czi_last_error()) return 6;
                rejected++;
            }
            czi_graph_free(g);
        }
        {
            const bool damage_base = it % 3 == 2;
            const Rows mi = (it && !damage_base) ? mutate(idx, rng) : idx;
            const Rows mb = (it && damage_base) ? mutate(base, rng) : base;
            const czi_rows vi = mi.view(), vb = mb.view();
            const uint32_t fields[2] = {2, 3};
            czi_hnsw *h = nullptr;
            const int rc = czi_hnsw_ingest(&vi, &vb, fields, 2, 8, 0, 6, 12, &h);
            if (rc > 0) return 7;
            if (rc == CZI_OK) {
                cz_hnsw_desc d;
                const float *vec = nullptr;
                if (czi_hnsw_desc(h, &d, &vec) != CZI_OK) return 8;
                for (int l = 0; l < d.n_levels; l++)  // every id the kernels would dereference is in range
                    for (uint64_t s = 0; s < (uint64_t)d.level_size[l] * d.level_width[l]; s++)
                        if (d.level_nbrs[l][s] != CZ_NONE && d.level_nbrs[l][s] >= d.n) return 9;
                if (d.n_levels > 0 && d.entry >= d.n) return 10;
                ok++;
            } else {
                if (!*czi_last_error()) return 11;
                rejected++;
            }
            czi_hnsw_free(h);
        }
    }
    printf("fuzz_ingest: %d inputs accepted, %d rejected, no fault\n", ok, rejected);
    return ok > 0 && rejected > 0 ? 0 : 12;
}
