// test_host.cpp -- tests of the C++ host mirror (cozo_amd/host) through the C ABI, against the CPU oracle.
//
//   test_host cpu   host logic only: DataValue order, option readers, id assignment / CSR vs the oracle, registry
//   test_host gpu   the fixed rules and HnswSearchRA on a real MI355X, rows compared with the oracle's
//   test_host_shim rules-cpu   (linked against tests/cpp/oracle_shim.c) the same rule checks without a device
//   test_host gpu-stored   the rules off a stored relation's bytes; a GPU-built index written out as `tbl:idx` stored bytes
//                          and read back off them (device)
// The oracle (oracle/cozo_oracle.h) is test infrastructure; it is linked into this test binary only.
// Reads like the reference's own tests: runtime/tests.rs:529-577 (custom rule), algos/shortest_path_bfs.rs:124-174
// (love graph), runtime/tests.rs:178-207 (PageRank options), runtime/tests.rs:700-809 (vector search).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <set>
#include <vector>

#include "cozo_gpu.h"
#include "cozo_host/codec.hpp"
#include "cozo_host/fixed_rule.hpp"
#include "cozo_host/graph_rules.hpp"
#include "cozo_host/hnsw.hpp"
#include "cozo_oracle.h"

using namespace cozo;

static int g_fail = 0, g_pass = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (cond) g_pass++;                                                      \
        else {                                                                   \
            g_fail++;                                                            \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
        }                                                                        \
    } while (0)
template <class E, class F>
static bool throws(F f, const char *code = nullptr) {
    try {
        f();
    } catch (const E &e) {
        return !code || e.code == code;
    } catch (...) {
        return false;
    }
    return false;
}

static Tuple T(std::initializer_list<DataValue> l) { return Tuple(l); }

// ---------------------------------------------------------------------------------------------------------------
static void test_value_order() {
    // Null < Bool < Num < Str < Bytes < List; Int before the equal Float; floats by total order
    std::vector<DataValue> v = {DataValue::list({DataValue(1)}), DataValue("b"), DataValue(2.0), DataValue(2), DataValue(true),
                                DataValue(), DataValue("a"), DataValue(1.5), DataValue(false), DataValue(-3),
                                DataValue(Bytes{{1, 2}}), DataValue(std::nan(""))};
    std::sort(v.begin(), v.end());
    const char *want[] = {"null", "false", "true", "-3", "1.5", "2", "2.0", "nan", "\"a\"", "\"b\"", "bytes(2)", "[1]"};
    for (size_t i = 0; i < v.size(); i++) CHECK(v[i].to_string() == want[i]);
    CHECK(DataValue(1) != DataValue(1.0));
    CHECK(DataValue(1).hash() != DataValue(1.0).hash());
    int64_t i;
    CHECK(DataValue(3.0).get_int(&i) && i == 3);
    CHECK(!DataValue(3.5).get_int(&i));
}

static void test_options() {
    std::map<std::string, DataValue> o = {{"theta", DataValue(0.5)}, {"iterations", DataValue(7)}, {"bad_iter", DataValue(2.5)},
                                          {"neg", DataValue(-1)}, {"flag", DataValue(true)}, {"name", DataValue("x")},
                                          {"big", DataValue(1.5)}};
    FixedRulePayload p("PageRank", {}, o);
    CHECK(p.unit_interval_option("theta") == 0.5);
    CHECK(p.unit_interval_option("absent", 0.85) == 0.85);
    CHECK(p.pos_integer_option("iterations") == 7);
    CHECK(p.pos_integer_option("absent", 10) == 10);
    CHECK(p.bool_option("flag") == true);
    CHECK(p.string_option("name") == "x");
    CHECK(p.float_option("iterations") == 7.0);
    CHECK((throws<FixedRuleOptionNotFoundError>([&] { p.bool_option("absent"); }, "fixed_rule::arg_not_found")));
    CHECK((throws<FixedRuleOptionNotFoundError>([&] { p.integer_option("bad_iter"); })));  // sic, see fixed_rule.cpp
    CHECK((throws<WrongFixedRuleOptionError>([&] { p.pos_integer_option("neg"); }, "fixed_rule::arg_wrong")));
    CHECK((throws<WrongFixedRuleOptionError>([&] { p.unit_interval_option("big"); })));
    CHECK((throws<WrongFixedRuleOptionError>([&] { p.integer_option("name"); })));
    CHECK((throws<WrongFixedRuleOptionError>([&] { p.bool_option("theta"); })));
    CHECK((throws<WrongFixedRuleOptionError>([&] { p.string_option("flag"); })));
    CHECK((throws<FixedRuleInputNotFoundError>([&] { p.get_input(0); }, "fixed_rule::not_enough_args")));
}

static std::vector<Tuple> random_edges(uint32_t n, size_t e, uint64_t seed, bool weighted, std::vector<int64_t> *from = nullptr,
                                       std::vector<int64_t> *to = nullptr) {
    std::mt19937_64 rng(seed);
    std::vector<Tuple> rows;
    for (size_t i = 0; i < e; i++) {
        int64_t a = (int64_t)(rng() % n) * 7 - 100, b = (int64_t)(rng() % n) * 7 - 100;
        if (weighted) rows.push_back(T({DataValue(a), DataValue(b), DataValue((double)(rng() % 1000) / 8.0)}));
        else rows.push_back(T({DataValue(a), DataValue(b)}));
    }
    FixedRuleInputRelation rel(rows);
    if (from) {
        from->clear();
        to->clear();
        for (const Tuple &t : rel.iter()) {
            int64_t a, b;
            t[0].get_int(&a);
            t[1].get_int(&b);
            from->push_back(a);
            to->push_back(b);
        }
    }
    return rel.iter();
}

static void test_as_directed_graph_vs_oracle() {
    for (int undirected = 0; undirected < 2; undirected++) {
        std::vector<int64_t> from, to;
        std::vector<Tuple> rows = random_edges(500, 4000, 11 + undirected, false, &from, &to);
        FixedRuleInputRelation rel(rows);
        GraphWithIndices g = rel.as_directed_graph(undirected != 0);
        const uint64_t E = from.size();
        std::vector<uint32_t> fi(E), ti(E);
        std::vector<int64_t> ind(2 * E);
        const uint32_t n = orc_assign_ids(from.data(), to.data(), E, fi.data(), ti.data(), ind.data());
        CHECK(n == g.graph.n);
        bool same = n == g.indices.size();
        for (uint32_t i = 0; same && i < n; i++) {
            int64_t v = 0;
            same = g.indices[i].get_int(&v) && v == ind[i] && g.inv_indices.at(g.indices[i]) == i;
        }
        CHECK(same);
        const uint64_t E2 = undirected ? 2 * E : E;
        std::vector<uint64_t> off(n + 1);
        std::vector<uint32_t> tgt(E2);
        orc_build_csr(n, E, fi.data(), ti.data(), nullptr, undirected, off.data(), tgt.data(), nullptr);
        bool ok = g.graph.out_targets.size() == E2;
        for (uint32_t v = 0; ok && v <= n; v++) ok = g.graph.out_offsets[v] == off[v];
        for (uint64_t e = 0; ok && e < E2; e++) ok = g.graph.out_targets[e] == tgt[e];
        CHECK(ok);
        // in-adjacency = CSR of the reversed rows
        orc_build_csr(n, E, ti.data(), fi.data(), nullptr, undirected, off.data(), tgt.data(), nullptr);
        ok = true;
        for (uint32_t v = 0; ok && v <= n; v++) ok = g.graph.in_offsets[v] == off[v];
        for (uint64_t e = 0; ok && e < E2; e++) ok = g.graph.in_sources[e] == tgt[e];
        CHECK(ok);
    }
    // weighted: default weight 1.0, bad weights rejected
    {
        FixedRuleInputRelation rel({T({DataValue("a"), DataValue("b"), DataValue(2.5)}), T({DataValue("b"), DataValue("c")})});
        GraphWithIndices g = rel.as_directed_weighted_graph(false, false);
        CHECK(g.graph.n == 3 && g.graph.out_weights.size() == 2 && g.graph.out_weights[0] == 2.5f && g.graph.out_weights[1] == 1.0f);
        FixedRuleInputRelation bad1({T({DataValue("a"), DataValue("b"), DataValue(-1.0)})});
        CHECK((throws<BadEdgeWeightError>([&] { bad1.as_directed_weighted_graph(false, false); }, "algo::invalid_edge_weight")));
        CHECK(bad1.as_directed_weighted_graph(false, true).graph.out_weights[0] == -1.0f);
        FixedRuleInputRelation bad2({T({DataValue("a"), DataValue("b"), DataValue("w")})});
        CHECK((throws<BadEdgeWeightError>([&] { bad2.as_directed_weighted_graph(false, false); })));
        FixedRuleInputRelation bad3({T({DataValue("a"), DataValue("b"), DataValue(INFINITY)})});
        CHECK((throws<BadEdgeWeightError>([&] { bad3.as_directed_weighted_graph(false, false); })));
        FixedRuleInputRelation bad4({T({DataValue("a")})});
        CHECK((throws<NotAnEdgeError>([&] { bad4.as_directed_graph(false); }, "algo::not_an_edge")));
    }
}

static void test_registry_and_simple_rule() {
    // runtime/tests.rs:529-577: a custom rule summing a column, scaled by an option
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    auto rule = std::make_shared<SimpleFixedRule>(1, [](const std::vector<NamedRows> &inputs, const std::map<std::string, DataValue> &opts) {
        int64_t mul = 1;
        opts.at("mul").get_int(&mul);
        NamedRows out;
        for (const Tuple &t : inputs[0].rows) {
            int64_t v = 0;
            t[0].get_int(&v);
            out.rows.push_back(T({DataValue(v * mul)}));
        }
        return out;
    });
    reg.register_fixed_rule("SumCols", rule);
    CHECK((throws<CozoError>([&] { reg.register_fixed_rule("SumCols", rule); })));
    CHECK((throws<CozoError>([&] { reg.register_fixed_rule("PageRank", rule); })));
    CHECK((throws<CozoError>([&] { reg.unregister_fixed_rule("PageRank"); })));
    FixedRulePayload p("SumCols", {FixedRuleInputRelation({T({DataValue(10)}), T({DataValue(26)})}, {"a"})}, {{"mul", DataValue(100)}});
    RegularTempStore out = reg.run("SumCols", p, Poison(), {"x"});
    std::vector<Tuple> rows = out.rows();
    int64_t a = 0, b = 0;
    CHECK(rows.size() == 2 && rows[0][0].get_int(&a) && rows[1][0].get_int(&b) && a == 1000 && b == 2600);
    CHECK((throws<CozoError>([&] { reg.run("SumCols", p, Poison(), {"x", "y"}); }, "parser::fixed_rule_head_arity_mismatch")));
    CHECK(reg.unregister_fixed_rule("SumCols"));
    CHECK(!reg.unregister_fixed_rule("SumCols"));
    CHECK((throws<CozoError>([&] { reg.get("SumCols"); }, "parser::fixed_rule_not_found")));
    // no device -> the GPU rules fail loudly (no CPU fallback); a killed query reports ProcessKilled
    Poison dead;
    dead.kill();
    CHECK((throws<ProcessKilled>([&] { dead.check(); }, "eval::killed")));
}

static void test_degree_centrality() {
    // degree_centrality.rs:24-76: (node, total, out, in); nodes of the optional second relation get zeros
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    FixedRuleInputRelation edges({T({DataValue("a"), DataValue("b")}), T({DataValue("a"), DataValue("c")}), T({DataValue("c"), DataValue("a")})});
    FixedRuleInputRelation nodes({T({DataValue("z")}), T({DataValue("a")})});
    RegularTempStore out = reg.run("DegreeCentrality", FixedRulePayload("DegreeCentrality", {edges, nodes}), Poison(), {"n", "t", "o", "i"});
    CHECK(out.size() == 4);
    CHECK(out.exists(T({DataValue("a"), DataValue(3), DataValue(2), DataValue(1)})));
    CHECK(out.exists(T({DataValue("b"), DataValue(1), DataValue(0), DataValue(1)})));
    CHECK(out.exists(T({DataValue("c"), DataValue(2), DataValue(1), DataValue(1)})));
    CHECK(out.exists(T({DataValue("z"), DataValue(0), DataValue(0), DataValue(0)})));
    FixedRuleInputRelation narrow({T({DataValue("a")})});
    CHECK((throws<InputRelationArityError>([&] { reg.run("DegreeCentrality", FixedRulePayload("DegreeCentrality", {narrow}), Poison()); })));
}

// ---- stored-row formats (cozo_host/codec.hpp); the property tests restate data/tests/memcmp.rs ---------------------
static std::vector<uint8_t> enc(const DataValue &v) { return memcmp_bytes(v); }
static DataValue dec(const std::vector<uint8_t> &b) {
    const uint8_t *p = b.data();
    DataValue v = decode_datavalue(p, b.data() + b.size());
    CHECK(p == b.data() + b.size());
    return v;
}

static void test_codec_numbers_and_order() {
    // data/tests/memcmp.rs:15-51: ints around every power of two, infinities, random floats and their reciprocals
    std::vector<DataValue> nums;
    const int64_t n = INT64_MAX;
    for (int i = 0; i < 54; i++)
        for (int j = 0; j < 1000; j += 41) {
            const int64_t vb = (n >> i) - j;
            nums.push_back(DataValue(vb));
            nums.push_back(DataValue(-vb - 1));
        }
    nums.push_back(DataValue(INFINITY));
    nums.push_back(DataValue(-INFINITY));
    std::mt19937_64 rng(3);
    for (int i = 0; i < 4000; i++) {
        const double f = ((double)(rng() >> 11) / 9007199254740992.0 - 0.5) * 2.0;
        nums.push_back(DataValue(f));
        nums.push_back(DataValue(1.0 / f));
    }
    for (int64_t i : {(int64_t)0, (int64_t)1, (int64_t)-1, (int64_t)1 << 53, ((int64_t)1 << 53) + 1, -((int64_t)1 << 53)}) nums.push_back(DataValue(i));
    for (double f : {0.0, -0.0, 1.0, -1.0, 9007199254740992.0}) nums.push_back(DataValue(f));
    bool round_trip = true;
    std::vector<std::pair<std::vector<uint8_t>, DataValue>> encoded;
    for (const DataValue &x : nums) {
        const std::vector<uint8_t> b = enc(x);
        const uint8_t *p = b.data();
        const DataValue y = decode_datavalue(p, b.data() + b.size());
        round_trip &= p == b.data() + b.size() && y.is_int() == x.is_int() && DataValue::compare(x, y) == 0;
        encoded.push_back({b, x});
    }
    CHECK(round_trip);
    // sorting the encodings sorts the numbers (Num::cmp: an Int before the Float it equals, floats by total_cmp)
    std::sort(encoded.begin(), encoded.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    bool ordered = true;
    for (size_t i = 1; i < encoded.size(); i++) ordered &= DataValue::compare(encoded[i - 1].second, encoded[i].second) <= 0;
    CHECK(ordered);
    CHECK(enc(DataValue((int64_t)1)) < enc(DataValue(1.0)) && enc(DataValue(1.0)) < enc(DataValue((int64_t)2)));
    CHECK(enc(DataValue(((int64_t)1 << 53) - 1)).size() == 10 && enc(DataValue((int64_t)1 << 53)).size() == 18);  // memcmp.rs:130-140
    double nan_back;
    CHECK(dec(enc(DataValue((double)NAN))).get_float(&nan_back) && std::isnan(nan_back));
}

static void test_codec_bytes_and_values() {
    // data/tests/memcmp.rs:65-98
    const std::string target = "Lorem ipsum dolor sit amet, consectetur adipiscing elit...";
    bool ok = true;
    for (size_t i = 0; i < target.size(); i++) {
        const std::string bs = target.substr(i);
        std::vector<uint8_t> e;
        for (const std::string *part : {&target, &bs, &bs, &target}) encode_bytes(e, (const uint8_t *)part->data(), part->size());
        const uint8_t *p = e.data();
        for (const std::string *part : {&target, &bs, &bs, &target}) {
            const std::vector<uint8_t> d = decode_bytes(p, e.data() + e.size());
            ok &= std::string(d.begin(), d.end()) == *part;
        }
        ok &= p == e.data() + e.size();
    }
    CHECK(ok);
    // :100-113 and :115-137
    std::vector<uint8_t> two = enc(DataValue((int64_t)2095));
    encode_datavalue(two, DataValue("MSS"));
    const uint8_t *p = two.data();
    CHECK(decode_datavalue(p, two.data() + two.size()) == DataValue((int64_t)2095));
    CHECK(decode_datavalue(p, two.data() + two.size()) == DataValue("MSS") && p == two.data() + two.size());
    std::vector<DataValue> dv = {DataValue(), DataValue(false), DataValue(true), DataValue((int64_t)1), DataValue(1.0), DataValue(INT64_MAX),
                                 DataValue(INT64_MAX - 1), DataValue(INT64_MAX - 2), DataValue(INT64_MIN), DataValue(INT64_MIN + 1),
                                 DataValue(INT64_MIN + 2), DataValue(INFINITY), DataValue(-INFINITY), DataValue::list({})};
    dv.push_back(DataValue::list(dv));
    dv.push_back(DataValue::list(dv));
    const DataValue nested = DataValue::list(dv);
    CHECK(dec(enc(nested)) == nested);
    // the store's row order (key bytes) is the evaluator's tuple order
    std::vector<DataValue> vals = {DataValue(), DataValue(false), DataValue(true), DataValue((int64_t)-5), DataValue(-5.0), DataValue((int64_t)0),
                                   DataValue(-0.0), DataValue(0.5), DataValue((int64_t)3), DataValue(3.0), DataValue(""), DataValue("a"),
                                   DataValue("ab"), DataValue("abcdefgh"), DataValue("abcdefghi"), DataValue("b"), DataValue(Bytes{{}}),
                                   DataValue(Bytes{{0}}), DataValue(Bytes{{0, 0}}), DataValue::list({}), DataValue::list({DataValue((int64_t)1)}),
                                   DataValue::list({DataValue((int64_t)1), DataValue("a")}), DataValue::list({DataValue::list({})})};
    bool same_order = true;
    for (const DataValue &a : vals)
        for (const DataValue &b : vals) {
            const int c = DataValue::compare(a, b);
            const std::vector<uint8_t> ea = enc(a), eb = enc(b);
            same_order &= (c < 0) == (ea < eb) && (c == 0) == (ea == eb);
        }
    CHECK(same_order);
    // ... except for Vec: VEC_TAG = 0x04 sorts a vector before every number in the store (memcmp.rs:25-26) while the enum
    // declares Vec after Set (value.rs:146-170) -- a stored relation and an in-memory one order vector columns differently
    CHECK(enc(DataValue(F32Vec{{1.5f}})) < enc(DataValue((int64_t)-5)) && DataValue::compare(DataValue(F32Vec{{1.5f}}), DataValue((int64_t)-5)) > 0);
    CHECK(enc(DataValue(F32Vec{{9.0f}})) < enc(DataValue(F32Vec{{1.0f, 1.0f}})) &&
          DataValue::compare(DataValue(F32Vec{{9.0f}}), DataValue(F32Vec{{1.0f, 1.0f}})) < 0);  // both: length first
    // a row through the store and back: key columns memcmp, value columns msgpack
    const Tuple row = T({DataValue((int64_t)9), DataValue("k"), DataValue(F32Vec{{1.5f, -2.0f, 0.25f}}),
                         DataValue::list({DataValue(F32Vec{{1.0f}}), DataValue("x")}), DataValue(2.5), DataValue(), DataValue(true),
                         DataValue(Bytes{{1, 2, 3}}), DataValue((int64_t)-70000), DataValue((int64_t)1 << 40)});
    for (uint32_t kcols : {0u, 2u, 10u}) {
        const std::vector<uint8_t> k = encode_key_for_store(3, row, kcols), v = encode_val_for_store(3, row, kcols);
        CHECK(k[7] == 3 && v[7] == 3);
        CHECK(decode_tuple_from_kv(k.data(), k.size(), v.data(), v.size()) == row);
    }
    const std::vector<uint8_t> v = encode_val_for_store(3, T({DataValue((int64_t)5), DataValue()}), 0);
    const uint8_t want[] = {0, 0, 0, 0, 0, 0, 0, 3, 0x92, 0x81, 0xa3, 'N', 'u', 'm', 0x81, 0xa3, 'I', 'n', 't', 5, 0xa4, 'N', 'u', 'l', 'l'};
    CHECK(v == std::vector<uint8_t>(want, want + sizeof want));  // [{"Num": {"Int": 5}}, "Null"]
}

static std::string hex(const std::vector<uint8_t> &b) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (uint8_t x : b) {
        s.push_back(d[x >> 4]);
        s.push_back(d[x & 15]);
    }
    return s;
}

static void test_codec_agrees_with_the_python_codec() {
    // tests/golden/stored_rows.json (written by cozo_amd/codec.py): the C++ encoders produce the same bytes
    const Tuple ints = T({DataValue((int64_t)2095), DataValue((int64_t)-1), DataValue((int64_t)0), DataValue((int64_t)1 << 53), DataValue(INT64_MIN)});
    CHECK(hex(encode_key_for_store(9, ints, 5)) ==
          "000000000000000905c0a05e00000000000005400fffffffffffff000580000000000000000005c340000000000000048020000000000000053c1fffffffffffff040000000000000000");
    CHECK(hex(encode_val_for_store(9, ints, 5)) == "000000000000000990");
    const Tuple strs = T({DataValue("MSS"), DataValue(""), DataValue("abcdefgh"), DataValue("abcdefghi"), DataValue("\xc3\xbc")});
    CHECK(hex(encode_key_for_store(9, strs, 5)) ==
          "0000000000000009064d53530000000000fa060000000000000000f7066162636465666768ff0000000000000000f7066162636465666768ff6900000000000000f806c3bc000000000000f9");
    const Tuple mixed = T({DataValue((int64_t)7), DataValue("k"), DataValue(), DataValue(true), DataValue(false), DataValue(Bytes{{0, 1}}),
                           DataValue::list({DataValue((int64_t)1), DataValue("x"), DataValue::list({DataValue(2.5)})}), DataValue(3.25),
                           DataValue("value"), DataValue((int64_t)-70000), DataValue((int64_t)1 << 40)});
    CHECK(hex(encode_key_for_store(9, mixed, 2)) == "000000000000000905c01c00000000000000066b00000000000000f8");
    CHECK(hex(encode_val_for_store(9, mixed, 2)) ==
          "000000000000000999a44e756c6c81a4426f6f6cc381a4426f6f6cc281a54279746573c402000181a44c6973749381a34e756d81a3496e740181a3537472a17881a44c"
          "6973749181a34e756d81a5466c6f6174cb400400000000000081a34e756d81a5466c6f6174cb400a00000000000081a3537472a576616c756581a34e756d81a3496e74d2"
          "fffeee9081a34e756d81a3496e74cf0000010000000000");
}

static void test_codec_rejects_damaged_bytes() {
    // decode_tuple_from_kv on mutated stored bytes: a tuple or a CodecError, never a fault (bounds are checked everywhere)
    const Tuple row = T({DataValue((int64_t)9), DataValue("key"), DataValue(F32Vec{{1.5f, -2.0f, 0.25f}}),
                         DataValue::list({DataValue("x"), DataValue((int64_t)1 << 40), DataValue(Bytes{{1, 2, 3}})}), DataValue(2.5), DataValue()});
    const std::vector<uint8_t> k = encode_key_for_store(3, row, 2), v = encode_val_for_store(3, row, 2);
    std::mt19937_64 rng(99);
    size_t decoded = 0, rejected = 0;
    for (int it = 0; it < 4000; it++) {
        std::vector<uint8_t> kk = k, vv = v;
        std::vector<uint8_t> &victim = (it & 1) ? kk : vv;
        const int kind = (int)(rng() % 3);
        if (kind == 0) victim[rng() % victim.size()] = (uint8_t)rng();
        else if (kind == 1) victim.resize(rng() % victim.size());
        else for (int j = 0; j < 8; j++) victim[rng() % victim.size()] = (uint8_t)rng();
        try {
            (void)decode_tuple_from_kv(kk.data(), kk.size(), vv.data(), vv.size());
            decoded++;
        } catch (const CodecError &) {
            rejected++;
        } catch (const std::length_error &) {  // an absurd length from a damaged header, refused by the allocator
            rejected++;
        } catch (const std::bad_alloc &) {
            rejected++;
        }
    }
    CHECK(decoded > 100 && rejected > 1000);
}

static bool same_graph(const GraphWithIndices &a, const GraphWithIndices &b) {
    return a.graph.n == b.graph.n && a.graph.out_offsets == b.graph.out_offsets && a.graph.out_targets == b.graph.out_targets &&
           a.graph.in_offsets == b.graph.in_offsets && a.graph.in_sources == b.graph.in_sources &&
           a.graph.out_weights == b.graph.out_weights && a.indices == b.indices && a.inv_indices.size() == b.inv_indices.size();
}

static std::vector<Tuple> mixed_rows(uint64_t seed, size_t n_rows) {
    const std::vector<DataValue> pool = {DataValue(), DataValue(true), DataValue((int64_t)0), DataValue((int64_t)1), DataValue(1.0),
                                         DataValue(-0.0), DataValue((int64_t)1 << 60), DataValue("a"), DataValue("node-17"),
                                         DataValue("a much longer string key than one group"), DataValue(Bytes{{0, 1}}),
                                         DataValue::list({DataValue((int64_t)1), DataValue("x")}), DataValue::list({}), DataValue((int64_t)7),
                                         DataValue(7.0), DataValue(2.5)};
    std::mt19937_64 rng(seed);
    std::vector<Tuple> rows;
    for (size_t i = 0; i < n_rows; i++) {
        const DataValue w = (rng() & 1) ? DataValue((double)(rng() % 200) / 4) : DataValue((int64_t)(rng() % 50));
        rows.push_back(T({pool[rng() % pool.size()], pool[rng() % pool.size()], w}));
    }
    return rows;
}

static void test_stored_relation_graphs() {
    // FixedRuleInputRelation::from_stored (bytes -> libcozo_ingest) against the tuple route, endpoints of every modelled
    // kind, in the key part or the value part of the row
    for (uint32_t kcols : {3u, 2u, 1u, 0u}) {
        const StoredRows stored = StoredRows::from_tuples(9, mixed_rows(5, 300), kcols);
        std::vector<Tuple> decoded;
        for (size_t i = 0; i < stored.size(); i++) decoded.push_back(stored.tuple(i));
        const FixedRuleInputRelation plain(decoded), fast = FixedRuleInputRelation::from_stored(stored);
        CHECK(fast.is_stored() && fast.arity() == 3 && fast.iter() == plain.iter());
        for (bool undirected : {false, true}) {
            CHECK(same_graph(fast.as_directed_graph(undirected), plain.as_directed_graph(undirected)));
            CHECK(same_graph(fast.as_directed_weighted_graph(undirected, false), plain.as_directed_weighted_graph(undirected, false)));
        }
        CHECK(same_graph(fast.as_ordered_graph({}), plain.as_ordered_graph({})));
        const std::vector<DataValue> extra = {DataValue("nobody"), decoded[0][0]};
        CHECK(same_graph(fast.as_ordered_graph(extra), plain.as_ordered_graph(extra)));
    }
    // an integer-keyed relation: the tuple route takes its flat i64 table here, the byte route does not care
    const std::vector<Tuple> ints = random_edges(500, 4000, 21, true);
    const FixedRuleInputRelation plain(ints), fast = FixedRuleInputRelation::from_stored(StoredRows::from_tuples(1, ints, 3));
    CHECK(fast.iter() == plain.iter());
    CHECK(same_graph(fast.as_directed_graph(false), plain.as_directed_graph(false)));
    CHECK(same_graph(fast.as_directed_weighted_graph(true, false), plain.as_directed_weighted_graph(true, false)));
    // errors of the reference, by the same types
    const FixedRuleInputRelation one_col = FixedRuleInputRelation::from_stored(StoredRows::from_tuples(1, {T({DataValue((int64_t)1)})}, 1));
    CHECK((throws<NotAnEdgeError>([&] { one_col.as_directed_graph(false); })));
    const FixedRuleInputRelation bad = FixedRuleInputRelation::from_stored(
        StoredRows::from_tuples(1, {T({DataValue((int64_t)1), DataValue((int64_t)2), DataValue("heavy")})}, 2));
    CHECK((throws<BadEdgeWeightError>([&] { bad.as_directed_weighted_graph(false, false); })));
    const FixedRuleInputRelation neg = FixedRuleInputRelation::from_stored(
        StoredRows::from_tuples(1, {T({DataValue((int64_t)1), DataValue((int64_t)2), DataValue(-1.5)})}, 2));
    CHECK((throws<BadEdgeWeightError>([&] { neg.as_directed_weighted_graph(false, false); })));
    CHECK(neg.as_directed_weighted_graph(false, true).graph.out_weights == std::vector<float>{-1.5f});
}

static void test_no_device_fails_loudly() {
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    FixedRulePayload p("PageRank", {FixedRuleInputRelation({T({DataValue(1), DataValue(2)})})});
    if (cz_device_count() <= 0) {
        CHECK((throws<GpuError>([&] { reg.run("PageRank", p, Poison()); })));
    } else {
        CHECK(reg.run("PageRank", p, Poison()).size() == 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GPU section
static std::vector<uint64_t> to_u64(const std::vector<uint32_t> &v) { return std::vector<uint64_t>(v.begin(), v.end()); }

static void gpu_pagerank() {
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    for (int undirected = 0; undirected < 2; undirected++) {
        std::vector<Tuple> rows = random_edges(3000, 30000, 5 + undirected, false);
        FixedRuleInputRelation rel(rows);
        std::map<std::string, DataValue> opts = {{"undirected", DataValue(undirected != 0)}, {"theta", DataValue(0.8)},
                                                 {"epsilon", DataValue(1e-6)}, {"iterations", DataValue(15)}};
        RegularTempStore out = reg.run("PageRank", FixedRulePayload("PageRank", {rel}, opts), Poison(), {"node", "rank"});
        GraphWithIndices g = rel.as_directed_graph(undirected != 0);
        std::vector<uint32_t> od = g.graph.out_degrees();
        std::vector<float> sc(g.graph.n);
        uint32_t it;
        double err;
        std::vector<uint64_t> off = to_u64(g.graph.in_offsets);
        orc_pagerank(g.graph.n, off.data(), g.graph.in_sources.data(), od.data(), 0.8f, (double)(float)1e-6, 15, sc.data(), &it, &err, 1);
        CHECK(out.size() == g.graph.n);
        bool same = true;
        for (const Tuple &t : out) {
            double s;
            same = same && t.size() == 2 && t[1].get_float(&s) && s == (double)sc[g.inv_indices.at(t[0])];
        }
        CHECK(same);  // bit-identical f32 scores, emitted as f64
    }
    // option validation happens before any device work, as in the reference (runtime/tests.rs:178-207)
    FixedRuleInputRelation rel({T({DataValue(1), DataValue(2)})});
    CHECK((throws<WrongFixedRuleOptionError>([&] { reg.run("PageRank", FixedRulePayload("PageRank", {rel}, {{"theta", DataValue(1.5)}}), Poison()); })));
    CHECK((throws<WrongFixedRuleOptionError>([&] { reg.run("PageRank", FixedRulePayload("PageRank", {rel}, {{"iterations", DataValue(0)}}), Poison()); })));
    CHECK(reg.run("PageRank", FixedRulePayload("PageRank", {FixedRuleInputRelation()}), Poison()).empty());  // :43-45
    Poison dead;
    dead.kill();
    CHECK((throws<ProcessKilled>([&] { reg.run("PageRank", FixedRulePayload("PageRank", {rel}), dead); })));
}

static void gpu_love_graph() {
    // algos/shortest_path_bfs.rs:124-174
    const char *e[][2] = {{"alice", "eve"}, {"bob", "alice"}, {"eve", "alice"}, {"eve", "bob"}, {"eve", "charlie"},
                          {"charlie", "eve"}, {"david", "george"}, {"george", "george"}};
    std::vector<Tuple> rows;
    for (auto &p : e) rows.push_back(T({DataValue(p[0]), DataValue(p[1])}));
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    FixedRuleInputRelation love(rows, {"loving", "loved"});
    FixedRuleInputRelation start({T({DataValue("alice")})}), end1({T({DataValue("bob")})}), end2({T({DataValue("george")})});
    RegularTempStore r1 = reg.run("ShortestPathBFS", FixedRulePayload("ShortestPathBFS", {love, start, end1}), Poison());
    CHECK(r1.size() == 1 && r1.rows()[0][2].get_slice() && r1.rows()[0][2].get_slice()->size() == 3);
    RegularTempStore r2 = reg.run("ShortestPathBFS", FixedRulePayload("ShortestPathBFS", {love, start, end2}), Poison());
    CHECK(r2.size() == 1 && r2.rows()[0][2].is_null());
    // ConnectedComponents: {alice, bob, eve, charlie} and {david, george}; an extra node from input 1 gets a fresh id
    FixedRuleInputRelation nodes({T({DataValue("zed")}), T({DataValue("alice")})});
    RegularTempStore cc = reg.run("ConnectedComponents", FixedRulePayload("ConnectedComponents", {love, nodes}), Poison());
    std::map<std::string, int64_t> grp;
    for (const Tuple &t : cc) {
        int64_t gid = 0;
        t[1].get_int(&gid);
        grp[*t[0].get_str()] = gid;
    }
    CHECK(grp.size() == 7 && grp["alice"] == grp["bob"] && grp["alice"] == grp["charlie"] && grp["david"] == grp["george"] &&
          grp["alice"] != grp["david"] && grp["zed"] == 2);
    CHECK((throws<GpuError>([&] { reg.run("SCC", FixedRulePayload("SCC", {love}), Poison()); })));
}

static void gpu_bfs_cc_dijkstra_random() {
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::vector<Tuple> rows = random_edges(2000, 7000, 99, false);
    FixedRuleInputRelation rel(rows);
    // --- ShortestPathBFS vs the oracle on the key-ordered graph
    std::vector<Tuple> st = {T({rows[0][0]}), T({rows[17][0]}), T({rows[400][1]})};
    std::vector<Tuple> en;
    for (int i = 0; i < 40; i++) en.push_back(T({rows[(size_t)i * 53 % rows.size()][1]}));
    FixedRuleInputRelation starts(st), ends(en);
    RegularTempStore sp = reg.run("ShortestPathBFS", FixedRulePayload("ShortestPathBFS", {rel, starts, ends}), Poison());
    std::vector<DataValue> extra;
    for (const Tuple &t : starts.iter()) extra.push_back(t[0]);
    for (const Tuple &t : ends.iter()) extra.push_back(t[0]);
    GraphWithIndices og = rel.as_ordered_graph(extra);
    std::vector<uint64_t> off = to_u64(og.graph.out_offsets);
    size_t checked = 0;
    bool same = true;
    for (const Tuple &s : starts.iter()) {
        std::vector<uint32_t> goals;
        for (const Tuple &t : ends.iter()) goals.push_back(og.inv_indices.at(t[0]));
        std::vector<uint32_t> par(og.graph.n);
        const uint32_t s_id = og.inv_indices.at(s[0]);
        orc_shortest_path_bfs(og.graph.n, off.data(), og.graph.out_targets.data(), s_id, goals.data(), (uint32_t)goals.size(), par.data());
        for (const Tuple &t : ends.iter()) {
            const uint32_t g_id = og.inv_indices.at(t[0]);
            Tuple want = T({s[0], t[0], DataValue()});
            if (par[g_id] != ORC_NONE) {
                std::vector<DataValue> path;
                for (uint32_t c = g_id; c != s_id; c = par[c]) path.push_back(og.indices[c]);
                path.push_back(og.indices[s_id]);
                std::reverse(path.begin(), path.end());
                want[2] = DataValue::list(path);
            }
            same = same && sp.exists(want);
            checked++;
        }
    }
    CHECK(same && sp.size() == checked);
    // --- BFS with a condition on the node id, limit 3 (bfs.rs): compare with the oracle's FIFO discovery order
    {
        int64_t threshold = 5000;
        ExprOption cond{[threshold](const Tuple &t) {
                            int64_t v = 0;
                            return t[0].get_int(&v) && v > threshold;
                        },
                        true};
        FixedRuleInputRelation nodes;  // unused: the condition binds only the id column
        RegularTempStore got = reg.run("BFS", FixedRulePayload("BFS", {rel, nodes, starts}, {{"limit", DataValue(3)}}, {{"condition", cond}}),
                                       Poison());
        std::vector<DataValue> sv;
        for (const Tuple &t : starts.iter()) sv.push_back(t[0]);
        GraphWithIndices bg = rel.as_ordered_graph(sv);
        std::vector<uint64_t> boff = to_u64(bg.graph.out_offsets);
        std::vector<uint8_t> visited(bg.graph.n, 0);
        std::vector<uint32_t> par(bg.graph.n, ORC_NONE), order(bg.graph.n);
        RegularTempStore want;
        size_t found = 0;
        for (const Tuple &s : starts.iter()) {
            if (found >= 3) break;
            const uint32_t s_id = bg.inv_indices.at(s[0]);
            if (visited[s_id]) continue;
            const uint32_t cnt = orc_bfs_order(bg.graph.n, boff.data(), bg.graph.out_targets.data(), s_id, visited.data(), par.data(), order.data());
            for (uint32_t j = 0; j < cnt && found < 3; j++) {
                int64_t v = 0;
                bg.indices[order[j]].get_int(&v);
                if (v > threshold) {
                    std::vector<DataValue> path;
                    for (uint32_t c = order[j]; c != s_id; c = par[c]) path.push_back(bg.indices[c]);
                    path.push_back(bg.indices[s_id]);
                    std::reverse(path.begin(), path.end());
                    want.put(T({s[0], bg.indices[order[j]], DataValue::list(path)}));
                    found++;
                }
            }
        }
        CHECK(got.rows() == want.rows() && got.size() == 3);
    }
    // --- ConnectedComponents vs Tarjan on the symmetrised first-appearance graph
    {
        RegularTempStore cc = reg.run("ConnectedComponents", FixedRulePayload("ConnectedComponents", {rel}), Poison());
        GraphWithIndices g = rel.as_directed_graph(true);
        std::vector<uint64_t> o2 = to_u64(g.graph.out_offsets);
        std::vector<uint32_t> grp(g.graph.n);
        orc_tarjan_groups(g.graph.n, o2.data(), g.graph.out_targets.data(), grp.data());
        bool ok = cc.size() == g.graph.n;
        for (const Tuple &t : cc) {
            int64_t gid = 0;
            ok = ok && t[1].get_int(&gid) && gid == (int64_t)grp[g.inv_indices.at(t[0])];
        }
        CHECK(ok);
    }
    // --- ShortestPathDijkstra: costs bit-exact vs the oracle, paths valid and tight
    {
        std::vector<Tuple> wrows = random_edges(1500, 6000, 7, true);
        FixedRuleInputRelation wrel(wrows);
        FixedRuleInputRelation wst({T({wrows[3][0]}), T({wrows[900][0]})});
        RegularTempStore dj = reg.run("ShortestPathDijkstra", FixedRulePayload("ShortestPathDijkstra", {wrel, wst}, {{"undirected", DataValue(true)}}),
                                      Poison());
        GraphWithIndices g = wrel.as_directed_weighted_graph(true, false);
        std::vector<uint64_t> o2 = to_u64(g.graph.out_offsets);
        bool ok = true;
        size_t nrows = 0;
        for (const Tuple &s : wst.iter()) {
            const uint32_t s_id = g.inv_indices.at(s[0]);
            std::vector<float> dist(g.graph.n);
            std::vector<uint32_t> par(g.graph.n);
            orc_dijkstra(g.graph.n, o2.data(), g.graph.out_targets.data(), g.graph.out_weights.data(), s_id, nullptr, 0, dist.data(), par.data());
            nrows += g.graph.n;
            for (const Tuple &t : dj) {
                if (!(t[0] == s[0])) continue;
                double cost;
                t[2].get_float(&cost);
                const uint32_t t_id = g.inv_indices.at(t[1]);
                ok = ok && ((double)dist[t_id] == cost || (std::isinf(cost) && std::isinf(dist[t_id])));
                const std::vector<DataValue> &path = *t[3].get_slice();
                if (std::isinf(cost)) {
                    ok = ok && path.empty();
                    continue;
                }
                // the path starts at s, ends at t, follows edges and its f32 running cost equals the reported cost
                ok = ok && !path.empty() && path.front() == s[0] && path.back() == t[1];
                float run = 0.f;
                for (size_t i = 0; ok && i + 1 < path.size(); i++) {
                    const uint32_t a = g.inv_indices.at(path[i]), b = g.inv_indices.at(path[i + 1]);
                    float best = INFINITY;
                    for (uint32_t e2 = g.graph.out_offsets[a]; e2 < g.graph.out_offsets[a + 1]; e2++)
                        if (g.graph.out_targets[e2] == b && run + g.graph.out_weights[e2] == dist[b]) best = g.graph.out_weights[e2];
                    ok = ok && std::isfinite(best);
                    run = dist[b];
                }
            }
        }
        CHECK(ok && dj.size() == nrows);
    }
}

static void gpu_clustering_coefficients() {
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::vector<Tuple> rows = random_edges(300, 2500, 21, false);
    for (size_t i = 0; i < 100; i++) rows.push_back(T({rows[i][1], rows[i][0]}));  // bidirectional pairs: doubled multiplicity
    rows.push_back(T({rows[0][0], rows[0][0]}));                                    // a self loop
    FixedRuleInputRelation rel(rows);
    RegularTempStore out = reg.run("ClusteringCoefficients", FixedRulePayload("ClusteringCoefficients", {rel}), Poison());
    GraphWithIndices g = rel.as_directed_graph(true);
    std::vector<uint64_t> off = to_u64(g.graph.out_offsets);
    std::vector<double> cc(g.graph.n);
    std::vector<uint64_t> tri(g.graph.n);
    std::vector<uint32_t> deg(g.graph.n);
    orc_clustering_coefficients(g.graph.n, off.data(), g.graph.out_targets.data(), cc.data(), tri.data(), deg.data());
    bool ok = out.size() == g.graph.n;
    for (const Tuple &t : out) {
        const uint32_t v = g.inv_indices.at(t[0]);
        double c;
        int64_t tt, dd;
        ok = ok && t[1].get_float(&c) && c == cc[v] && t[2].get_int(&tt) && (uint64_t)tt == tri[v] && t[3].get_int(&dd) && (uint32_t)dd == deg[v];
    }
    CHECK(ok);
}

static void gpu_closeness_centrality() {
    // all_pairs_shortest_path.rs:97-176: Dijkstra costs from every node (oracle), f32 sum in node order, nc * nc / total / (n - 1)
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::vector<Tuple> rows = random_edges(300, 1500, 33, true);
    FixedRuleInputRelation rel(rows);
    RegularTempStore out = reg.run("ClosenessCentrality", FixedRulePayload("ClosenessCentrality", {rel}, {{"undirected", DataValue(true)}}), Poison());
    GraphWithIndices g = rel.as_directed_weighted_graph(true, false);
    const uint32_t n = g.graph.n;
    std::vector<uint64_t> off = to_u64(g.graph.out_offsets);
    bool ok = out.size() == n;
    std::vector<float> dist(n);
    std::vector<uint32_t> par(n);
    for (const Tuple &t : out) {
        const uint32_t s = g.inv_indices.at(t[0]);
        orc_dijkstra(n, off.data(), g.graph.out_targets.data(), g.graph.out_weights.data(), s, nullptr, 0, dist.data(), par.data());
        float total = 0.f, nc = 0.f;
        for (uint32_t v = 0; v < n; v++)
            if (std::isfinite(dist[v])) {
                total = total + dist[v];
                nc = nc + 1.f;
            }
        const float c = nc * nc / total / (float)(n - 1);
        double got;
        ok = ok && t[1].get_float(&got) && (got == (double)c || (std::isnan(got) && std::isnan(c)));
    }
    CHECK(ok);
}

static void gpu_betweenness_centrality() {
    // all_pairs_shortest_path.rs:31-95: against the oracle's literal enumeration of all shortest paths (f32, the reference's
    // order); small integer weights make ties, i.e. several shortest paths per pair and fractional shares
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::mt19937_64 rng(41);
    std::vector<Tuple> rows;
    for (int i = 0; i < 110; i++) {
        const int64_t a = (int64_t)(rng() % 30), b = (int64_t)(rng() % 30);
        if (a != b) rows.push_back(T({DataValue(a), DataValue(b), DataValue((int64_t)(1 + rng() % 3))}));
    }
    for (bool undirected : {false, true}) {
        FixedRuleInputRelation rel(rows);
        RegularTempStore out = reg.run("BetweennessCentrality",
                                       FixedRulePayload("BetweennessCentrality", {rel}, {{"undirected", DataValue(undirected)}}), Poison());
        GraphWithIndices g = rel.as_directed_weighted_graph(undirected, false);
        const uint32_t n = g.graph.n;
        std::vector<uint64_t> off = to_u64(g.graph.out_offsets);
        std::vector<float> want(n);
        CHECK(orc_betweenness(n, off.data(), g.graph.out_targets.data(), g.graph.out_weights.data(), want.data(), 10000000) == 0);
        bool ok = out.size() == n, fractional = false;
        for (const Tuple &t : out) {
            const uint32_t v = g.inv_indices.at(t[0]);
            double got = 0;
            ok = ok && t[1].get_float(&got) && std::fabs(got - (double)want[v]) <= 1e-5 * std::fabs((double)want[v]) + 1e-6;
            fractional |= std::fabs(want[v] - std::round(want[v])) > 1e-3f;
        }
        CHECK(ok && fractional);
    }
    CHECK((throws<CozoError>([&] {
        reg.run("BetweennessCentrality", FixedRulePayload("BetweennessCentrality", {FixedRuleInputRelation({T({DataValue("a"), DataValue("b"), DataValue(0.0)})})}), Poison());
    }, "algo::betweenness_needs_positive_weights")));
}

static void gpu_label_propagation() {
    // label_propagation.rs:56-109 in the fixed order (colour classes ascending) with the smallest label on ties: the oracle's loop
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::mt19937_64 rng(77);
    std::vector<Tuple> rows;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 120; i++) {
            const int64_t a = c * 25 + (int64_t)(rng() % 25), b = c * 25 + (int64_t)(rng() % 25);
            rows.push_back(T({DataValue(a), DataValue(b), DataValue((double)(1 + rng() % 8) / 4.0)}));
        }
    for (int i = 0; i < 10; i++) rows.push_back(T({DataValue((int64_t)(rng() % 100)), DataValue((int64_t)(rng() % 100)), DataValue(0.25)}));
    for (bool undirected : {false, true}) {
        FixedRuleInputRelation rel(rows);
        RegularTempStore out = reg.run("LabelPropagation",
                                       FixedRulePayload("LabelPropagation", {rel}, {{"undirected", DataValue(undirected)}, {"max_iter", DataValue((int64_t)10)}}),
                                       Poison());
        GraphWithIndices g = rel.as_directed_weighted_graph(undirected, true);
        const uint32_t n = g.graph.n;
        std::vector<uint64_t> off = to_u64(g.graph.out_offsets);
        std::vector<uint32_t> colour(n), order, want(n);
        const uint32_t k = orc_lp_colouring(n, off.data(), g.graph.out_targets.data(), colour.data());
        for (uint32_t c = 0; c < k; c++)
            for (uint32_t v = 0; v < n; v++)
                if (colour[v] == c) order.push_back(v);
        CHECK(orc_label_propagation_in_order(n, off.data(), g.graph.out_targets.data(), g.graph.out_weights.data(), order.data(), 10, want.data()) > 0);
        bool ok = out.size() == n;
        std::set<int64_t> distinct;
        for (const Tuple &t : out) {
            int64_t lab = -1;
            ok = ok && t[0].get_int(&lab) && lab == (int64_t)want[g.inv_indices.at(t[1])];
            distinct.insert(lab);
        }
        CHECK(ok && distinct.size() < n / 4);
    }
    CHECK((throws<CozoError>([&] {
        reg.run("LabelPropagation", FixedRulePayload("LabelPropagation", {FixedRuleInputRelation(rows)}, {{"max_iter", DataValue((int64_t)0)}}), Poison());
    })));
}

static void gpu_dijkstra_keep_ties() {
    // ShortestPathDijkstra{keep_ties: true} (shortest_path_dijkstra.rs:341-450): every shortest path is a row.  Checked by
    // properties that do not share the implementation's route: every row's path is a real path whose f32 cost, added left to
    // right, is the oracle's Dijkstra cost; the rows of a pair are distinct; and their number equals the path count of an
    // independent DP over nodes in order of distance.
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::mt19937_64 rng(57);
    std::vector<Tuple> rows;
    for (int i = 0; i < 130; i++) {
        const int64_t a = (int64_t)(rng() % 32), b = (int64_t)(rng() % 32);
        if (a != b) rows.push_back(T({DataValue(a), DataValue(b), DataValue((int64_t)(1 + rng() % 2))}));
    }
    FixedRuleInputRelation rel(rows);
    GraphWithIndices g = rel.as_directed_weighted_graph(true, false);
    const uint32_t n = g.graph.n;
    std::vector<Tuple> st = {T({g.indices[0]}), T({g.indices[5]})}, en;
    for (uint32_t v = 0; v < n; v += 2) en.push_back(T({g.indices[v]}));
    RegularTempStore out = reg.run("ShortestPathDijkstra", FixedRulePayload("ShortestPathDijkstra", {rel, FixedRuleInputRelation(st), FixedRuleInputRelation(en)},
                                                                              {{"undirected", DataValue(true)}, {"keep_ties", DataValue(true)}}), Poison());
    std::vector<uint64_t> off = to_u64(g.graph.out_offsets);
    bool ok = true, saw_tie = false;
    size_t checked = 0;
    for (uint32_t s : {0u, 5u}) {
        std::vector<float> dist(n);
        std::vector<uint32_t> par(n), order(n);
        orc_dijkstra(n, off.data(), g.graph.out_targets.data(), g.graph.out_weights.data(), s, nullptr, 0, dist.data(), par.data());
        for (uint32_t v = 0; v < n; v++) order[v] = v;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return dist[a] < dist[b]; });
        std::vector<double> sigma(n, 0.0);
        sigma[s] = 1.0;
        for (uint32_t u : order) {
            if (!std::isfinite(dist[u])) continue;
            for (uint32_t e = g.graph.out_offsets[u]; e < g.graph.out_offsets[u + 1]; e++)
                if ((float)(dist[u] + g.graph.out_weights[e]) == dist[g.graph.out_targets[e]]) sigma[g.graph.out_targets[e]] += sigma[u];
        }
        std::map<uint32_t, size_t> count;
        for (const Tuple &t : out) {
            if (g.inv_indices.at(t[0]) != s) continue;
            const uint32_t tv = g.inv_indices.at(t[1]);
            const std::vector<DataValue> &path = *t[3].get_slice();
            double cost = 0;
            t[2].get_float(&cost);
            if (path.empty()) {
                ok &= std::isinf(cost) && !std::isfinite(dist[tv]);
                continue;
            }
            count[tv]++;
            ok &= path.front() == g.indices[s] && path.back() == g.indices[tv] && cost == (double)dist[tv];
            float acc = 0.0f;
            for (size_t i = 0; i + 1 < path.size(); i++) {  // every hop is an edge; the cheapest parallel edge carries the path
                const uint32_t u = g.inv_indices.at(path[i]), v = g.inv_indices.at(path[i + 1]);
                float best = INFINITY;
                for (uint32_t e = g.graph.out_offsets[u]; e < g.graph.out_offsets[u + 1]; e++)
                    if (g.graph.out_targets[e] == v) best = std::min(best, g.graph.out_weights[e]);
                ok &= std::isfinite(best);
                acc = acc + best;
            }
            ok &= acc == dist[tv];
            checked++;
        }
        for (uint32_t v = 0; v < n; v += 2) {
            if (v == s || !std::isfinite(dist[v])) {
                ok &= count.count(v) == 0;  // the start as its own target: no row; unreachable: the (inf, []) row only
                continue;
            }
            // the set store folds identical paths (parallel edges of equal weight): distinct rows <= sigma, and >= 1
            ok &= count[v] >= 1 && (double)count[v] <= sigma[v];
            saw_tie |= count[v] >= 2;
        }
    }
    CHECK(ok && saw_tie && checked > 20);
    // without a termination relation keep_ties changes nothing (:73-86)
    RegularTempStore plain = reg.run("ShortestPathDijkstra", FixedRulePayload("ShortestPathDijkstra", {rel, FixedRuleInputRelation(st)}), Poison());
    RegularTempStore kt = reg.run("ShortestPathDijkstra", FixedRulePayload("ShortestPathDijkstra", {rel, FixedRuleInputRelation(st)}, {{"keep_ties", DataValue(true)}}), Poison());
    CHECK(plain.rows() == kt.rows() && plain.size() == 2 * (size_t)n);
}

static void gpu_rules_on_stored_relation() {
    // every rule off the stored bytes of its edge relation (FixedRuleInputRelation::from_stored -> libcozo_ingest) and off
    // the decoded tuples: the same rows
    FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
    std::vector<Tuple> rows;
    std::mt19937_64 rng(77);
    for (int i = 0; i < 900; i++) {
        const std::string a = "n" + std::to_string(rng() % 150), b = "n" + std::to_string(rng() % 150);
        if (a != b) rows.push_back(T({DataValue(a), DataValue(b), DataValue((double)(1 + rng() % 40) / 4)}));
    }
    for (uint32_t kcols : {3u, 2u}) {
        const StoredRows stored = StoredRows::from_tuples(4, rows, kcols);
        std::vector<Tuple> decoded;
        for (size_t i = 0; i < stored.size(); i++) decoded.push_back(stored.tuple(i));
        const FixedRuleInputRelation plain(decoded), fast = FixedRuleInputRelation::from_stored(stored);
        const FixedRuleInputRelation starts({T({decoded[0][0]}), T({decoded[40][0]})}), ends({T({decoded[7][1]}), T({decoded[99][1]})});
        auto both = [&](const char *name, std::vector<std::optional<FixedRuleInputRelation>> extra, std::map<std::string, DataValue> opts) {
            std::vector<std::optional<FixedRuleInputRelation>> a{plain}, b{fast};
            a.insert(a.end(), extra.begin(), extra.end());
            b.insert(b.end(), extra.begin(), extra.end());
            const RegularTempStore ra = reg.run(name, FixedRulePayload(name, a, opts), Poison());
            const RegularTempStore rb = reg.run(name, FixedRulePayload(name, b, opts), Poison());
            CHECK(ra.size() > 0 && ra.rows() == rb.rows());
        };
        both("PageRank", {}, {});
        both("PageRank", {}, {{"undirected", DataValue(true)}, {"iterations", DataValue((int64_t)3)}});
        both("ConnectedComponents", {}, {});
        both("ShortestPathBFS", {starts, ends}, {});
        both("ShortestPathDijkstra", {starts}, {});
        both("ShortestPathDijkstra", {starts, ends}, {{"undirected", DataValue(true)}});
        both("ClusteringCoefficients", {}, {});
        both("DegreeCentrality", {}, {});
    }
}

static void gpu_hnsw_search_ra() {
    // a base relation {k => v: <F32; 24>} with an L2 index, searched through HnswSearchRA (runtime/tests.rs:700-809 shape)
    const size_t n = 3000, dim = 24;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    BaseRelation base;
    base.keys = {"k"};
    base.non_keys = {"v"};
    std::vector<float> flat;
    for (size_t i = 0; i < n; i++) {
        std::vector<float> v(dim);
        for (float &x : v) x = U(rng);
        flat.insert(flat.end(), v.begin(), v.end());
        base.rows.push_back(T({DataValue((int64_t)i), DataValue(F32Vec{v})}));
    }
    HnswIndexManifest mf = HnswIndexManifest::create("a", "vec", dim, {1}, HnswDistance::L2, 8, 40);
    std::vector<int32_t> levels(n);
    {
        std::mt19937_64 r2(9);
        for (auto &l : levels) {
            double u = (double)(r2() >> 11) / 9007199254740992.0;
            l = (int32_t)std::floor(-std::log(u > 0 ? u : 1e-300) * mf.level_multiplier);
        }
    }
    // max_batch = 1 == the sequential hnsw_put order: the link tables equal the oracle's, so the search rows do too
    GpuHnswIndex ix = GpuHnswIndex::create(mf, base, 0, 1, &levels);
    CHECK(ix.node_count() == n && ix.device_bytes() > n * dim * 4);
    orc_hnsw *ob = orc_hnsw_new((int)dim, ORC_L2, 8, 40, 0, 0, ORC_DOT_GPU);
    orc_hnsw_insert(ob, flat.data(), (uint32_t)n, levels.data());
    const int nl = orc_hnsw_n_levels(ob);
    std::vector<uint32_t> lsize(nl);
    std::vector<int32_t> lwidth(nl);
    std::vector<std::vector<uint32_t>> lnodes(nl), lnbrs(nl);
    std::vector<const uint32_t *> pn(nl), pb(nl);
    for (int l = 0; l < nl; l++) {
        lsize[l] = orc_hnsw_level_size(ob, l);
        lwidth[l] = orc_hnsw_level_width(ob, l);
        lnodes[l].resize(lsize[l]);
        lnbrs[l].resize((size_t)lsize[l] * lwidth[l]);
        orc_hnsw_export_level(ob, l, lnodes[l].data(), lnbrs[l].data());
        pn[l] = lnodes[l].data();
        pb[l] = lnbrs[l].data();
    }
    orc_flat_index fx = {(uint32_t)n, (int)dim, ORC_L2, ORC_DOT_GPU, flat.data(), nl, lsize.data(), lwidth.data(), pn.data(), pb.data(),
                         orc_hnsw_entry(ob)};
    // parent tuples (qid, query vector); k = 5, ef = 30, bind distance + field, radius, and a filter on the key
    std::vector<Tuple> parent;
    std::vector<std::vector<float>> qs;
    for (int i = 0; i < 20; i++) {
        std::vector<float> q(dim);
        for (float &x : q) x = U(rng);
        qs.push_back(q);
        parent.push_back(T({DataValue((int64_t)i), DataValue(F32Vec{q})}));
    }
    HnswSearchRA ra{&ix, HnswSearch{}, 1};
    ra.hnsw_search.k = 5;
    ra.hnsw_search.ef = 30;
    ra.hnsw_search.bind_distance = true;
    std::vector<Tuple> got = ra.iter(parent, Poison());
    bool ok = true;
    size_t pos = 0;
    for (int i = 0; i < 20; i++) {
        uint32_t ids[5];
        double dd[5];
        uint64_t nd = 0;
        const int cnt = orc_hnsw_knn(&fx, qs[i].data(), 5, 30, 0, 0.0, ids, dd, &nd);
        for (int j = 0; j < cnt; j++, pos++) {
            ok = ok && pos < got.size();
            if (!ok) break;
            const Tuple &t = got[pos];  // (qid, q, k, v, dist)
            int64_t qid, key;
            double d;
            ok = ok && t.size() == 5 && t[0].get_int(&qid) && qid == i && t[2].get_int(&key) && key == (int64_t)ids[j] &&
                 t[4].get_float(&d) && d == dd[j];
        }
    }
    CHECK(ok && pos == got.size());
    // filter + radius: the filter sees all ef candidates (hnsw.rs:943-947), then truncate(k)
    ra.hnsw_search.filter = [](const Tuple &t) {
        int64_t key;
        return t[0].get_int(&key) && key % 2 == 0;
    };
    ra.hnsw_search.radius = 1.2;
    got = ra.iter(parent, Poison());
    ok = true;
    pos = 0;
    for (int i = 0; i < 20; i++) {
        uint32_t ids[30];
        double dd[30];
        uint64_t nd = 0;
        const int cnt = orc_hnsw_knn(&fx, qs[i].data(), 30, 30, 0, 0.0, ids, dd, &nd);
        int taken = 0;
        for (int j = 0; j < cnt && taken < 5; j++) {
            if (dd[j] > 1.2 || ids[j] % 2 != 0) continue;
            ok = ok && pos < got.size();
            if (!ok) break;
            int64_t key;
            ok = ok && got[pos][2].get_int(&key) && key == (int64_t)ids[j];
            pos++;
            taken++;
        }
    }
    CHECK(ok && pos == got.size());
    // the same search with the comparison handed over as column predicates: the key column is purely Int, so they are
    // evaluated on the device over all ef candidates (cz_hnsw_search_filtered) -- `k >= 1000 and k != 1500`, radius as before;
    // rows must equal the host-filter route's
    {
        HnswSearchRA dev_ra{&ix, HnswSearch{}, 1}, host_ra{&ix, HnswSearch{}, 1};
        for (HnswSearchRA *r : {&dev_ra, &host_ra}) {
            r->hnsw_search.k = 5;
            r->hnsw_search.ef = 30;
            r->hnsw_search.bind_distance = true;
            r->hnsw_search.radius = 1.2;
        }
        dev_ra.hnsw_search.predicates = {ColumnPredicate{0, CZ_OP_GE, DataValue((int64_t)1000)}, ColumnPredicate{0, CZ_OP_NE, DataValue(1500.0)}};
        host_ra.hnsw_search.filter = [](const Tuple &t) {
            int64_t key;
            return t[0].get_int(&key) && key >= 1000 && key != 1500;
        };
        const std::vector<Tuple> a = dev_ra.iter(parent, Poison()), b = host_ra.iter(parent, Poison());
        CHECK(!a.empty() && a == b);
        // a predicate on the vector column cannot go to the device: same rows through the host comparison ... which raises the
        // reference's error for a non-numeric value
        dev_ra.hnsw_search.predicates = {ColumnPredicate{1, CZ_OP_GE, DataValue((int64_t)0)}};
        CHECK((throws<CozoError>([&] { dev_ra.iter(parent, Poison()); })));
        // ... but op_eq / op_neq never check types (data/functions.rs:298-304, :337-343): `v != 0` holds for every row and
        // `v == 0` for none, exactly as if the filter were absent / always false
        HnswSearchRA plain{&ix, HnswSearch{}, 1};
        plain.hnsw_search = host_ra.hnsw_search;
        plain.hnsw_search.filter.reset();
        dev_ra.hnsw_search.predicates = {ColumnPredicate{1, CZ_OP_NE, DataValue((int64_t)0)}};
        CHECK(dev_ra.iter(parent, Poison()) == plain.iter(parent, Poison()));
        dev_ra.hnsw_search.predicates = {ColumnPredicate{1, CZ_OP_EQ, DataValue((int64_t)0)}};
        CHECK(dev_ra.iter(parent, Poison()).empty());
    }
    CHECK((throws<CozoError>([&] { ra.iter({T({DataValue(1), DataValue("not a vector")})}, Poison()); })));
    CHECK((throws<CozoError>([&] { ix.hnsw_knn(std::vector<float>(dim + 1, 0.f), HnswSearch{}, Poison()); })));
    orc_hnsw_free(ob);
}

static void gpu_index_through_the_store() {
    // build on the GPU -> the `tbl:idx` rows as stored bytes (GpuHnswIndex::index_rows) -> read back off those bytes
    // (GpuHnswIndex::from_stored): the same index, the same rows from HnswSearchRA
    const size_t n = 1500, dim = 24;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    BaseRelation base;
    base.keys = {"id"};
    base.non_keys = {"tag", "v"};
    for (size_t i = 0; i < n; i++) {
        std::vector<float> v(dim);
        for (float &x : v) x = U(rng);
        char key[16];
        std::snprintf(key, sizeof key, "doc-%05zu", i);  // string keys in key order
        base.rows.push_back(T({DataValue(std::string(key)), DataValue((int64_t)(i % 7)), DataValue(F32Vec{v})}));
    }
    HnswIndexManifest mf = HnswIndexManifest::create("docs", "vec", dim, {2}, HnswDistance::Cosine, 8, 40);
    GpuHnswIndex built = GpuHnswIndex::create(mf, base, 11, 64, nullptr);
    const StoredRows idx = built.index_rows(31), stored_base = StoredRows::from_tuples(30, base.rows, 1);
    CHECK(idx.n_key_cols == 7 && idx.size() > n);
    // the rows read like the reference's: first row = the entry point on the top layer, last row = the canary
    const Tuple first = idx.tuple(0), last = idx.tuple(idx.size() - 1);
    int64_t top = 0, canary_layer = 0;
    CHECK(first[0].get_int(&top) && top <= 0 && first.size() == 10 && first[1] == first[4] && first[2] == first[5]);
    CHECK(last[0].get_int(&canary_layer) && canary_layer == 1 && last[1].is_null() && last[6].is_null());
    size_t self_rows = 0, link_rows = 0;
    bool shapes = true;
    for (size_t i = 0; i + 1 < idx.size(); i++) {
        const Tuple t = idx.tuple(i);
        const bool self = t[1] == t[4] && t[2] == t[5] && t[3] == t[6];
        bool ignore = true;
        shapes &= t.size() == 10 && t[7].is_float() && t[9].get_bool(&ignore) && !ignore && (self ? !t[8].is_null() : t[8].is_null());
        (self ? self_rows : link_rows)++;
    }
    CHECK(shapes && self_rows >= n && link_rows > n);
    GpuHnswIndex read = GpuHnswIndex::from_stored(mf, idx, stored_base, base);
    CHECK(read.node_count() == built.node_count());
    bool same_nodes = true;
    for (uint32_t v = 0; v < read.node_count(); v++)
        same_nodes &= read.node(v).row == built.node(v).row && read.node(v).field == built.node(v).field && read.node(v).sub == built.node(v).sub;
    CHECK(same_nodes);
    CHECK(read.index_rows(31).keys == idx.keys && read.index_rows(31).vals == idx.vals);  // and back again: a fixed point
    std::vector<Tuple> parent;
    for (int i = 0; i < 24; i++) {
        std::vector<float> q(dim);
        for (float &x : q) x = U(rng);
        parent.push_back(T({DataValue((int64_t)i), DataValue(F32Vec{q})}));
    }
    HnswSearch hs;
    hs.k = 6;
    hs.ef = 40;
    hs.bind_distance = true;
    hs.bind_field = true;
    hs.filter = [](const Tuple &t) {
        int64_t tag = 0;
        return t[1].get_int(&tag) && tag != 3;
    };
    const std::vector<Tuple> a = HnswSearchRA{&built, hs, 1}.iter(parent, Poison());
    const std::vector<Tuple> b = HnswSearchRA{&read, hs, 1}.iter(parent, Poison());
    CHECK(!a.empty() && a == b);
    // an empty index goes through the store as nothing at all
    BaseRelation none;
    none.keys = {"id"};
    none.non_keys = {"tag", "v"};
    GpuHnswIndex empty = GpuHnswIndex::create(mf, none, 0, 0, nullptr);
    CHECK(empty.index_rows(31).size() == 0);
    GpuHnswIndex empty_read = GpuHnswIndex::from_stored(mf, empty.index_rows(31), StoredRows::from_tuples(30, {}, 1), none);
    const HnswSearchRA empty_ra{&empty_read, hs, 1};
    CHECK(empty_read.node_count() == 0 && empty_ra.iter(parent, Poison()).empty());
}

static void gpu_index_over_rows_with_several_vectors() {
    // hnsw_put indexes every Vec inside a List (runtime/hnsw.rs:694-706); hnsw_get_neighbours never returns a link between two
    // vectors of one base row (:609-610).  create() hands the library every node's base row (cz_hnsw_set_row_of): the stored
    // rows hold no link inside a row, the self rows' degrees count them, and the index reads back as itself.
    const size_t n_rows = 300, dim = 12;
    std::mt19937 rng(17);
    std::normal_distribution<float> N(0.f, 1.f);
    BaseRelation base;
    base.keys = {"id"};
    base.non_keys = {"vs"};
    size_t n_vec = 0;
    for (size_t i = 0; i < n_rows; i++) {
        std::vector<float> centre(dim);
        for (float &x : centre) x = N(rng);
        std::vector<DataValue> items;
        for (size_t j = 0; j < 1 + i % 3; j++) {  // 1..3 vectors per row, close to each other
            std::vector<float> v = centre;
            for (float &x : v) x += 0.05f * N(rng);
            items.push_back(DataValue(F32Vec{v}));
            n_vec++;
        }
        base.rows.push_back(T({DataValue((int64_t)i), DataValue::list(items)}));
    }
    HnswIndexManifest mf = HnswIndexManifest::create("docs", "vec", dim, {1}, HnswDistance::L2, 4, 16);
    GpuHnswIndex built = GpuHnswIndex::create(mf, base, 3, 1, nullptr);
    CHECK(built.node_count() == n_vec);
    const StoredRows idx = built.index_rows(41), stored_base = StoredRows::from_tuples(40, base.rows, 1);
    size_t inside = 0, link_rows = 0;
    double degrees = 0;
    for (size_t i = 0; i + 1 < idx.size(); i++) {
        const Tuple t = idx.tuple(i);  // [layer, fr key, fr field, fr sub, to key, to field, to sub] -> [f64, hash | Null, ignore]
        const bool self = t[1] == t[4] && t[2] == t[5] && t[3] == t[6];
        if (!self && t[1] == t[4]) inside++;
        double d = 0;
        if (self && t[7].get_float(&d)) degrees += d;
        if (!self) link_rows++;
    }
    CHECK(inside == 0);
    CHECK(degrees > (double)link_rows);  // row mates were selected (they are each other's nearest), counted, and not kept
    GpuHnswIndex read = GpuHnswIndex::from_stored(mf, idx, stored_base, base);
    CHECK(read.node_count() == built.node_count());
    std::vector<Tuple> parent;
    for (int i = 0; i < 16; i++) {
        std::vector<float> q(dim);
        for (float &x : q) x = N(rng);
        parent.push_back(T({DataValue((int64_t)i), DataValue(F32Vec{q})}));
    }
    HnswSearch hs;
    hs.k = 5;
    hs.ef = 30;
    hs.bind_distance = true;
    const std::vector<Tuple> a = HnswSearchRA{&built, hs, 1}.iter(parent, Poison());
    const std::vector<Tuple> b = HnswSearchRA{&read, hs, 1}.iter(parent, Poison());
    CHECK(!a.empty() && a == b);
}

static void test_stored_rows_delta_cpu() {
    // stored_rows_delta: puts = new or changed rows, dels = vanished keys; old with the delta applied is new
    std::mt19937 rng(17);
    for (int trial = 0; trial < 20; trial++) {
        std::vector<Tuple> old_t, new_t;
        size_t want_puts = 0, want_dels = 0;
        for (int64_t k = 0; k < 300; k++) {
            if (rng() % 2) continue;
            old_t.push_back(T({DataValue(k), DataValue(std::string("v")), DataValue((double)k)}));
            const unsigned what = rng() % 10;
            if (what == 0) { want_dels++; continue; }
            if (what == 1) { new_t.push_back(T({DataValue(k), DataValue(std::string("changed")), DataValue((double)k)})); want_puts++; }
            else new_t.push_back(old_t.back());
        }
        for (int64_t k = 300; k < 330; k++)
            if (rng() % 3 == 0) { new_t.push_back(T({DataValue(k), DataValue(std::string("new")), DataValue(0.5)})); want_puts++; }
        const StoredRows a = StoredRows::from_tuples(9, old_t, 1), b = StoredRows::from_tuples(9, new_t, 1);
        StoredRows puts;
        std::vector<std::vector<uint8_t>> dels;
        stored_rows_delta(a, b, &puts, &dels);
        CHECK(puts.size() == want_puts && dels.size() == want_dels);
        std::map<std::vector<uint8_t>, std::vector<uint8_t>> kv;
        for (size_t i = 0; i < a.size(); i++)
            kv[std::vector<uint8_t>(a.keys.begin() + a.key_off[i], a.keys.begin() + a.key_off[i + 1])] =
                std::vector<uint8_t>(a.vals.begin() + a.val_off[i], a.vals.begin() + a.val_off[i + 1]);
        for (const auto &k : dels) kv.erase(k);
        for (size_t i = 0; i < puts.size(); i++)
            kv[std::vector<uint8_t>(puts.keys.begin() + puts.key_off[i], puts.keys.begin() + puts.key_off[i + 1])] =
                std::vector<uint8_t>(puts.vals.begin() + puts.val_off[i], puts.vals.begin() + puts.val_off[i + 1]);
        std::vector<uint8_t> keys, vals;
        for (const auto &e : kv) {
            keys.insert(keys.end(), e.first.begin(), e.first.end());
            vals.insert(vals.end(), e.second.begin(), e.second.end());
        }
        CHECK(kv.size() == b.size() && keys == b.keys && vals == b.vals);
    }
}

static void gpu_index_maintenance_writeback() {
    // hnsw_put / hnsw_remove on a later write, on the device (GpuHnswIndex::put_rows / remove_rows), and what goes back to the
    // store: build(first rows) + put_rows(the rest) with max_batch = 1 leaves byte for byte the `tbl:idx` rows of ONE sequential
    // build over all rows; the delta against the rows of the first index is a small part of them and turns the one into the
    // other; after remove_rows no row names a removed node and the delta applies again.
    const size_t n0 = 1200, n1 = 30, dim = 16;
    std::mt19937 rng(23);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    BaseRelation base, all;
    base.keys = all.keys = {"id"};
    base.non_keys = all.non_keys = {"v"};
    for (size_t i = 0; i < n0 + n1; i++) {
        std::vector<float> v(dim);
        for (float &x : v) x = U(rng);
        char key[16];
        std::snprintf(key, sizeof key, "r%05zu", i);
        all.rows.push_back(T({DataValue(std::string(key)), DataValue(F32Vec{v})}));
        if (i < n0) base.rows.push_back(all.rows.back());
    }
    HnswIndexManifest mf = HnswIndexManifest::create("t", "vec", dim, {1}, HnswDistance::L2, 8, 40);
    std::vector<int32_t> levels(n0 + n1);
    {
        std::mt19937_64 r2(4);
        for (auto &l : levels) {
            const double u = (double)(r2() >> 11) / 9007199254740992.0;
            l = (int32_t)std::floor(-std::log(u > 0 ? u : 1e-300) * mf.level_multiplier);
        }
    }
    const std::vector<int32_t> lv0(levels.begin(), levels.begin() + n0), lv1(levels.begin() + n0, levels.end());
    GpuHnswIndex ix = GpuHnswIndex::create(mf, base, 0, 1, &lv0);
    StoredRows stored = ix.index_rows(31);
    for (size_t i = n0; i < n0 + n1; i++) base.rows.push_back(all.rows[i]);
    ix.put_rows((uint32_t)n0, 0, 1, &lv1);
    CHECK(ix.node_count() == n0 + n1);
    const StoredRows after = ix.index_rows(31);
    GpuHnswIndex whole = GpuHnswIndex::create(mf, all, 0, 1, &levels);
    const StoredRows want = whole.index_rows(31);
    CHECK(after.keys == want.keys && after.vals == want.vals);
    StoredRows puts;
    std::vector<std::vector<uint8_t>> dels;
    stored_rows_delta(stored, after, &puts, &dels);
    CHECK(puts.size() > n1 && puts.size() < after.size() / 5 && dels.size() < after.size() / 20);
    auto apply = [](const StoredRows &a, const StoredRows &p, const std::vector<std::vector<uint8_t>> &d) {
        std::map<std::vector<uint8_t>, std::vector<uint8_t>> kv;
        for (size_t i = 0; i < a.size(); i++)
            kv[std::vector<uint8_t>(a.keys.begin() + a.key_off[i], a.keys.begin() + a.key_off[i + 1])] =
                std::vector<uint8_t>(a.vals.begin() + a.val_off[i], a.vals.begin() + a.val_off[i + 1]);
        for (const auto &k : d) kv.erase(k);
        for (size_t i = 0; i < p.size(); i++)
            kv[std::vector<uint8_t>(p.keys.begin() + p.key_off[i], p.keys.begin() + p.key_off[i + 1])] =
                std::vector<uint8_t>(p.vals.begin() + p.val_off[i], p.vals.begin() + p.val_off[i + 1]);
        std::pair<std::vector<uint8_t>, std::vector<uint8_t>> out;
        for (const auto &e : kv) {
            out.first.insert(out.first.end(), e.first.begin(), e.first.end());
            out.second.insert(out.second.end(), e.second.begin(), e.second.end());
        }
        return out;
    };
    auto applied = apply(stored, puts, dels);
    CHECK(applied.first == after.keys && applied.second == after.vals);
    // remove three rows: their self rows and every link row from / to them leave
    ix.remove_rows({5, 700, (uint32_t)n0 + 2});
    const StoredRows after2 = ix.index_rows(31);
    stored_rows_delta(after, after2, &puts, &dels);
    CHECK(!dels.empty() && puts.size() + dels.size() < after2.size() / 5);
    applied = apply(after, puts, dels);
    CHECK(applied.first == after2.keys && applied.second == after2.vals);
    bool named = false;
    for (size_t i = 0; i + 1 < after2.size(); i++) {  // (the last row is the canary)
        const Tuple t = after2.tuple(i);
        for (const char *gone : {"r00005", "r00700", "r01202"})
            named |= (t[1] == DataValue(std::string(gone))) || (t[4] == DataValue(std::string(gone)));
    }
    CHECK(!named);
    // a row carrying two vectors: each selects the other, both degrees say 1, and no link row is kept (hnsw.rs:609-610)
    {
        BaseRelation two;
        two.keys = {"id"};
        two.non_keys = {"v"};
        std::vector<float> a(dim, 0.25f), b(dim, 0.5f);
        two.rows.push_back(T({DataValue(std::string("x")), DataValue::list({DataValue(F32Vec{a}), DataValue(F32Vec{b})})}));
        const std::vector<int32_t> flat_levels{0, 0};
        GpuHnswIndex pair = GpuHnswIndex::create(mf, two, 0, 1, &flat_levels);
        const StoredRows rows = pair.index_rows(77);
        CHECK(pair.node_count() == 2 && rows.size() == 3);  // two self rows and the canary
        double d0 = -1, d1 = -1;
        CHECK(rows.tuple(0)[7].get_float(&d0) && rows.tuple(1)[7].get_float(&d1) && d0 == 1.0 && d1 == 1.0);
    }
    // the index still answers, and never with a removed row
    HnswSearchRA ra{&ix, HnswSearch{}, 1};
    ra.hnsw_search.k = 10;
    ra.hnsw_search.ef = 40;
    std::vector<Tuple> parent;
    for (int i = 0; i < 16; i++) {
        std::vector<float> q(dim);
        for (float &x : q) x = U(rng);
        parent.push_back(T({DataValue((int64_t)i), DataValue(F32Vec{q})}));
    }
    const std::vector<Tuple> got = ra.iter(parent, Poison());
    bool clean = got.size() == 160;
    for (const Tuple &t : got)
        for (const char *gone : {"r00005", "r00700", "r01202"}) clean &= !(t[2] == DataValue(std::string(gone)));
    CHECK(clean);
}

// ---- `test_host run-rule <in> <out>`: one rule invocation handed over by tests/test_mirrors_agree.py -----------------------------
// in : u32 magic, str rule, u32 n_options x (str name, blob memcmp-encoded DataValue), u32 n_inputs x (u32 n_rows x blob stored key)
// out: u32 1 + u32 n_rows x blob (the row as a stored key of relation 0)   |   u32 0 + str diagnostic code
// (str / blob = u32 length + bytes, little endian).  The Python mirror runs the same invocation; the rows must be the same bytes.
static int run_rule_from_file(const char *in_path, const char *out_path, bool need_device) {
    std::vector<uint8_t> buf;
    {
        FILE *f = std::fopen(in_path, "rb");
        if (!f) return 3;
        uint8_t tmp[65536];
        size_t got;
        while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
        std::fclose(f);
    }
    size_t at = 0;
    auto u32 = [&]() {
        if (at + 4 > buf.size()) throw std::runtime_error("truncated");
        uint32_t v;
        std::memcpy(&v, buf.data() + at, 4);
        at += 4;
        return v;
    };
    auto blob = [&]() {
        const uint32_t n = u32();
        if (at + n > buf.size()) throw std::runtime_error("truncated");
        std::vector<uint8_t> b(buf.begin() + at, buf.begin() + at + n);
        at += n;
        return b;
    };
    std::vector<uint8_t> out;
    auto put_u32 = [&](uint32_t v) {
        uint8_t b[4];
        std::memcpy(b, &v, 4);
        out.insert(out.end(), b, b + 4);
    };
    auto put_blob = [&](const std::vector<uint8_t> &b) {
        put_u32((uint32_t)b.size());
        out.insert(out.end(), b.begin(), b.end());
    };
    try {
        if (u32() != 0x52525A43u) throw std::runtime_error("bad magic");
        const std::vector<uint8_t> name_b = blob();
        const std::string name(name_b.begin(), name_b.end());
        std::map<std::string, DataValue> options;
        std::map<std::string, ExprOption> exprs;
        for (uint32_t n = u32(); n > 0; n--) {
            const std::vector<uint8_t> kb = blob(), vb = blob();
            const uint8_t *p = vb.data();
            const std::string key(kb.begin(), kb.end());
            DataValue val = decode_datavalue(p, vb.data() + vb.size());
            if (key.rfind("expr:", 0) == 0) {
                // an expression option on the wire (tests/test_mirrors_agree.py): [op, column, constant] = `tuple[column] OP constant`
                // in DataValue's total order; binding_indices = {column}
                const std::vector<DataValue> *l = val.get_slice();
                int64_t col = 0;
                if (!l || l->size() != 3 || !(*l)[0].get_str() || !(*l)[1].get_int(&col)) throw std::runtime_error("bad expr option");
                const std::string op = *(*l)[0].get_str();
                const DataValue c = (*l)[2];
                ExprOption e;
                e.only_first_binding = col == 0;
                e.eval = [op, col, c](const Tuple &t) {
                    const DataValue &v = t.at((size_t)col);
                    if (op == "eq") return v == c;
                    if (op == "ne") return !(v == c);
                    if (op == "lt") return v < c;
                    if (op == "ge") return !(v < c);
                    if (op == "gt") return c < v;
                    if (op == "le") return !(c < v);
                    throw std::runtime_error("bad expr op");
                };
                exprs[key.substr(5)] = std::move(e);
            } else {
                options[key] = std::move(val);
            }
        }
        std::vector<std::optional<FixedRuleInputRelation>> inputs;
        for (uint32_t n = u32(); n > 0; n--) {
            std::vector<Tuple> rows;
            for (uint32_t r = u32(); r > 0; r--) rows.push_back(decode_tuple_from_key(blob()));
            inputs.emplace_back(FixedRuleInputRelation(rows));
        }
        if (need_device && cz_init(0) != CZ_OK) throw std::runtime_error(cz_last_error());
        FixedRuleRegistry reg = FixedRuleRegistry::with_gpu_defaults();
        try {
            RegularTempStore res = reg.run(name, FixedRulePayload(name, inputs, options, exprs), Poison());
            put_u32(1);
            put_u32((uint32_t)res.size());
            for (const Tuple &t : res) put_blob(encode_key_for_store(0, t, t.size()));
        } catch (const CozoError &e) {
            out.clear();
            put_u32(0);
            put_blob(std::vector<uint8_t>(e.code.begin(), e.code.end()));
        }
    } catch (const std::exception &e) {
        std::printf("run-rule: %s\n", e.what());
        return 2;
    }
    FILE *f = std::fopen(out_path, "wb");
    if (!f) return 3;
    std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    if ((mode == "run-rule" || mode == "run-rule-gpu") && argc == 4) return run_rule_from_file(argv[2], argv[3], mode == "run-rule-gpu");
    test_value_order();
    test_options();
    test_as_directed_graph_vs_oracle();
    test_registry_and_simple_rule();
    test_degree_centrality();
    test_codec_numbers_and_order();
    test_codec_bytes_and_values();
    test_codec_agrees_with_the_python_codec();
    test_codec_rejects_damaged_bytes();
    test_stored_relation_graphs();
    test_stored_rows_delta_cpu();
    test_no_device_fails_loudly();
    if (mode == "rules-cpu") {
        // the binary was linked against tests/cpp/oracle_shim.c ahead of libcozo_gpu.so: the rules' host logic runs on
        // CPU with the oracle standing in for the device entry points (TEST ONLY; HNSW needs the real library)
        gpu_pagerank();
        gpu_love_graph();
        gpu_bfs_cc_dijkstra_random();
        gpu_clustering_coefficients();
        gpu_closeness_centrality();
        gpu_betweenness_centrality();
        gpu_label_propagation();
        gpu_dijkstra_keep_ties();
        gpu_rules_on_stored_relation();
    }
    if (mode == "gpu") {
        if (cz_init(0) != CZ_OK) {
            std::printf("FAIL: cz_init: %s\n", cz_last_error());
            return 2;
        }
        gpu_pagerank();
        gpu_love_graph();
        gpu_bfs_cc_dijkstra_random();
        gpu_clustering_coefficients();
        gpu_closeness_centrality();
        gpu_hnsw_search_ra();
    }
    if (mode == "gpu-stored") {
        if (cz_init(0) != CZ_OK) {
            std::printf("FAIL: cz_init: %s\n", cz_last_error());
            return 2;
        }
        gpu_betweenness_centrality();
        gpu_label_propagation();
        gpu_dijkstra_keep_ties();
        gpu_rules_on_stored_relation();
        gpu_index_through_the_store();
        gpu_index_over_rows_with_several_vectors();
        gpu_index_maintenance_writeback();
    }
    std::printf("%s: %d checks passed, %d failed\n", mode.c_str(), g_pass, g_fail);
    return g_fail ? 1 : 0;
}
