// fixed_rule.cpp -- payload / input-relation / registry half of the C++ host mirror (see fixed_rule.hpp).
#include "cozo_host/fixed_rule.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

#include "cozo_gpu.h"
#include "cozo_host/graph_rules.hpp"

namespace cozo {

void check_gpu(int rc) {
    if (rc == CZ_OK) return;
    if (rc == CZ_E_CANCELLED) throw ProcessKilled();
    const char *m = cz_last_error();
    throw GpuError(rc, std::string("libcozo_gpu: ") + (m && *m ? m : "error") + " (status " + std::to_string(rc) + ")");
}

// ---- CSR ------------------------------------------------------------------------------------------------------
namespace {
// one direction: rows = `key`, entries = `val`, sorted by (key, val), ties in input order
void csr_one(uint32_t n, const std::vector<uint32_t> &key, const std::vector<uint32_t> &val, const std::vector<float> *w,
             std::vector<uint32_t> &off, std::vector<uint32_t> &tgt, std::vector<float> *wout) {
    const size_t m = key.size();
    // pass 1: stable counting sort by val
    std::vector<uint32_t> cnt((size_t)n + 1, 0);
    for (size_t i = 0; i < m; i++) cnt[(size_t)val[i] + 1]++;
    for (uint32_t v = 0; v < n; v++) cnt[v + 1] += cnt[v];
    std::vector<uint32_t> by_val(m);
    for (size_t i = 0; i < m; i++) by_val[cnt[val[i]]++] = (uint32_t)i;
    // pass 2: stable counting sort by key
    off.assign((size_t)n + 1, 0);
    for (size_t i = 0; i < m; i++) off[(size_t)key[i] + 1]++;
    for (uint32_t v = 0; v < n; v++) off[v + 1] += off[v];
    std::vector<uint32_t> pos(off.begin(), off.end() - 1);
    tgt.resize(m);
    if (wout) wout->resize(m);
    for (size_t j = 0; j < m; j++) {
        const uint32_t i = by_val[j];
        const uint32_t p = pos[key[i]]++;
        tgt[p] = val[i];
        if (wout) (*wout)[p] = (*w)[i];
    }
}
}  // namespace

DirectedCsrGraph DirectedCsrGraph::build(uint32_t n, const std::vector<uint32_t> &from, const std::vector<uint32_t> &to,
                                         const std::vector<float> *weights) {
    DirectedCsrGraph g;
    g.n = n;
    csr_one(n, from, to, weights, g.out_offsets, g.out_targets, weights ? &g.out_weights : nullptr);
    csr_one(n, to, from, nullptr, g.in_offsets, g.in_sources, nullptr);
    return g;
}

// ---- FixedRuleInputRelation -----------------------------------------------------------------------------------
FixedRuleInputRelation::FixedRuleInputRelation(std::vector<Tuple> rows, std::vector<std::string> bindings,
                                               std::optional<size_t> arity)
    : bindings_(std::move(bindings)) {
    std::sort(rows.begin(), rows.end(), tuple_less);
    rows.erase(std::unique(rows.begin(), rows.end(),
                           [](const Tuple &a, const Tuple &b) { return !tuple_less(a, b) && !tuple_less(b, a); }),
               rows.end());
    arity_ = arity ? *arity : (!rows.empty() ? rows[0].size() : bindings_.size());
    rows_ = std::make_shared<const std::vector<Tuple>>(std::move(rows));
}

FixedRuleInputRelation FixedRuleInputRelation::from_stored(StoredRows rows, std::vector<std::string> bindings,
                                                           std::optional<size_t> arity) {
    FixedRuleInputRelation r;
    r.rows_.reset();
    r.bindings_ = std::move(bindings);
    r.arity_ = arity ? *arity : (rows.size() ? rows.tuple(0).size() : r.bindings_.size());
    r.stored_ = std::make_shared<const StoredRows>(std::move(rows));
    return r;
}

void FixedRuleInputRelation::ensure_rows() const {
    if (rows_) return;
    std::vector<Tuple> rows;
    rows.reserve(stored_->size());
    for (size_t i = 0; i < stored_->size(); i++) rows.push_back(stored_->tuple(i));  // key-byte order == tuple order
    rows_ = std::make_shared<const std::vector<Tuple>>(std::move(rows));
}

namespace {
// the ingest route: bytes -> ids + CSR (include/cozo_ingest.h); returns false when the relation holds a bad weight, so
// that the caller can take the tuple route and report the offending value like the reference does
bool graph_from_stored(const StoredRows &stored, uint32_t flags, GraphWithIndices &r) {
    const czi_rows view = stored.view();
    czi_graph *g = nullptr;
    const int rc = czi_graph_ingest(&view, flags, &g);
    if (rc == CZI_E_NOT_AN_EDGE) throw NotAnEdgeError();
    if (rc == CZI_E_BAD_WEIGHT) return false;
    if (rc != CZI_OK) throw CozoError("ingest::error", std::string("libcozo_ingest: ") + czi_last_error());
    std::unique_ptr<czi_graph, void (*)(czi_graph *)> hold(g, czi_graph_free);
    const uint32_t n = czi_graph_node_count(g);
    const uint64_t e = czi_graph_edge_count(g);
    DirectedCsrGraph &d = r.graph;
    d.n = n;
    d.out_offsets.resize((size_t)n + 1);
    d.out_targets.resize(e);
    d.in_offsets.resize((size_t)n + 1);
    d.in_sources.resize(e);
    if (flags & CZI_WEIGHTED) d.out_weights.resize(e);
    if (czi_graph_csr(g, 0, d.out_offsets.data(), d.out_targets.data(), (flags & CZI_WEIGHTED) ? d.out_weights.data() : nullptr) ||
        czi_graph_csr(g, 1, d.in_offsets.data(), d.in_sources.data(), nullptr))
        throw CozoError("ingest::error", std::string("libcozo_ingest: ") + czi_last_error());
    const uint8_t *bytes = nullptr;
    const uint64_t *off = nullptr;
    czi_graph_node_keys(g, &bytes, &off);
    r.indices.reserve(n);
    r.inv_indices.reserve(n);
    for (uint32_t i = 0; i < n; i++) {  // N decodes, not 2E
        const uint8_t *p = bytes + off[i];
        r.indices.push_back(decode_datavalue(p, bytes + off[i + 1]));
        r.inv_indices.emplace(r.indices.back(), i);
    }
    return true;
}
}  // namespace

std::pair<std::vector<Tuple>::const_iterator, std::vector<Tuple>::const_iterator>
FixedRuleInputRelation::prefix_iter(const DataValue &prefix) const {
    ensure_rows();
    auto lo = std::lower_bound(rows_->begin(), rows_->end(), prefix, [](const Tuple &t, const DataValue &p) {
        return !t.empty() && DataValue::compare(t[0], p) < 0;
    });
    auto hi = lo;
    while (hi != rows_->end() && !hi->empty() && (*hi)[0] == prefix) ++hi;
    return {lo, hi};
}

namespace {
struct IdAssigner {
    std::vector<DataValue> indices;
    std::unordered_map<DataValue, uint32_t, DataValueHash> inv;
    uint32_t id(const DataValue &v) {
        auto it = inv.find(v);
        if (it != inv.end()) return it->second;
        const uint32_t i = (uint32_t)indices.size();
        inv.emplace(v, i);
        indices.push_back(v);
        return i;
    }
};
}  // namespace

namespace {
// Fast path of the id assignment for the common relation whose two node columns are integers: a flat open-addressing
// table (i64 key -> dense id) instead of a node-based map of DataValues.  Same first-appearance order.
struct IntIdAssigner {
    std::vector<int64_t> keys;   // slot -> key
    std::vector<uint32_t> vals;  // slot -> id, UINT32_MAX = empty
    std::vector<int64_t> order;  // id -> key
    uint64_t mask = 0;
    explicit IntIdAssigner(size_t expected) {
        size_t cap = 64;
        while (cap < expected * 2 + 16) cap <<= 1;
        keys.resize(cap);
        vals.assign(cap, UINT32_MAX);
        mask = cap - 1;
    }
    static uint64_t mix(uint64_t x) {
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdull;
        x ^= x >> 33;
        x *= 0xc4ceb9fe1a85ec53ull;
        return x ^ (x >> 33);
    }
    void grow() {
        std::vector<int64_t> ok;
        std::vector<uint32_t> ov;
        ok.swap(keys);
        ov.swap(vals);
        keys.resize(ok.size() * 2);
        vals.assign(ok.size() * 2, UINT32_MAX);
        mask = keys.size() - 1;
        for (size_t i = 0; i < ok.size(); i++)
            if (ov[i] != UINT32_MAX) {
                uint64_t h = mix((uint64_t)ok[i]) & mask;
                while (vals[h] != UINT32_MAX) h = (h + 1) & mask;
                keys[h] = ok[i];
                vals[h] = ov[i];
            }
    }
    uint32_t id(int64_t k) {
        uint64_t h = mix((uint64_t)k) & mask;
        while (vals[h] != UINT32_MAX) {
            if (keys[h] == k) return vals[h];
            h = (h + 1) & mask;
        }
        const uint32_t i = (uint32_t)order.size();
        keys[h] = k;
        vals[h] = i;
        order.push_back(k);
        if (order.size() * 2 > keys.size()) grow();
        return i;
    }
};

// true when the first two columns of every row are Int (not Float: 1 and 1.0 are different DataValues)
bool int_keyed(const std::vector<Tuple> &rows) {
    for (const Tuple &t : rows)
        if (t.size() < 2 || !t[0].is_int() || !t[1].is_int()) return false;
    return true;
}

void finish_int_ids(IntIdAssigner &ids, GraphWithIndices &r) {
    r.indices.reserve(ids.order.size());
    r.inv_indices.reserve(ids.order.size());
    for (uint32_t i = 0; i < ids.order.size(); i++) {
        r.indices.emplace_back(ids.order[i]);
        r.inv_indices.emplace(r.indices.back(), i);
    }
}
}  // namespace

GraphWithIndices FixedRuleInputRelation::as_directed_graph(bool undirected) const {
    if (stored_) {
        GraphWithIndices r;
        graph_from_stored(*stored_, undirected ? CZI_UNDIRECTED : 0u, r);
        return r;
    }
    if (!rows_->empty() && int_keyed(*rows_)) {
        IntIdAssigner ids(rows_->size());
        std::vector<uint32_t> from, to;
        from.reserve(rows_->size() * (undirected ? 2 : 1));
        to.reserve(rows_->size() * (undirected ? 2 : 1));
        for (const Tuple &t : *rows_) {
            const uint32_t f = ids.id(std::get<int64_t>(t[0].r));
            const uint32_t d = ids.id(std::get<int64_t>(t[1].r));
            from.push_back(f);
            to.push_back(d);
            if (undirected) {
                from.push_back(d);
                to.push_back(f);
            }
        }
        GraphWithIndices r;
        r.graph = DirectedCsrGraph::build((uint32_t)ids.order.size(), from, to, nullptr);
        finish_int_ids(ids, r);
        return r;
    }
    IdAssigner ids;
    ids.inv.reserve(rows_->size());
    std::vector<uint32_t> from, to;
    from.reserve(rows_->size() * (undirected ? 2 : 1));
    to.reserve(rows_->size() * (undirected ? 2 : 1));
    for (const Tuple &t : *rows_) {
        if (t.size() < 2) throw NotAnEdgeError();
        const uint32_t f = ids.id(t[0]);
        const uint32_t d = ids.id(t[1]);
        from.push_back(f);
        to.push_back(d);
        if (undirected) {
            from.push_back(d);
            to.push_back(f);
        }
    }
    GraphWithIndices r;
    r.graph = DirectedCsrGraph::build((uint32_t)ids.indices.size(), from, to, nullptr);
    r.indices = std::move(ids.indices);
    r.inv_indices = std::move(ids.inv);
    return r;
}

GraphWithIndices FixedRuleInputRelation::as_directed_weighted_graph(bool undirected, bool allow_negative_weights) const {
    if (stored_) {
        GraphWithIndices r;
        if (graph_from_stored(*stored_, CZI_WEIGHTED | (undirected ? CZI_UNDIRECTED : 0u) |
                                            (allow_negative_weights ? CZI_ALLOW_NEGATIVE_WEIGHTS : 0u), r))
            return r;
        ensure_rows();  // a bad weight: the tuple route below throws BadEdgeWeightError with the value
    }
    IdAssigner ids;
    ids.inv.reserve(rows_->size());
    std::vector<uint32_t> from, to;
    std::vector<float> w;
    for (const Tuple &t : *rows_) {
        if (t.size() < 2) throw NotAnEdgeError();
        const uint32_t f = ids.id(t[0]);
        const uint32_t d = ids.id(t[1]);
        float weight = 1.0f;
        if (t.size() >= 3) {
            double x;
            if (!t[2].get_float(&x)) throw BadEdgeWeightError(t[2]);
            if (!std::isfinite(x)) throw BadEdgeWeightError(t[2]);
            if (x < 0.0 && !allow_negative_weights) throw BadEdgeWeightError(t[2]);
            weight = (float)x;
        }
        from.push_back(f);
        to.push_back(d);
        w.push_back(weight);
        if (undirected) {
            from.push_back(d);
            to.push_back(f);
            w.push_back(weight);
        }
    }
    GraphWithIndices r;
    r.graph = DirectedCsrGraph::build((uint32_t)ids.indices.size(), from, to, &w);
    r.indices = std::move(ids.indices);
    r.inv_indices = std::move(ids.inv);
    return r;
}

GraphWithIndices FixedRuleInputRelation::as_ordered_graph(const std::vector<DataValue> &extra_nodes) const {
    if (stored_) {
        GraphWithIndices r;
        graph_from_stored(*stored_, CZI_ORDERED_IDS, r);
        bool all_present = true;
        for (const DataValue &v : extra_nodes) all_present &= r.inv_indices.count(v) != 0;
        if (all_present) return r;
        ensure_rows();  // a start / goal without an edge needs an id of its own: the tuple route
    }
    std::vector<DataValue> vals;
    vals.reserve(rows_->size() * 2 + extra_nodes.size());
    for (const Tuple &t : *rows_) {
        if (t.size() < 2) throw NotAnEdgeError();
        vals.push_back(t[0]);
        vals.push_back(t[1]);
    }
    for (const DataValue &v : extra_nodes) vals.push_back(v);
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
    GraphWithIndices r;
    r.inv_indices.reserve(vals.size());
    for (uint32_t i = 0; i < vals.size(); i++) r.inv_indices.emplace(vals[i], i);
    std::vector<uint32_t> from, to;
    from.reserve(rows_->size());
    to.reserve(rows_->size());
    for (const Tuple &t : *rows_) {
        from.push_back(r.inv_indices.at(t[0]));
        to.push_back(r.inv_indices.at(t[1]));
    }
    r.graph = DirectedCsrGraph::build((uint32_t)vals.size(), from, to, nullptr);
    r.indices = std::move(vals);
    return r;
}

// ---- FixedRulePayload -----------------------------------------------------------------------------------------
ExprOption FixedRulePayload::expr_option(const std::string &name, std::optional<ExprOption> dflt) const {
    auto it = exprs_.find(name);
    if (it != exprs_.end()) return it->second;
    if (dflt) return *dflt;
    throw FixedRuleOptionNotFoundError(name, name_);
}

std::string FixedRulePayload::string_option(const std::string &name, std::optional<std::string> dflt) const {
    auto it = options_.find(name);
    if (it != options_.end()) {
        if (const std::string *s = it->second.get_str()) return *s;
        throw WrongFixedRuleOptionError(name, name_, "a string is required");
    }
    if (dflt) return *dflt;
    throw FixedRuleOptionNotFoundError(name, name_);
}

int64_t FixedRulePayload::integer_option(const std::string &name, std::optional<int64_t> dflt) const {
    auto it = options_.find(name);
    if (it != options_.end()) {
        if (!it->second.is_num()) throw WrongFixedRuleOptionError(name, name_, "an integer is required");
        int64_t i;
        // sic: a non-integral number is reported as "not found" (fixed_rule/mod.rs:414-421)
        if (!it->second.get_int(&i)) throw FixedRuleOptionNotFoundError(name, name_);
        return i;
    }
    if (dflt) return *dflt;
    throw FixedRuleOptionNotFoundError(name, name_);
}

size_t FixedRulePayload::pos_integer_option(const std::string &name, std::optional<size_t> dflt) const {
    const int64_t i = integer_option(name, dflt ? std::optional<int64_t>((int64_t)*dflt) : std::nullopt);
    if (i <= 0) {
        // the error needs option_span(name), which itself fails when the offending value was the default (:443-455)
        if (!options_.count(name)) throw FixedRuleOptionNotFoundError(name, name_);
        throw WrongFixedRuleOptionError(name, name_, "a positive integer is required");
    }
    return (size_t)i;
}

size_t FixedRulePayload::non_neg_integer_option(const std::string &name, std::optional<size_t> dflt) const {
    const int64_t i = integer_option(name, dflt ? std::optional<int64_t>((int64_t)*dflt) : std::nullopt);
    if (i < 0) {
        if (!options_.count(name)) throw FixedRuleOptionNotFoundError(name, name_);
        throw WrongFixedRuleOptionError(name, name_, "a non-negative integer is required");
    }
    return (size_t)i;
}

double FixedRulePayload::float_option(const std::string &name, std::optional<double> dflt) const {
    auto it = options_.find(name);
    if (it != options_.end()) {
        double f;
        if (!it->second.get_float(&f)) throw WrongFixedRuleOptionError(name, name_, "a floating number is required");
        return f;
    }
    if (dflt) return *dflt;
    throw FixedRuleOptionNotFoundError(name, name_);
}

double FixedRulePayload::unit_interval_option(const std::string &name, std::optional<double> dflt) const {
    const double f = float_option(name, dflt);
    if (!(f >= 0.0 && f <= 1.0)) {
        if (!options_.count(name)) throw FixedRuleOptionNotFoundError(name, name_);
        throw WrongFixedRuleOptionError(name, name_, "a number between 0. and 1. is required");
    }
    return f;
}

bool FixedRulePayload::bool_option(const std::string &name, std::optional<bool> dflt) const {
    auto it = options_.find(name);
    if (it != options_.end()) {
        bool b;
        if (!it->second.get_bool(&b)) throw WrongFixedRuleOptionError(name, name_, "a boolean value is required");
        return b;
    }
    if (dflt) return *dflt;
    throw FixedRuleOptionNotFoundError(name, name_);
}

// ---- SimpleFixedRule ------------------------------------------------------------------------------------------
void SimpleFixedRule::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    std::vector<NamedRows> inputs;
    for (size_t i = 0; i < payload.inputs_count(); i++) {
        const FixedRuleInputRelation &rel = payload.get_input(i);
        NamedRows nr;
        nr.headers.resize(rel.arity());
        for (const auto &kv : rel.get_binding_map(0))
            if (kv.second < nr.headers.size()) nr.headers[kv.second] = kv.first;
        nr.rows = rel.iter();
        inputs.push_back(std::move(nr));
    }
    poison.check();
    NamedRows res = rule_(inputs, payload.options());
    for (Tuple &row : res.rows) {
        if (row.size() != return_arity_)  // fixed_rule/mod.rs:676-683
            throw CozoError("fixed_rule::simple::bad_arity", "arity mismatch: fixed rule returned a row of length " +
                                                                std::to_string(row.size()) + " instead of " +
                                                                std::to_string(return_arity_));
        out.put(std::move(row));
    }
}

// ---- registry -------------------------------------------------------------------------------------------------
FixedRuleRegistry FixedRuleRegistry::with_gpu_defaults() {
    FixedRuleRegistry r;
    auto add = [&](const char *name, std::shared_ptr<const FixedRule> impl) {
        r.rules_[name] = std::move(impl);
        r.builtin_.insert(name);
    };
    add("PageRank", std::make_shared<PageRank>());
    add("ShortestPathBFS", std::make_shared<ShortestPathBFS>());
    add("BreadthFirstSearch", std::make_shared<Bfs>());
    add("BFS", std::make_shared<Bfs>());
    add("ConnectedComponents", std::make_shared<StronglyConnectedComponent>(false));
    add("StronglyConnectedComponents", std::make_shared<StronglyConnectedComponent>(true));
    add("SCC", std::make_shared<StronglyConnectedComponent>(true));
    add("ShortestPathDijkstra", std::make_shared<ShortestPathDijkstra>());
    add("ClusteringCoefficients", std::make_shared<ClusteringCoefficients>());
    add("DegreeCentrality", std::make_shared<DegreeCentrality>());
    add("BetweennessCentrality", std::make_shared<BetweennessCentrality>());
    add("ClosenessCentrality", std::make_shared<ClosenessCentrality>());
    add("LabelPropagation", std::make_shared<LabelPropagation>());
    return r;
}

void FixedRuleRegistry::register_fixed_rule(const std::string &name, std::shared_ptr<const FixedRule> rule) {
    std::lock_guard<std::mutex> g(*mu_);
    if (rules_.count(name)) throw CozoError("", "A fixed rule with the name " + name + " is already registered");
    rules_[name] = std::move(rule);
}

bool FixedRuleRegistry::unregister_fixed_rule(const std::string &name) {
    std::lock_guard<std::mutex> g(*mu_);
    if (builtin_.count(name)) throw CozoError("", "Cannot unregister builtin fixed rule " + name);
    return rules_.erase(name) != 0;
}

std::shared_ptr<const FixedRule> FixedRuleRegistry::get(const std::string &name) const {
    std::lock_guard<std::mutex> g(*mu_);
    auto it = rules_.find(name);
    if (it == rules_.end())  // FixedRuleNotFoundError, parse/query.rs:1010-1019
        throw CozoError("parser::fixed_rule_not_found", "The fixed rule '" + name + "' is not found");
    return it->second;
}

RegularTempStore FixedRuleRegistry::run(const std::string &name, const FixedRulePayload &payload, const Poison &poison,
                                        const std::vector<std::string> &rule_head) const {
    std::shared_ptr<const FixedRule> impl = get(name);
    std::map<std::string, DataValue> opts = payload.options();
    impl->init_options(opts);
    const size_t arity = impl->arity(opts, rule_head);
    if (!rule_head.empty() && rule_head.size() != arity)  // FixedRuleHeadArityMismatch, parse/query.rs:1022-1031
        throw CozoError("parser::fixed_rule_head_arity_mismatch",
                        "Fixed rule head arity mismatch: expected " + std::to_string(arity) + ", found " + std::to_string(rule_head.size()));
    RegularTempStore out;
    impl->run(payload, out, poison);
    for (const Tuple &t : out)
        if (t.size() != arity) throw CozoError("fixed_rule::bad_arity", "rule '" + name + "' produced a row of the wrong arity");
    return out;
}

}  // namespace cozo
