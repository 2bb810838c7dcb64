// fixed_rule.hpp -- C++ host mirror of cozo-core's fixed-rule plugin surface (the reference is compiled code,
// so the host side above the C ABI is compiled code too; the Rust toolchain is absent from this image).
//
// Same names, argument meaning and error behaviour as cozo-core/src/fixed_rule/mod.rs:
//   FixedRule                 trait            :538-567   (init_options / arity / run(payload, out, poison))
//   FixedRulePayload          :47-51, 331-535  option readers: expr/string/integer/pos_integer/non_neg_integer/
//                                              float/unit_interval/bool _option, get_input, inputs_count, name
//   FixedRuleInputRelation    :54-328          arity, ensure_min_len, iter, prefix_iter, as_directed_graph
//                                              (:136-200), as_directed_weighted_graph (:208-328)
//   RegularTempStore          runtime/temp_store.rs:26-29   ordered set of tuples, `put`
//   Poison                    runtime/db.rs:1926-1942       cooperative kill flag (the byte the C ABI polls)
//   Db::register_fixed_rule / unregister_fixed_rule          runtime/db.rs:760-793
//   SimpleFixedRule           :571-688         closure-backed rule over materialised NamedRows
// Errors carry the reference's diagnostic codes (`#[diagnostic(code(...))]`).
#pragma once
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "codec.hpp"
#include "value.hpp"

namespace cozo {

// ---- errors ---------------------------------------------------------------------------------------------------
struct CozoError : std::runtime_error {
    std::string code;
    CozoError(std::string code_, const std::string &msg) : std::runtime_error(msg), code(std::move(code_)) {}
};
struct NotAnEdgeError : CozoError {  // fixed_rule/mod.rs:846-850
    NotAnEdgeError() : CozoError("algo::not_an_edge", "The relation cannot be interpreted as an edge") {}
};
struct BadEdgeWeightError : CozoError {  // :852-860
    explicit BadEdgeWeightError(const DataValue &v)
        : CozoError("algo::invalid_edge_weight",
                    "The value " + v.to_string() + " at the third position in the relation cannot be interpreted as edge weights") {}
};
struct InputRelationArityError : CozoError {  // :68-72
    InputRelationArityError(size_t need, size_t got)
        : CozoError("algo::input_relation_bad_arity", "Input relation to algorithm has insufficient arity: should be at least " +
                                                          std::to_string(need) + " but is " + std::to_string(got)) {}
};
struct FixedRuleOptionNotFoundError : CozoError {  // data/program.rs:291-300
    FixedRuleOptionNotFoundError(const std::string &name, const std::string &rule)
        : CozoError("fixed_rule::arg_not_found", "Cannot find a required named option '" + name + "' for '" + rule + "'") {}
};
struct WrongFixedRuleOptionError : CozoError {  // data/program.rs:301-312
    WrongFixedRuleOptionError(const std::string &name, const std::string &rule, const std::string &help)
        : CozoError("fixed_rule::arg_wrong", "Wrong value for option '" + name + "' of '" + rule + "': " + help) {}
};
struct FixedRuleInputNotFoundError : CozoError {  // FixedRuleNotEnoughRelationError, data/program.rs:339-348
    FixedRuleInputNotFoundError(size_t idx, const std::string &rule)
        : CozoError("fixed_rule::not_enough_args",
                    "Cannot find a required positional argument at index " + std::to_string(idx) + " for '" + rule + "'") {}
};
struct NodeNotFoundError : CozoError {  // fixed_rule/mod.rs:875-885
    explicit NodeNotFoundError(const DataValue &missing)
        : CozoError("algo::node_with_key_not_found", "Required node with key " + missing.to_string() + " not found") {}
};
struct ProcessKilled : CozoError {  // runtime/db.rs:1932-1940
    ProcessKilled() : CozoError("eval::killed", "Running query is killed before completion") {}
};
// a failure reported by libcozo_gpu (no CPU fallback exists: without a device every `run` throws this)
struct GpuError : CozoError {
    int status;
    GpuError(int status_, const std::string &msg) : CozoError("gpu::error", msg), status(status_) {}
};
// throws GpuError / ProcessKilled unless rc == CZ_OK
void check_gpu(int rc);

// ---- Poison ---------------------------------------------------------------------------------------------------
class Poison {
    std::shared_ptr<std::atomic<uint8_t>> flag_ = std::make_shared<std::atomic<uint8_t>>(0);

public:
    void kill() { flag_->store(1, std::memory_order_relaxed); }
    void check() const {
        if (flag_->load(std::memory_order_relaxed)) throw ProcessKilled();
    }
    // what `poison` of the C ABI reads between launches
    const volatile uint8_t *flag_ptr() const { return reinterpret_cast<const volatile uint8_t *>(flag_.get()); }
};

// ---- RegularTempStore -----------------------------------------------------------------------------------------
class RegularTempStore {
    std::set<Tuple, TupleLess> rows_;

public:
    void put(Tuple t) { rows_.insert(std::move(t)); }
    size_t size() const { return rows_.size(); }
    bool empty() const { return rows_.empty(); }
    auto begin() const { return rows_.begin(); }
    auto end() const { return rows_.end(); }
    bool exists(const Tuple &t) const { return rows_.count(t) != 0; }
    std::vector<Tuple> rows() const { return std::vector<Tuple>(rows_.begin(), rows_.end()); }
};

// ---- graphs ---------------------------------------------------------------------------------------------------
// What GraphBuilder::new().csr_layout(CsrLayout::Sorted).edges(..).build() yields (graph_builder 0.4.0, called at
// fixed_rule/mod.rs:192-195, 318-321): both adjacency directions, neighbour lists ascending by dense id (weighted:
// ascending target, input order among equal targets), parallel edges kept.
struct DirectedCsrGraph {
    uint32_t n = 0;
    std::vector<uint32_t> out_offsets, out_targets, in_offsets, in_sources;
    std::vector<float> out_weights;  // empty when unweighted
    uint32_t node_count() const { return n; }
    uint64_t edge_count() const { return out_targets.size(); }
    std::vector<uint32_t> out_degrees() const {
        std::vector<uint32_t> d(n);
        for (uint32_t v = 0; v < n; v++) d[v] = out_offsets[v + 1] - out_offsets[v];
        return d;
    }
    // counting-sort CSR construction (two stable passes = sort by (key, value)); O(E + N)
    static DirectedCsrGraph build(uint32_t n, const std::vector<uint32_t> &from, const std::vector<uint32_t> &to,
                                  const std::vector<float> *weights);
};

struct GraphWithIndices {
    DirectedCsrGraph graph;
    std::vector<DataValue> indices;                                   // dense id -> node value
    std::unordered_map<DataValue, uint32_t, DataValueHash> inv_indices;  // node value -> dense id
};

// ---- FixedRuleInputRelation -----------------------------------------------------------------------------------
// A stored or in-memory relation is a set of tuples scanned in key order; this mirror materialises it sorted.
class FixedRuleInputRelation {
    mutable std::shared_ptr<const std::vector<Tuple>> rows_;  // sorted, de-duplicated (decoded on demand when stored_)
    std::shared_ptr<const StoredRows> stored_;                // from_stored: the graph conversions read the bytes
    std::vector<std::string> bindings_;
    size_t arity_ = 0;
    void ensure_rows() const;

public:
    FixedRuleInputRelation() : rows_(std::make_shared<std::vector<Tuple>>()) {}
    FixedRuleInputRelation(std::vector<Tuple> rows, std::vector<std::string> bindings = {}, std::optional<size_t> arity = {});
    // A relation that lives in the store (MagicFixedRuleRuleArg::Stored, fixed_rule/mod.rs:94-101): the key / value bytes
    // of its scan.  as_directed_graph / as_directed_weighted_graph / as_ordered_graph hand them to libcozo_ingest
    // (include/cozo_ingest.h) -- same ids, same CSR, no tuple per row; iter() / prefix_iter() decode on first use.
    static FixedRuleInputRelation from_stored(StoredRows rows, std::vector<std::string> bindings = {},
                                              std::optional<size_t> arity = {});
    bool is_stored() const { return stored_ != nullptr; }

    size_t arity() const { return arity_; }
    const FixedRuleInputRelation &ensure_min_len(size_t len) const {
        if (arity_ < len) throw InputRelationArityError(len, arity_);
        return *this;
    }
    std::map<std::string, size_t> get_binding_map(size_t offset) const {
        std::map<std::string, size_t> m;
        for (size_t i = 0; i < bindings_.size(); i++) m[bindings_[i]] = i + offset;
        return m;
    }
    const std::vector<Tuple> &iter() const {
        ensure_rows();
        return *rows_;
    }
    // all tuples whose first column equals `prefix`, in key order
    std::pair<std::vector<Tuple>::const_iterator, std::vector<Tuple>::const_iterator> prefix_iter(const DataValue &prefix) const;

    // fixed_rule/mod.rs:136-200: ids in first-appearance order (from before to, row by row); `undirected` mirrors
    // every row after id assignment.  The reference looks every value up in a BTreeMap<DataValue,u32> (O(E log N)
    // comparisons of DataValues); this mirror hashes (O(E)) and builds the CSR by counting sort.
    GraphWithIndices as_directed_graph(bool undirected) const;
    // :208-328: third column -> f32 weight (default 1.0); non-numeric, non-finite or (unless allowed) negative
    // weights are rejected with BadEdgeWeightError
    GraphWithIndices as_directed_weighted_graph(bool undirected, bool allow_negative_weights) const;
    // for the rules that walk `prefix_iter` (ShortestPathBFS, Bfs): neighbours must come in KEY order of the `to`
    // value, so ids are the rank of the value in DataValue order (plus `extra_nodes`: starts / goals without edges)
    GraphWithIndices as_ordered_graph(const std::vector<DataValue> &extra_nodes) const;
};

// ---- FixedRulePayload -----------------------------------------------------------------------------------------
// `options` holds already-evaluated constants (the reference holds Exprs and calls eval_to_const); an
// `expr_option` that must stay an expression (Bfs `condition`) is a predicate over the bound tuple.
using TuplePredicate = std::function<bool(const Tuple &)>;
struct ExprOption {
    TuplePredicate eval;
    bool only_first_binding = false;  // binding_indices() subset of {0}: lets Bfs skip the node lookup (bfs.rs:37-41)
};

class FixedRulePayload {
    std::string name_;
    std::vector<std::optional<FixedRuleInputRelation>> inputs_;
    std::map<std::string, DataValue> options_;
    std::map<std::string, ExprOption> exprs_;

public:
    FixedRulePayload(std::string name, std::vector<std::optional<FixedRuleInputRelation>> inputs,
                     std::map<std::string, DataValue> options = {}, std::map<std::string, ExprOption> exprs = {})
        : name_(std::move(name)), inputs_(std::move(inputs)), options_(std::move(options)), exprs_(std::move(exprs)) {}

    size_t inputs_count() const { return inputs_.size(); }
    const FixedRuleInputRelation &get_input(size_t idx) const {
        if (idx >= inputs_.size() || !inputs_[idx]) throw FixedRuleInputNotFoundError(idx, name_);
        return *inputs_[idx];
    }
    const std::string &name() const { return name_; }
    const std::map<std::string, DataValue> &options() const { return options_; }

    ExprOption expr_option(const std::string &name, std::optional<ExprOption> dflt = {}) const;
    std::string string_option(const std::string &name, std::optional<std::string> dflt = {}) const;
    int64_t integer_option(const std::string &name, std::optional<int64_t> dflt = {}) const;
    size_t pos_integer_option(const std::string &name, std::optional<size_t> dflt = {}) const;
    size_t non_neg_integer_option(const std::string &name, std::optional<size_t> dflt = {}) const;
    double float_option(const std::string &name, std::optional<double> dflt = {}) const;
    double unit_interval_option(const std::string &name, std::optional<double> dflt = {}) const;
    bool bool_option(const std::string &name, std::optional<bool> dflt = {}) const;
};

// ---- FixedRule ------------------------------------------------------------------------------------------------
class FixedRule {
public:
    virtual ~FixedRule() = default;
    virtual void init_options(std::map<std::string, DataValue> & /*options*/) const {}
    virtual size_t arity(const std::map<std::string, DataValue> &options, const std::vector<std::string> &rule_head) const = 0;
    virtual void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const = 0;
};

// SimpleFixedRule (fixed_rule/mod.rs:571-688): every input is materialised as NamedRows, options as constants
struct NamedRows {
    std::vector<std::string> headers;
    std::vector<Tuple> rows;
};
class SimpleFixedRule : public FixedRule {
public:
    using Fn = std::function<NamedRows(const std::vector<NamedRows> &, const std::map<std::string, DataValue> &)>;
    SimpleFixedRule(size_t return_arity, Fn rule) : return_arity_(return_arity), rule_(std::move(rule)) {}
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return return_arity_; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;

private:
    size_t return_arity_;
    Fn rule_;
};

// ---- registry (Db::fixed_rules, runtime/db.rs:103, 273, 760-793) ------------------------------------------------
class FixedRuleRegistry {
    std::unique_ptr<std::mutex> mu_ = std::make_unique<std::mutex>();  // Arc<ShardedLock<..>> in the reference
    std::map<std::string, std::shared_ptr<const FixedRule>> rules_;
    std::set<std::string> builtin_;

public:
    // DEFAULT_FIXED_RULES (fixed_rule/mod.rs:699-836) restricted to the rules on the GPU path, under their own names
    static FixedRuleRegistry with_gpu_defaults();
    // fails if the name exists (db.rs:769-774)
    void register_fixed_rule(const std::string &name, std::shared_ptr<const FixedRule> rule);
    // false if absent; built-ins cannot be removed (db.rs:779-784)
    bool unregister_fixed_rule(const std::string &name);
    std::shared_ptr<const FixedRule> get(const std::string &name) const;
    // what the evaluator does for `?[..] <~ Name(inputs.., options..)`: arity check at "parse time"
    // (parse/query.rs:1020-1031), then run into a fresh store
    RegularTempStore run(const std::string &name, const FixedRulePayload &payload, const Poison &poison,
                         const std::vector<std::string> &rule_head = {}) const;
};

}  // namespace cozo
