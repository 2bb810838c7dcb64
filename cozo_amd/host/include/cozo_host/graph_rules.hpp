// graph_rules.hpp -- the whole-graph fixed rules of the GPU path as `impl FixedRule` (C++ host mirror).
//
// Each rule reads its options and inputs exactly like the reference's rule of the same name, maps node values to
// dense ids on the host (as the reference does), hands a CSR to libcozo_gpu (include/cozo_gpu.h) and writes the
// reference's rows to `out`.  There is no CPU fallback: without the device library / a gfx950 device `run` throws.
//   PageRank                     fixed_rule/algos/pagerank.rs:29-56                    -> cz_pagerank
//   ShortestPathBFS              fixed_rule/algos/shortest_path_bfs.rs:35-113          -> cz_bfs
//   Bfs                          fixed_rule/algos/bfs.rs:25-113                        -> cz_bfs (share_visited)
//   StronglyConnectedComponent   fixed_rule/algos/strongly_connected_components.rs:42-77 (strong = false only)
//                                                                                      -> cz_connected_components
//   ShortestPathDijkstra         fixed_rule/algos/shortest_path_dijkstra.rs:33-153     -> cz_sssp
//   ClusteringCoefficients       fixed_rule/algos/triangles.rs:25-110                  -> cz_clustering_coefficients
//   DegreeCentrality             fixed_rule/algos/degree_centrality.rs:24-76           (a scan with counters: host only)
//   ClosenessCentrality          fixed_rule/algos/all_pairs_shortest_path.rs:97-176    -> cz_closeness
//   BetweennessCentrality        fixed_rule/algos/all_pairs_shortest_path.rs:31-95     -> cz_betweenness
//   LabelPropagation             fixed_rule/algos/label_propagation.rs:27-109          -> cz_label_propagation (one fixed execution)
//                                                                                      accumulation on the tight-edge DAG (host)
#pragma once
#include "fixed_rule.hpp"

namespace cozo {

class PageRank : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 2; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class ShortestPathBFS : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 3; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class Bfs : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 3; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class StronglyConnectedComponent : public FixedRule {
    bool strong_;

public:
    explicit StronglyConnectedComponent(bool strong) : strong_(strong) {}
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 2; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class ShortestPathDijkstra : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 4; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class ClusteringCoefficients : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 4; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class ClosenessCentrality : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 2; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class BetweennessCentrality : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 2; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

// algos/label_propagation.rs:27-109 as ONE fixed execution of the reference's loop (it shuffles the node order in every iteration
// and breaks ties with thread_rng): colour classes of a deterministic colouring in ascending order, the smallest label on ties
// (include/cozo_gpu.h cz_label_propagation).  Options: undirected (false), max_iter (10).  Rows: (label as i64, node).
class LabelPropagation : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 2; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

class DegreeCentrality : public FixedRule {
public:
    size_t arity(const std::map<std::string, DataValue> &, const std::vector<std::string> &) const override { return 4; }
    void run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const override;
};

}  // namespace cozo
