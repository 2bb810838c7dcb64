// graph_rules.cpp -- the GPU forms of the whole-graph fixed rules (see graph_rules.hpp for the reference map).
#include "cozo_host/graph_rules.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_set>

#include "cozo_gpu.h"

namespace cozo {

namespace {

// walk the backtrace from `goal` to `start` (shortest_path_bfs.rs:85-92, bfs.rs:97-106)
std::vector<uint32_t> walk_back(const uint32_t *parent, uint32_t start, uint32_t goal) {
    std::vector<uint32_t> route;
    uint32_t cur = goal;
    while (cur != start) {
        route.push_back(cur);
        cur = parent[cur];
        if (cur == CZ_NONE) throw CozoError("gpu::broken_backtrace", "broken backtrace returned by cz_bfs / cz_sssp");
    }
    route.push_back(start);
    std::reverse(route.begin(), route.end());
    return route;
}

DataValue path_value(const std::vector<uint32_t> &route, const std::vector<DataValue> &indices) {
    std::vector<DataValue> items;
    items.reserve(route.size());
    for (uint32_t u : route) items.push_back(indices[u]);
    return DataValue::list(std::move(items));
}

}  // namespace

// ---- PageRank -------------------------------------------------------------------------------------------------
void PageRank::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    const bool undirected = payload.bool_option("undirected", false);
    const float theta = (float)payload.unit_interval_option("theta", 0.85);
    const float epsilon = (float)payload.unit_interval_option("epsilon", 0.0001);  // narrowed to f32 (:38), widened at :49
    const size_t iterations = payload.pos_integer_option("iterations", 10);
    // an option of the GPU rule alone: graph::page_rank under the reading that refreshes a contribution inside the sweep (the
    // reference's one-thread execution if the crate does that; cz_pagerank_inplace)
    const bool in_place = payload.bool_option("in_place", false);
    GraphWithIndices g = edges.as_directed_graph(undirected);
    if (g.indices.empty()) return;  // :43-45
    const DirectedCsrGraph &gr = g.graph;
    const std::vector<uint32_t> out_degree = gr.out_degrees();
    std::vector<float> scores(gr.n);
    uint32_t iters_run = 0;
    double final_err = 0.0;
    if (in_place)
        check_gpu(cz_pagerank_inplace(gr.in_offsets.data(), gr.in_sources.data(), out_degree.data(), gr.n, gr.edge_count(), theta,
                                      (double)epsilon, (uint32_t)iterations, 0, scores.data(), &iters_run, &final_err, nullptr,
                                      poison.flag_ptr()));
    else
        check_gpu(cz_pagerank(gr.in_offsets.data(), gr.in_sources.data(), out_degree.data(), gr.n, gr.edge_count(), theta,
                              (double)epsilon, (uint32_t)iterations, scores.data(), &iters_run, &final_err, poison.flag_ptr()));
    for (uint32_t i = 0; i < gr.n; i++) out.put(Tuple{g.indices[i], DataValue((double)scores[i])});
}

// ---- ShortestPathBFS ------------------------------------------------------------------------------------------
void ShortestPathBFS::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0).ensure_min_len(2);
    const FixedRuleInputRelation &starting = payload.get_input(1).ensure_min_len(1);
    const FixedRuleInputRelation &ending = payload.get_input(2).ensure_min_len(1);
    std::vector<DataValue> starting_nodes;
    for (const Tuple &t : starting.iter()) starting_nodes.push_back(t[0]);
    std::vector<DataValue> ending_nodes;  // BTreeSet<DataValue> (:53-57): sorted, unique
    for (const Tuple &t : ending.iter()) ending_nodes.push_back(t[0]);
    std::sort(ending_nodes.begin(), ending_nodes.end());
    ending_nodes.erase(std::unique(ending_nodes.begin(), ending_nodes.end()), ending_nodes.end());
    if (starting_nodes.empty() || ending_nodes.empty()) return;
    std::vector<DataValue> extra(starting_nodes);
    extra.insert(extra.end(), ending_nodes.begin(), ending_nodes.end());
    GraphWithIndices g = edges.as_ordered_graph(extra);
    const DirectedCsrGraph &gr = g.graph;
    std::vector<uint32_t> starts, goals;
    for (const DataValue &s : starting_nodes) starts.push_back(g.inv_indices.at(s));
    for (const DataValue &e : ending_nodes) goals.push_back(g.inv_indices.at(e));
    std::vector<uint32_t> parent((size_t)starts.size() * gr.n);
    check_gpu(cz_bfs(gr.out_offsets.data(), gr.out_targets.data(), gr.n, gr.edge_count(), starts.data(), (uint32_t)starts.size(),
                     goals.data(), (uint32_t)goals.size(), 0, parent.data(), nullptr, nullptr, nullptr, poison.flag_ptr()));
    for (size_t si = 0; si < starts.size(); si++) {
        const uint32_t *par = parent.data() + si * gr.n;
        for (size_t gi = 0; gi < goals.size(); gi++) {
            // a goal equal to the start is never "discovered" (:66-84 only looks at neighbours): Null, like the reference
            if (par[goals[gi]] != CZ_NONE)
                out.put(Tuple{starting_nodes[si], ending_nodes[gi], path_value(walk_back(par, starts[si], goals[gi]), g.indices)});
            else
                out.put(Tuple{starting_nodes[si], ending_nodes[gi], DataValue()});
        }
        poison.check();
    }
}

// ---- Bfs ------------------------------------------------------------------------------------------------------
void Bfs::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0).ensure_min_len(2);
    const FixedRuleInputRelation &nodes = payload.get_input(1);
    const FixedRuleInputRelation *starting_rel = &nodes;  // `payload.get_input(2).unwrap_or(nodes)` (:33)
    if (payload.inputs_count() > 2) {
        try {
            starting_rel = &payload.get_input(2);
        } catch (const FixedRuleInputNotFoundError &) {
        }
    }
    const size_t limit = payload.pos_integer_option("limit", 1);
    const ExprOption condition = payload.expr_option("condition");
    const bool skip_query_nodes = condition.only_first_binding;
    std::vector<DataValue> start_vals;
    for (const Tuple &t : starting_rel->iter()) start_vals.push_back(t[0]);
    if (start_vals.empty()) return;
    GraphWithIndices g = edges.as_ordered_graph(start_vals);
    const DirectedCsrGraph &gr = g.graph;
    std::vector<uint32_t> starts;
    for (const DataValue &s : start_vals) starts.push_back(g.inv_indices.at(s));
    const size_t ns = starts.size();
    // one backtrace and one discovery sequence for all starts (cz_bfs_shared_until): the default is EVERY node as a start (:33), for
    // which a row of N per start would be O(N^2).  The condition is evaluated level by level as the device discovers the nodes; the
    // traversal ends with the level in which the `limit`-th node passed (:88-91 `break 'outer`).
    std::vector<uint32_t> parent(gr.n), order(gr.n), first(ns + 1);
    struct Found {
        uint32_t start, end;
    };
    struct Ctx {
        const FixedRuleInputRelation &nodes;
        const GraphWithIndices &g;
        const ExprOption &condition;
        const Poison &poison;
        bool skip_query_nodes;
        size_t limit;
        std::vector<Found> found;
        bool missing = false;
        uint32_t missing_node = 0;
        std::exception_ptr raised;  // (nothing may unwind through the library's frames)
    } ctx{nodes, g, condition, poison, skip_query_nodes, limit, {}, false, 0, nullptr};
    auto on_level = [](void *vp, uint32_t start, const uint32_t *level, uint32_t n) -> int {
        Ctx &c = *static_cast<Ctx *>(vp);
        try {
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t to = level[j];
                const DataValue &to_val = c.g.indices[to];
                Tuple cand_tuple;
                if (c.skip_query_nodes) {
                    cand_tuple = Tuple{to_val};
                } else {
                    auto range = c.nodes.prefix_iter(to_val);
                    if (range.first == range.second) {
                        c.missing = true;
                        c.missing_node = to;
                        return 1;
                    }
                    cand_tuple = *range.first;
                }
                if (c.condition.eval(cand_tuple)) {
                    c.found.push_back({start, to});
                    if (c.found.size() >= c.limit) return 1;
                }
                c.poison.check();
            }
        } catch (...) {
            c.raised = std::current_exception();
            return 1;
        }
        return 0;
    };
    check_gpu(cz_bfs_shared_until(gr.out_offsets.data(), gr.out_targets.data(), gr.n, gr.edge_count(), starts.data(), (uint32_t)ns, on_level,
                                  &ctx, parent.data(), order.data(), first.data(), poison.flag_ptr()));
    if (ctx.raised) std::rethrow_exception(ctx.raised);
    if (ctx.missing)  // sic: the reference reports the *discoverer* as missing (:74-77)
        throw NodeNotFoundError(g.indices[parent[ctx.missing_node]]);
    const std::vector<Found> &found = ctx.found;
    // the backtrace is shared across starts (:44); every node has exactly one discoverer
    for (const Found &f : found)
        out.put(Tuple{g.indices[f.start], g.indices[f.end], path_value(walk_back(parent.data(), f.start, f.end), g.indices)});
}

// ---- ConnectedComponents / SCC --------------------------------------------------------------------------------
void StronglyConnectedComponent::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    if (strong_)  // Tarjan's DFS numbering is sequential and its group ids are DFS-order dependent: not on the GPU path
        throw GpuError(CZ_E_UNSUPPORTED, "StronglyConnectedComponents (strong = true) is not available on the GPU path");
    const FixedRuleInputRelation &edges = payload.get_input(0);
    GraphWithIndices g = edges.as_directed_graph(/*undirected=*/true);  // `!self.strong` (:49)
    uint32_t n_groups = 0;
    if (!g.indices.empty()) {
        const DirectedCsrGraph &gr = g.graph;
        std::vector<uint32_t> group(gr.n);
        check_gpu(cz_connected_components(gr.out_offsets.data(), gr.out_targets.data(), gr.n, gr.edge_count(), group.data(),
                                          &n_groups, poison.flag_ptr()));
        for (uint32_t i = 0; i < gr.n; i++) out.put(Tuple{g.indices[i], DataValue((int64_t)group[i])});
    }
    int64_t counter = n_groups;  // :61-74 nodes that appear only in the optional node relation
    if (payload.inputs_count() > 1) {
        const FixedRuleInputRelation *nodes = nullptr;
        try {
            nodes = &payload.get_input(1);
        } catch (const FixedRuleInputNotFoundError &) {
        }
        if (nodes) {
            for (const Tuple &t : nodes->iter()) {
                if (t.empty()) continue;
                if (!g.inv_indices.count(t[0])) {
                    g.inv_indices.emplace(t[0], (uint32_t)g.inv_indices.size());
                    out.put(Tuple{t[0], DataValue(counter)});
                    counter++;
                }
            }
        }
    }
}

// ---- ShortestPathDijkstra -------------------------------------------------------------------------------------
void ShortestPathDijkstra::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    const FixedRuleInputRelation &starting = payload.get_input(1);
    const FixedRuleInputRelation *termination = nullptr;
    if (payload.inputs_count() > 2) {
        try {
            termination = &payload.get_input(2);
        } catch (const FixedRuleInputNotFoundError &) {
        }
    }
    const bool undirected = payload.bool_option("undirected", false);
    const bool keep_ties = payload.bool_option("keep_ties", false);
    GraphWithIndices g = edges.as_directed_weighted_graph(undirected, false);
    const DirectedCsrGraph &gr = g.graph;
    auto id_set = [&](const FixedRuleInputRelation &rel) {
        std::vector<uint32_t> ids;  // BTreeSet<u32>
        for (const Tuple &t : rel.iter()) {
            if (t.empty()) continue;
            auto it = g.inv_indices.find(t[0]);
            if (it != g.inv_indices.end()) ids.push_back(it->second);
        }
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        return ids;
    };
    const std::vector<uint32_t> starting_nodes = id_set(starting);
    std::vector<uint32_t> termination_nodes;
    if (termination) termination_nodes = id_set(*termination);
    if (starting_nodes.empty()) return;
    if (termination && termination_nodes.empty()) return;
    // keep_ties only takes effect with a termination relation (:73-86: without one the reference runs the plain dijkstra)
    const bool ties = keep_ties && termination;
    if (ties)
        for (float w : gr.out_weights)
            if (!(w > 0.0f)) throw CozoError("algo::keep_ties_needs_positive_weights", "keep_ties on the GPU path needs positive edge weights");
    const size_t ns = starting_nodes.size();
    std::vector<float> dist(ns * gr.n);
    std::vector<uint32_t> parent(ns * gr.n);
    // with a termination relation the search stops once every target is settled, like dijkstra()'s goal set (:300-306)
    check_gpu(cz_sssp_goals(gr.out_offsets.data(), gr.out_targets.data(), gr.out_weights.data(), gr.n, gr.edge_count(),
                            starting_nodes.data(), (uint32_t)ns, termination ? termination_nodes.data() : nullptr,
                            termination ? (uint32_t)termination_nodes.size() : 0u, dist.data(), parent.data(), poison.flag_ptr()));
    for (size_t si = 0; si < ns && ties; si++) {
        // dijkstra_keep_ties (:341-450): back_pointers[v] = every edge (u, v) with dist[u] + w == dist[v] in f32 -- read off the
        // device's bit-exact distances -- and EVERY path through them is a row; the start as its own target collects nothing
        const uint32_t s = starting_nodes[si];
        const float *d = dist.data() + si * gr.n;
        std::vector<std::vector<uint32_t>> preds(gr.n);
        for (uint32_t u = 0; u < gr.n; u++) {
            if (!std::isfinite(d[u])) continue;
            for (uint32_t e = gr.out_offsets[u]; e < gr.out_offsets[u + 1]; e++)
                if ((float)(d[u] + gr.out_weights[e]) == d[gr.out_targets[e]]) preds[gr.out_targets[e]].push_back(u);
        }
        for (uint32_t t : termination_nodes) {
            if (!std::isfinite(d[t])) {
                out.put(Tuple{g.indices[s], g.indices[t], DataValue((double)d[t]), DataValue::list({})});
                continue;
            }
            std::vector<std::vector<uint32_t>> stack{{t}};
            size_t emitted = 0;
            while (!stack.empty()) {
                std::vector<uint32_t> chain = std::move(stack.back());
                stack.pop_back();
                for (uint32_t u : preds[chain.back()]) {
                    std::vector<uint32_t> next = chain;
                    next.push_back(u);
                    if (u == s) {
                        std::reverse(next.begin(), next.end());
                        out.put(Tuple{g.indices[s], g.indices[t], DataValue((double)d[t]), path_value(next, g.indices)});
                        if (++emitted > 1000000) throw CozoError("algo::too_many_paths", "keep_ties: more than 1 000 000 shortest paths between one pair");
                    } else {
                        stack.push_back(std::move(next));
                    }
                }
            }
            poison.check();
        }
    }
    for (size_t si = 0; si < ns && !ties; si++) {
        const uint32_t s = starting_nodes[si];
        auto emit = [&](uint32_t t) {
            const float cost = dist[si * gr.n + t];
            DataValue path = std::isfinite(cost) ? path_value(walk_back(parent.data() + si * gr.n, s, t), g.indices)
                                                 : DataValue::list({});  // unreachable: (inf, []) (:319-324)
            out.put(Tuple{g.indices[s], g.indices[t], DataValue((double)cost), std::move(path)});
        };
        if (termination)
            for (uint32_t t : termination_nodes) emit(t);
        else
            for (uint32_t t = 0; t < gr.n; t++) emit(t);
        poison.check();
    }
}

// ---- ClusteringCoefficients -----------------------------------------------------------------------------------
void ClusteringCoefficients::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    GraphWithIndices g = edges.as_directed_graph(/*undirected=*/true);  // triangles.rs:36
    if (g.indices.empty()) return;
    const DirectedCsrGraph &gr = g.graph;
    std::vector<uint64_t> tri(gr.n);
    std::vector<uint32_t> deg(gr.n);
    check_gpu(cz_clustering_coefficients(gr.out_offsets.data(), gr.out_targets.data(), gr.n, gr.edge_count(), tri.data(),
                                         deg.data(), poison.flag_ptr(), CZ_ADJ_SYMMETRIC));  // built two lines up with undirected = true
    for (uint32_t i = 0; i < gr.n; i++) {
        const double d = (double)deg[i];
        const double cc = deg[i] < 2 ? 0.0 : 2.0 * (double)tri[i] / (d * (d - 1.0));  // :80-82, :102
        out.put(Tuple{g.indices[i], DataValue(cc), DataValue((int64_t)tri[i]), DataValue((int64_t)deg[i])});
    }
}

// ---- ClosenessCentrality --------------------------------------------------------------------------------------
void ClosenessCentrality::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    const bool undirected = payload.bool_option("undirected", false);
    GraphWithIndices g = edges.as_directed_weighted_graph(undirected, false);
    const DirectedCsrGraph &gr = g.graph;
    const uint32_t n = gr.n;
    if (n == 0) return;
    // all_pairs_shortest_path.rs:113-144: the all-sources SSSP and the per-start f32 sums (:118-122) both run on the device
    std::vector<double> cent(n, 0.0);
    check_gpu(cz_closeness(gr.out_offsets.data(), gr.out_targets.data(), gr.out_weights.data(), n, gr.edge_count(), cent.data(),
                           poison.flag_ptr()));
    poison.check();
    for (uint32_t v = 0; v < n; v++) out.put(Tuple{g.indices[v], DataValue(cent[v])});
}

// ---- BetweennessCentrality (algos/all_pairs_shortest_path.rs:31-95 over dijkstra_keep_ties) -----------------------
// The reference enumerates ALL shortest paths from every start and gives each path's middle nodes 1 / (paths to that target).
// cz_betweenness computes the same sums on the device without enumerating: SSSP from every start in batches (bit-exact f32
// costs); the tight edges (dist[u] + w == dist[v] in f32, one per edge occurrence = the reference's back_pointers) form a DAG;
// sigma = path counts along it; Brandes' dependency delta(v) = sum over tight (v, x) of sigma(v) / sigma(x) * (1 + delta(x))
// is sum over targets of (paths through v) / l.  f64 sums (the reference adds 1/l path by path in f32): equal within 1e-5.
void BetweennessCentrality::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    const bool undirected = payload.bool_option("undirected", false);
    GraphWithIndices g = edges.as_directed_weighted_graph(undirected, false);
    const DirectedCsrGraph &gr = g.graph;
    const uint32_t n = gr.n;
    if (n == 0) return;
    for (float w : gr.out_weights)
        if (!(w > 0.0f)) throw CozoError("algo::betweenness_needs_positive_weights", "BetweennessCentrality on the GPU path needs positive edge weights");
    std::vector<double> cent(n, 0.0);
    const int rc = cz_betweenness(gr.out_offsets.data(), gr.out_targets.data(), gr.out_weights.data(), n, gr.edge_count(), cent.data(),
                                  poison.flag_ptr());
    if (rc == CZ_E_UNSUPPORTED) throw CozoError("algo::betweenness_absorbed_weight", cz_last_error());
    check_gpu(rc);
    poison.check();
    for (uint32_t v = 0; v < n; v++) out.put(Tuple{g.indices[v], DataValue(cent[v])});
}

// ---- LabelPropagation (algos/label_propagation.rs:27-109, one fixed execution: see the header) ------------------------------
void LabelPropagation::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    const FixedRuleInputRelation &edges = payload.get_input(0);
    const bool undirected = payload.bool_option("undirected", false);
    const size_t max_iter = payload.pos_integer_option("max_iter", 10);
    GraphWithIndices g = edges.as_directed_weighted_graph(undirected, true);
    const DirectedCsrGraph &gr = g.graph;
    if (gr.n == 0) return;
    std::vector<uint32_t> labels(gr.n);
    check_gpu(cz_label_propagation(gr.out_offsets.data(), gr.out_targets.data(), gr.out_weights.data(), gr.n, gr.edge_count(),
                                   (uint32_t)std::min<size_t>(max_iter, 0xFFFFFFFFu), labels.data(), nullptr, nullptr, poison.flag_ptr(),
                                   undirected ? CZ_ADJ_SYMMETRIC : 0u));  // mirrored rows: symmetric by construction
    poison.check();
    for (uint32_t v = 0; v < gr.n; v++) out.put(Tuple{DataValue((int64_t)labels[v]), g.indices[v]});
}

// ---- DegreeCentrality (host only: the reference's rule is a scan with three counters per node) -----------------
void DegreeCentrality::run(const FixedRulePayload &payload, RegularTempStore &out, const Poison &poison) const {
    struct Deg {
        int64_t total = 0, out = 0, in = 0;
    };
    std::unordered_map<DataValue, Deg, DataValueHash> counter;
    for (const Tuple &t : payload.get_input(0).ensure_min_len(2).iter()) {
        Deg &f = counter[t[0]];
        f.total++;
        f.out++;
        Deg &d = counter[t[1]];
        d.total++;
        d.in++;
        poison.check();
    }
    if (payload.inputs_count() > 1) {
        const FixedRuleInputRelation *nodes = nullptr;
        try {
            nodes = &payload.get_input(1);
        } catch (const FixedRuleInputNotFoundError &) {
        }
        if (nodes)
            for (const Tuple &t : nodes->iter()) {
                if (!t.empty()) counter.emplace(t[0], Deg{});
                poison.check();
            }
    }
    for (const auto &kv : counter)
        out.put(Tuple{kv.first, DataValue(kv.second.total), DataValue(kv.second.out), DataValue(kv.second.in)});
}

}  // namespace cozo
