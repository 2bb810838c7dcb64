"""Host-side mirror of cozo-core's fixed-rule plugin surface over libcozo_gpu (include/cozo_gpu.h).

Names, argument meaning and error behaviour follow cozo-core/src/fixed_rule/mod.rs:
`FixedRule` (:538-567), `FixedRulePayload` (:47-51, 331-535), `FixedRuleInputRelation` (:54-328, incl.
`as_directed_graph` :136-200 and `as_directed_weighted_graph` :208-328), `RegularTempStore`
(runtime/temp_store.rs:26-29), `Poison` (runtime/db.rs:1926-1942), the registry rule of
`Db::register_fixed_rule` (runtime/db.rs:760-776).  The rules themselves are the GPU forms of
`PageRank` (algos/pagerank.rs:29-56), `ShortestPathBFS` (algos/shortest_path_bfs.rs:35-113), `Bfs`
(algos/bfs.rs:25-113), `ConnectedComponents` (algos/strongly_connected_components.rs:42-77, strong = false)
and `ShortestPathDijkstra` (algos/shortest_path_dijkstra.rs:33-153): each one reads its options and inputs
exactly like the reference, maps node values to dense ids, hands a CSR to the C ABI (cozo_amd.graph) and
writes the reference's rows to `out`.  There is no CPU fallback: without the device library every `run` fails.

A `DataValue` is any of None / bool / int / float / str / bytes / list / tuple here, ordered like
data/value.rs (Null < Bool < Num < Str < Bytes < List); relations are sets of tuples iterated in key order.
"""
from __future__ import annotations

import math
import struct
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import graph as _graph


# ---- errors (same names / diagnostic codes as the reference) -------------------------------------------------
class FixedRuleError(Exception):
    code = "algo::error"


class NotAnEdgeError(FixedRuleError):  # fixed_rule/mod.rs:846-850
    code = "algo::not_an_edge"

    def __init__(self):
        super().__init__("The relation cannot be interpreted as an edge")


class BadEdgeWeightError(FixedRuleError):  # fixed_rule/mod.rs:852-860
    code = "algo::invalid_edge_weight"

    def __init__(self, val):
        super().__init__(f"The value {val!r} at the third position in the relation cannot be interpreted as edge weights")


class InputRelationArityError(FixedRuleError):  # fixed_rule/mod.rs:68-72
    code = "algo::input_relation_bad_arity"

    def __init__(self, need, got):
        super().__init__(f"Input relation to algorithm has insufficient arity: should be at least {need} but is {got}")


class FixedRuleOptionNotFoundError(FixedRuleError):  # data/program.rs:291-300
    code = "fixed_rule::arg_not_found"

    def __init__(self, name, rule_name):
        super().__init__(f"Cannot find a required named option '{name}' for '{rule_name}'")


class WrongFixedRuleOptionError(FixedRuleError):  # data/program.rs:301-312
    code = "fixed_rule::arg_wrong"

    def __init__(self, name, rule_name, help_):
        super().__init__(f"Wrong value for option '{name}' of '{rule_name}': {help_}")


class FixedRuleInputNotFoundError(FixedRuleError):  # FixedRuleNotEnoughRelationError, data/program.rs:339-348
    code = "fixed_rule::not_enough_args"

    def __init__(self, idx, rule_name):
        super().__init__(f"Cannot find a required positional argument at index {idx} for '{rule_name}'")


class NodeNotFoundError(FixedRuleError):  # fixed_rule/mod.rs:875-885
    code = "algo::node_with_key_not_found"

    def __init__(self, missing):
        super().__init__(f"Required node with key {missing!r} not found")


class FixedRuleNameConflict(FixedRuleError):  # runtime/db.rs:769-774, 780-782 (plain `bail!`s, no code)
    code = None


ProcessKilled = _lib.ProcessKilled  # runtime/db.rs:1932-1940, raised when the C ABI returns CZ_E_CANCELLED


# ---- DataValue ordering (data/value.rs: derive(Ord) over the enum, Num compares numerically) ------------------
def _rank(v) -> int:
    if v is None:
        return 0
    if isinstance(v, (bool, np.bool_)):
        return 1
    if isinstance(v, (int, float, np.integer, np.floating)):
        return 2
    if isinstance(v, str):
        return 3
    if isinstance(v, (bytes, bytearray)):
        return 4
    if isinstance(v, (list, tuple)):
        return 9
    raise TypeError(f"not a DataValue: {type(v).__name__}")


def sort_key(v):
    """total order of DataValue: Null < Bool < Num < Str < Bytes < List; an Int sorts before the equal Float
    (data/value.rs Num::cmp); NaN greatest among numbers."""
    r = _rank(v)
    if r == 2:
        # floats by f64::total_cmp (-0.0 < +0.0, NaN by its sign), an Int through its f64 image and before the Float it
        # equals (data/value.rs:575-598); Ints sharing an image by their exact value
        is_float = isinstance(v, (float, np.floating))
        u = struct.unpack(">Q", struct.pack(">d", float(v)))[0]
        u = (~u & 0xFFFFFFFFFFFFFFFF) if u >> 63 else (u | 0x8000000000000000)
        return (2, u, 1, 0) if is_float else (2, u, 0, int(v))
    if r == 9:
        return (9, tuple(sort_key(x) for x in v))
    if r == 1:
        return (1, bool(v))
    if r == 0:
        return (0,)
    return (r, v)


def _canon(v):
    """hashable canonical form of a value (lists become tuples; Int 1 and Float 1.0 stay distinct like DataValue)."""
    if isinstance(v, (list, tuple)):
        return tuple(_canon(x) for x in v)
    if isinstance(v, (bool, np.bool_)):
        return ("b", bool(v))
    if isinstance(v, (float, np.floating)):
        # identity by bit pattern, like DataValue's Ord (f64::total_cmp, data/value.rs:595): -0.0 and 0.0 are two nodes,
        # NaNs with equal bits are one -- Python's == / hash would merge the former and never match the latter
        return ("f", struct.pack(">d", float(v)))
    if isinstance(v, np.integer):
        return int(v)
    return v


def _tuple_key(t):
    return tuple(sort_key(x) for x in t)


def _get_float(v) -> Optional[float]:  # DataValue::get_float: Num only (data/value.rs)
    if isinstance(v, (bool, np.bool_)) or v is None:
        return None
    if isinstance(v, (int, float, np.integer, np.floating)):
        return float(v)
    return None


# ---- Poison / temp store --------------------------------------------------------------------------------------
class Poison:
    """runtime/db.rs:1926-1942: an atomic flag the running rule polls; the byte is what `poison` of the C ABI reads."""

    def __init__(self):
        self.flag = np.zeros(1, dtype=np.uint8)

    def kill(self):
        self.flag[0] = 1

    def check(self):
        if self.flag[0]:
            raise ProcessKilled(_lib.CZ_E_CANCELLED, "Running query is killed before completion")


class RegularTempStore:
    """runtime/temp_store.rs:26-29: an ordered set of tuples."""

    def __init__(self):
        self._rows: Dict[Any, tuple] = {}

    def put(self, tuple_: Sequence):
        t = tuple(tuple_)
        self._rows[_canon(t)] = t

    def __len__(self):
        return len(self._rows)

    def __iter__(self):
        return iter(sorted(self._rows.values(), key=_tuple_key))

    def rows(self) -> List[tuple]:
        return list(iter(self))


# ---- input relations -----------------------------------------------------------------------------------------
class DirectedCsrGraph:
    """What GraphBuilder::csr_layout(CsrLayout::Sorted).edges(..).build() yields (graph_builder 0.4.0): both
    adjacency directions, neighbour lists ascending by dense id, parallel edges kept."""

    def __init__(self, n: int, frm: np.ndarray, to: np.ndarray, weights: Optional[np.ndarray] = None):
        self.n = int(n)
        frm = np.asarray(frm, dtype=np.uint32)
        to = np.asarray(to, dtype=np.uint32)
        self.out_offsets, self.out_targets, self.out_weights = self._csr(self.n, frm, to, weights)
        self.in_offsets, self.in_sources, _ = self._csr(self.n, to, frm, None)

    @staticmethod
    def _csr(n, a, b, w):
        # sort by (a, b); CsrLayout::Sorted sorts each adjacency list by target (for weighted targets: by target
        # id, ties keep input order -- a stable sort)
        order = np.lexsort((b, a)) if a.size else np.zeros(0, dtype=np.int64)
        off = np.zeros(n + 1, dtype=np.uint64)
        if a.size:
            off[1:] = np.cumsum(np.bincount(a, minlength=n))
        return off, b[order].astype(np.uint32), (None if w is None else np.asarray(w, dtype=np.float32)[order])

    def node_count(self) -> int:
        return self.n

    def edge_count(self) -> int:
        return int(self.out_targets.size)

    def out_degrees(self) -> np.ndarray:
        return np.diff(self.out_offsets).astype(np.uint32)


class FixedRuleInputRelation:
    """fixed_rule/mod.rs:54-328.  `rows`: an iterable of tuples (a stored or in-memory relation: a set, scanned in
    key order); `bindings`: the symbols the rule application binds the columns to."""

    def __init__(self, rows: Iterable[Sequence], bindings: Optional[Sequence[str]] = None, arity: Optional[int] = None):
        uniq: Dict[Any, tuple] = {}
        for r in rows:
            t = tuple(r)
            uniq[_canon(t)] = t
        self._rows = sorted(uniq.values(), key=_tuple_key)
        self._bindings = list(bindings) if bindings is not None else None
        self._arity = arity if arity is not None else (len(self._rows[0]) if self._rows else
                                                       (len(self._bindings) if self._bindings else 0))
        self._prefix_index: Optional[Dict[Any, List[tuple]]] = None

    def arity(self) -> int:
        return self._arity

    def ensure_min_len(self, n: int) -> "FixedRuleInputRelation":
        if self._arity < n:
            raise InputRelationArityError(n, self._arity)
        return self

    def get_binding_map(self, offset: int = 0) -> Dict[str, int]:
        return {s: i + offset for i, s in enumerate(self._bindings or [])}

    def iter(self):
        return iter(self._rows)

    def prefix_iter(self, prefix):
        if self._prefix_index is None:
            ix: Dict[Any, List[tuple]] = {}
            for t in self._rows:
                if t:
                    ix.setdefault(_canon(t[0]), []).append(t)
            self._prefix_index = ix
        return iter(self._prefix_index.get(_canon(prefix), ()))

    # -- node-value -> dense id mapping -------------------------------------------------------------------
    def _edge_columns(self):
        frm, to = [], []
        for t in self._rows:
            if len(t) < 2:
                raise NotAnEdgeError()
            frm.append(t[0])
            to.append(t[1])
        return frm, to

    @staticmethod
    def _assign_first_appearance(frm: list, to: list):
        """ids in first-appearance order, source before destination, row by row (fixed_rule/mod.rs:163-180)."""
        m = len(frm)
        if m and all(type(x) is int for x in frm) and all(type(x) is int for x in to):
            inter = np.empty(2 * m, dtype=np.int64)
            inter[0::2] = frm
            inter[1::2] = to
            vals, first, inv = np.unique(inter, return_index=True, return_inverse=True)
            order = np.argsort(first, kind="stable")  # unique value j gets id rank_of_first_appearance
            idmap = np.empty(order.size, dtype=np.uint32)
            idmap[order] = np.arange(order.size, dtype=np.uint32)
            ids = idmap[inv]
            indices = [int(v) for v in vals[order]]
            inv_indices = {v: i for i, v in enumerate(indices)}
            return ids[0::2].copy(), ids[1::2].copy(), indices, inv_indices
        indices: list = []
        inv_indices: Dict[Any, int] = {}
        fi = np.empty(m, dtype=np.uint32)
        ti = np.empty(m, dtype=np.uint32)
        for i in range(m):
            for val, arr in ((frm[i], fi), (to[i], ti)):
                c = _canon(val)
                idx = inv_indices.get(c)
                if idx is None:
                    idx = len(indices)
                    inv_indices[c] = idx
                    indices.append(val)
                arr[i] = idx
        return fi, ti, indices, inv_indices

    def as_directed_graph(self, undirected: bool):
        """-> (DirectedCsrGraph, indices: id -> value, inv_indices: canonical value -> id)   (mod.rs:136-200)"""
        frm, to = self._edge_columns()
        fi, ti, indices, inv = self._assign_first_appearance(frm, to)
        if undirected:  # each row yields (f, t) then (t, f): mod.rs:187-191
            f2 = np.empty(2 * fi.size, dtype=np.uint32)
            t2 = np.empty(2 * fi.size, dtype=np.uint32)
            f2[0::2], f2[1::2] = fi, ti
            t2[0::2], t2[1::2] = ti, fi
            fi, ti = f2, t2
        return DirectedCsrGraph(len(indices), fi, ti), indices, inv

    def as_directed_weighted_graph(self, undirected: bool, allow_negative_weights: bool):
        """third column -> f32 weight, default 1.0; non-numeric / non-finite / negative rejected (mod.rs:208-328)"""
        frm, to = self._edge_columns()
        w = np.empty(len(frm), dtype=np.float32)
        for i, t in enumerate(self._rows):
            if len(t) < 3:
                w[i] = 1.0
                continue
            f = _get_float(t[2])
            if f is None or not math.isfinite(f) or (f < 0.0 and not allow_negative_weights):
                raise BadEdgeWeightError(t[2])
            w[i] = np.float32(f)
        fi, ti, indices, inv = self._assign_first_appearance(frm, to)
        if undirected:
            f2 = np.empty(2 * fi.size, dtype=np.uint32)
            t2 = np.empty(2 * fi.size, dtype=np.uint32)
            w2 = np.empty(2 * fi.size, dtype=np.float32)
            f2[0::2], f2[1::2] = fi, ti
            t2[0::2], t2[1::2] = ti, fi
            w2[0::2], w2[1::2] = w, w
            fi, ti, w = f2, t2, w2
        return DirectedCsrGraph(len(indices), fi, ti, w), indices, inv

    def as_ordered_graph(self, extra_nodes: Iterable = ()):
        """For the rules that walk `prefix_iter` (ShortestPathBFS, Bfs): neighbours must come in KEY order of the
        `to` value, so ids are the rank of the value in DataValue order (plus `extra_nodes`: starts / goals that
        have no edge).  -> (DirectedCsrGraph, indices, inv_indices)"""
        frm, to = self._edge_columns()
        vals: Dict[Any, Any] = {}
        for v in frm:
            vals.setdefault(_canon(v), v)
        for v in to:
            vals.setdefault(_canon(v), v)
        for v in extra_nodes:
            vals.setdefault(_canon(v), v)
        indices = sorted(vals.values(), key=sort_key)
        inv = {_canon(v): i for i, v in enumerate(indices)}
        fi = np.fromiter((inv[_canon(v)] for v in frm), dtype=np.uint32, count=len(frm))
        ti = np.fromiter((inv[_canon(v)] for v in to), dtype=np.uint32, count=len(to))
        return DirectedCsrGraph(len(indices), fi, ti), indices, inv


class FixedRulePayload:
    """fixed_rule/mod.rs:47-51, 331-535.  `options`: already-evaluated constants (the reference holds `Expr`s and
    calls eval_to_const); an `expr_option` is a Python callable over the bound tuple."""

    def __init__(self, name: str, inputs: Sequence[FixedRuleInputRelation], options: Optional[Dict[str, Any]] = None):
        self._name = name
        self._inputs = list(inputs)
        self.options = dict(options or {})

    def inputs_count(self) -> int:
        return len(self._inputs)

    def get_input(self, idx: int) -> FixedRuleInputRelation:
        if idx >= len(self._inputs) or self._inputs[idx] is None:
            raise FixedRuleInputNotFoundError(idx, self._name)
        return self._inputs[idx]

    def name(self) -> str:
        return self._name

    def _missing(self, name):
        return FixedRuleOptionNotFoundError(name, self._name)

    def expr_option(self, name: str, default=None):
        if name in self.options:
            return self.options[name]
        if default is None:
            raise self._missing(name)
        return default

    def string_option(self, name: str, default: Optional[str] = None) -> str:
        if name in self.options:
            v = self.options[name]
            if not isinstance(v, str):
                raise WrongFixedRuleOptionError(name, self._name, "a string is required")
            return v
        if default is None:
            raise self._missing(name)
        return default

    def integer_option(self, name: str, default: Optional[int] = None) -> int:
        if name in self.options:
            v = self.options[name]
            if isinstance(v, (bool, np.bool_)) or not isinstance(v, (int, float, np.integer, np.floating)):
                raise WrongFixedRuleOptionError(name, self._name, "an integer is required")
            if isinstance(v, (float, np.floating)):  # Num::get_int: a float with an integral value counts
                if float(v) != math.floor(float(v)) or not math.isfinite(float(v)):
                    raise self._missing(name)  # sic: the reference reports "not found" here (mod.rs:421-427)
                return int(v)
            return int(v)
        if default is None:
            raise self._missing(name)
        return default

    def pos_integer_option(self, name: str, default: Optional[int] = None) -> int:
        i = self.integer_option(name, default)
        if i <= 0:
            raise WrongFixedRuleOptionError(name, self._name, "a positive integer is required")
        return i

    def non_neg_integer_option(self, name: str, default: Optional[int] = None) -> int:
        i = self.integer_option(name, default)
        if i < 0:
            raise WrongFixedRuleOptionError(name, self._name, "a non-negative integer is required")
        return i

    def float_option(self, name: str, default: Optional[float] = None) -> float:
        if name in self.options:
            v = self.options[name]
            if isinstance(v, (bool, np.bool_)) or not isinstance(v, (int, float, np.integer, np.floating)):
                raise WrongFixedRuleOptionError(name, self._name, "a floating number is required")
            return float(v)
        if default is None:
            raise self._missing(name)
        return default

    def unit_interval_option(self, name: str, default: Optional[float] = None) -> float:
        f = self.float_option(name, default)
        if not (0.0 <= f <= 1.0):
            raise WrongFixedRuleOptionError(name, self._name, "a number between 0. and 1. is required")
        return f

    def bool_option(self, name: str, default: Optional[bool] = None) -> bool:
        if name in self.options:
            v = self.options[name]
            if not isinstance(v, (bool, np.bool_)):
                raise WrongFixedRuleOptionError(name, self._name, "a boolean value is required")
            return bool(v)
        if default is None:
            raise self._missing(name)
        return default


class FixedRule:
    """`pub trait FixedRule: Send + Sync` (fixed_rule/mod.rs:538-567)."""

    def init_options(self, options: Dict[str, Any]) -> None:
        return None

    def arity(self, options: Dict[str, Any], rule_head: Sequence[str]) -> int:
        raise NotImplementedError

    def run(self, payload: FixedRulePayload, out: RegularTempStore, poison: Poison) -> None:
        raise NotImplementedError


def _path(parent: np.ndarray, start: int, goal: int) -> List[int]:
    """walk the backtrace from `goal` to `start` (shortest_path_bfs.rs:85-92)"""
    route = []
    cur = goal
    while cur != start:
        route.append(cur)
        cur = int(parent[cur])
        if cur == _lib.CZ_NONE:
            raise AssertionError("broken backtrace")
    route.append(start)
    route.reverse()
    return route


# ---- the rules ------------------------------------------------------------------------------------------------
class PageRank(FixedRule):
    """algos/pagerank.rs:29-56 -> cz_pagerank.  Rows: (node, score as f64)."""

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        undirected = payload.bool_option("undirected", False)
        theta = np.float32(payload.unit_interval_option("theta", 0.85))
        epsilon = np.float32(payload.unit_interval_option("epsilon", 0.0001))
        iterations = payload.pos_integer_option("iterations", 10)
        # an option of the GPU rule alone (absent = false): `in_place: true` runs graph::page_rank under the reading that refreshes a
        # node's contribution inside the sweep -- the reference's one-thread execution if the crate does that (cz_pagerank_inplace;
        # DESIGN section 3).  Which reading is the crate's is settled by oracle/ref_fixtures on a box with cargo.
        in_place = payload.bool_option("in_place", False)
        graph, indices, _ = edges.as_directed_graph(undirected)
        if not indices:
            return
        if in_place:
            scores = _graph.pagerank_inplace(graph.in_offsets, graph.in_sources, graph.out_degrees(), damping=theta,
                                             tolerance=float(epsilon), max_iter=iterations, poison=poison.flag)[0]
        else:
            scores, _n_run, _err = _graph.pagerank(graph.in_offsets, graph.in_sources, graph.out_degrees(), damping=theta,
                                                   tolerance=float(epsilon), max_iter=iterations, poison=poison.flag)
        for idx, score in enumerate(scores):
            out.put((indices[idx], float(score)))


class ShortestPathBFS(FixedRule):
    """algos/shortest_path_bfs.rs:35-113 -> cz_bfs.  Rows: (start, end, List(path) | Null)."""

    def arity(self, options, rule_head) -> int:
        return 3

    def run(self, payload, out, poison):
        edges = payload.get_input(0).ensure_min_len(2)
        starting_nodes = [t[0] for t in payload.get_input(1).ensure_min_len(1).iter()]
        ending = {}
        for t in payload.get_input(2).ensure_min_len(1).iter():
            ending.setdefault(_canon(t[0]), t[0])
        ending_nodes = sorted(ending.values(), key=sort_key)  # BTreeSet iteration order
        if not starting_nodes:
            return
        graph, indices, inv = edges.as_ordered_graph(list(starting_nodes) + ending_nodes)
        starts = np.array([inv[_canon(s)] for s in starting_nodes], dtype=np.uint32)
        goals = np.array([inv[_canon(e)] for e in ending_nodes], dtype=np.uint32)
        if goals.size == 0:
            return
        parent, _, _, _ = _graph.bfs(graph.out_offsets, graph.out_targets, starts, goals=goals, poison=poison.flag)
        for si, s in enumerate(starting_nodes):
            for gi, e in enumerate(ending_nodes):
                g = int(goals[gi])
                if parent[si, g] != _lib.CZ_NONE:
                    out.put((s, e, [indices[i] for i in _path(parent[si], int(starts[si]), g)]))
                else:
                    out.put((s, e, None))
            poison.check()


class Bfs(FixedRule):
    """algos/bfs.rs:25-113 -> cz_bfs_shared.  `condition` is a predicate over the node tuple (the
    reference compiles an Expr to bytecode and evaluates it on the tuple bound from `nodes`)."""

    def arity(self, options, rule_head) -> int:
        return 3

    def run(self, payload, out, poison):
        edges = payload.get_input(0).ensure_min_len(2)
        nodes = payload.get_input(1)
        try:
            starting_rel = payload.get_input(2)
        except FixedRuleInputNotFoundError:
            starting_rel = nodes
        limit = payload.pos_integer_option("limit", 1)
        condition: Callable = payload.expr_option("condition", None)
        skip_query_nodes = bool(getattr(condition, "only_node_id", False))  # binding_indices subset of {0}
        start_vals = [t[0] for t in starting_rel.iter()]
        if not start_vals:
            return
        graph, indices, inv = edges.as_ordered_graph(start_vals)
        starts = np.array([inv[_canon(s)] for s in start_vals], dtype=np.uint32)
        # one backtrace and one discovery sequence for all starts: the default is EVERY node as a start (bfs.rs:33).  The condition
        # is evaluated level by level as the device discovers the nodes, and the traversal ends with the level in which the
        # `limit`-th node passed (bfs.rs:88-91: `break 'outer`) -- with the default limit of 1 that is usually a few levels in.
        found: List[Tuple[int, int, int]] = []
        start_pos = {}
        for si, s in enumerate(starts):
            start_pos.setdefault(int(s), si)  # (a repeated start is skipped as already visited: the first one discovered)
        missing: List[int] = []

        def on_level(start: int, level_nodes) -> bool:
            si = start_pos[start]
            for to in level_nodes.tolist():
                to_val = indices[to]
                if skip_query_nodes:
                    cand_tuple = (to_val,)
                else:
                    cand_tuple = next(nodes.prefix_iter(to_val), None)
                    if cand_tuple is None:
                        missing.append(to)
                        return True
                if condition(cand_tuple):
                    found.append((si, start, to))
                    if len(found) >= limit:
                        return True
                poison.check()
            return False

        parent, order, first = _graph.bfs_shared(graph.out_offsets, graph.out_targets, starts, poison=poison.flag, on_level=on_level)
        if missing:
            # sic: the reference reports the *candidate* (the discoverer) as missing (bfs.rs:74-77)
            raise NodeNotFoundError(indices[int(parent[missing[0]])])
        # the backtrace is shared across starts (bfs.rs:44); every node has exactly one discoverer
        for si, s, e in found:
            out.put((indices[s], indices[e], [indices[i] for i in _path(parent, s, e)]))


class ConnectedComponents(FixedRule):
    """StronglyConnectedComponent::new(false) (algos/strongly_connected_components.rs:42-77) ->
    cz_connected_components.  Rows: (node, group id i64); nodes that appear only in input 1 get fresh ids."""

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        graph, indices, inv = edges.as_directed_graph(True)
        n_groups = 0
        if indices:
            grp, n_groups = _graph.connected_components(graph.out_offsets, graph.out_targets, poison=poison.flag)
            for idx, g in enumerate(grp):
                out.put((indices[idx], int(g)))
        counter = n_groups
        try:
            nodes = payload.get_input(1)
        except FixedRuleInputNotFoundError:
            nodes = None
        if nodes is not None:
            seen = set(inv.keys())
            for t in nodes.iter():
                c = _canon(t[0])
                if c not in seen:
                    seen.add(c)
                    out.put((t[0], counter))
                    counter += 1


class StronglyConnectedComponent(FixedRule):
    """`strong = true` is Tarjan's DFS numbering -- sequential by nature and its group ids depend on the DFS
    order; it is not on the GPU path (out of scope).  Refuses loudly instead of falling back."""

    def __init__(self, strong: bool):
        self.strong = strong
        self._cc = ConnectedComponents()

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        if self.strong:
            raise _lib.CozoGpuError(_lib.CZ_E_UNSUPPORTED, "StronglyConnectedComponents is not available on the GPU path")
        self._cc.run(payload, out, poison)


class ShortestPathDijkstra(FixedRule):
    """algos/shortest_path_dijkstra.rs:33-153 -> cz_sssp.  Rows: (start, target, cost f64, List(path)); an
    unreachable target has cost inf and an empty path (:319-321)."""

    def arity(self, options, rule_head) -> int:
        return 4

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        starting = payload.get_input(1)
        try:
            termination = payload.get_input(2)
        except FixedRuleInputNotFoundError:
            termination = None
        undirected = payload.bool_option("undirected", False)
        keep_ties = payload.bool_option("keep_ties", False)
        graph, indices, inv = edges.as_directed_weighted_graph(undirected, False)
        starting_nodes = sorted({inv[_canon(t[0])] for t in starting.iter() if _canon(t[0]) in inv})
        termination_nodes = None
        if termination is not None:
            termination_nodes = sorted({inv[_canon(t[0])] for t in termination.iter() if _canon(t[0]) in inv})
        if not starting_nodes:
            return
        targets = range(graph.n) if termination_nodes is None else termination_nodes
        if termination_nodes is not None and not termination_nodes:
            return
        # `keep_ties` only takes effect when a termination relation is given (:73-86: without one the reference calls the plain
        # `dijkstra` whatever the option says)
        keep_ties = keep_ties and termination_nodes is not None
        if keep_ties and graph.out_weights.size and not (graph.out_weights > 0).all():
            raise FixedRuleError("keep_ties on the GPU path needs positive edge weights")
        starts = np.array(starting_nodes, dtype=np.uint32)
        # with a termination relation the search stops once every target is settled, like dijkstra()'s goal set (:300-306)
        goals = None if termination_nodes is None else np.array(termination_nodes, dtype=np.uint32)
        dist, parent = _graph.sssp(graph.out_offsets, graph.out_targets, graph.out_weights, starts, poison=poison.flag, goals=goals)
        if keep_ties:
            off = np.asarray(graph.out_offsets, dtype=np.int64)
            src_of = np.repeat(np.arange(graph.n, dtype=np.int64), np.diff(off))
        for si, s in enumerate(starting_nodes):
            if not keep_ties:
                for t in targets:
                    cost = float(dist[si, t])
                    path = [] if not math.isfinite(cost) else [indices[i] for i in _path(parent[si], s, t)]
                    out.put((indices[s], indices[t], cost, path))
                continue
            # dijkstra_keep_ties (:341-450): back_pointers[v] = every edge (u, v) with dist[u] + w == dist[v] in f32 -- read off
            # the device's bit-exact distances -- and EVERY path through them is a row.  The start as its own target has no
            # back pointer and therefore no row (:397-430 collects nothing for it).
            d = dist[si]
            with np.errstate(invalid="ignore"):
                tight = np.isfinite(d[src_of]) & ((d[src_of] + graph.out_weights).astype(np.float32) == d[graph.out_targets])
            preds: Dict[int, List[int]] = {}
            for u, v in zip(src_of[tight].tolist(), graph.out_targets[tight].tolist()):
                preds.setdefault(v, []).append(u)
            for t in targets:
                cost = float(d[t])
                if not math.isfinite(cost):
                    out.put((indices[s], indices[t], cost, []))
                    continue
                stack = [[t]]
                emitted = 0
                while stack:
                    chain = stack.pop()
                    for u in preds.get(chain[-1], ()):
                        if u == s:
                            out.put((indices[s], indices[t], cost, [indices[i] for i in reversed(chain + [u])]))
                            emitted += 1
                            if emitted > 1_000_000:
                                raise FixedRuleError("keep_ties: more than 1 000 000 shortest paths between one pair")
                        else:
                            stack.append(chain + [u])
                poison.check()


class ClusteringCoefficients(FixedRule):
    """algos/triangles.rs:25-110 -> cz_clustering_coefficients.  Rows: (node, coefficient f64, triangles, degree)."""

    def arity(self, options, rule_head) -> int:
        return 4

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        graph, indices, _ = edges.as_directed_graph(True)
        if not indices:
            return
        tri, deg = _graph.clustering_coefficients(graph.out_offsets, graph.out_targets, poison=poison.flag, symmetric=True)  # as_directed_graph(True)
        for idx in range(graph.n):
            d, t = int(deg[idx]), int(tri[idx])
            cc = 0.0 if d < 2 else 2.0 * float(t) / (float(d) * (float(d) - 1.0))  # :80-82, :102
            out.put((indices[idx], cc, t, d))


class ClosenessCentrality(FixedRule):
    """algos/all_pairs_shortest_path.rs:97-144: one cost-only Dijkstra per node (`dijkstra_cost_only`, :146-176, the same
    strict-`<` f32 relaxation as `dijkstra`) and, per start, the reference's f32 arithmetic in the reference's order:
    total = sequential sum of the finite distances in node order, nc = their count, centrality = nc * nc / total / (n - 1)
    -> cz_closeness (the all-sources SSSP in batches and the per-start sums, both on the device).
    Rows: (node, centrality as f64)."""

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        undirected = payload.bool_option("undirected", False)
        graph, indices, _ = edges.as_directed_weighted_graph(undirected, False)
        n = graph.n
        if n == 0:
            return
        cent = _graph.closeness(graph.out_offsets, graph.out_targets, graph.out_weights, poison=poison.flag)
        poison.check()
        for i in range(n):
            out.put((indices[i], float(cent[i])))


class BetweennessCentrality(FixedRule):
    """algos/all_pairs_shortest_path.rs:31-95 over `dijkstra_keep_ties` (shortest_path_dijkstra.rs:341-450): from every start,
    ALL shortest paths to every target; each path of >= 3 nodes gives 1 / (number of paths to that target) to each of its
    middle nodes.  The reference enumerates the paths (exponential in ties); cz_betweenness computes the same sums on the
    device without enumerating: SSSP from every node in batches (bit-exact f32 costs), the tight edges
    (dist[u] + w == dist[v] in f32, one per edge occurrence, exactly the reference's back_pointers) form a DAG,
    sigma = path counts along it, and Brandes' dependency
    delta(v) = sum over tight (v, x) of sigma(v) / sigma(x) * (1 + delta(x)) IS sum over targets of (paths through v) / l.
    Accumulated in f64 (the reference adds 1/l path by path in f32): equal within 1e-5 relative, not bit for bit.
    Weights must be positive (a zero-weight cycle sends the reference's path recursion into the ground), and none may be
    absorbed by an f32 path cost (dist[u] + w == dist[u]): such tight edges join nodes of equal distance and can close
    cycles, through which the reference's enumeration would never end.
    Rows: (node, centrality as f64)."""

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        undirected = payload.bool_option("undirected", False)
        graph, indices, _ = edges.as_directed_weighted_graph(undirected, False)
        n = graph.n
        if n == 0:
            return
        w = graph.out_weights
        if w.size and not (w > 0).all():
            raise FixedRuleError("BetweennessCentrality on the GPU path needs positive edge weights")
        try:
            cent = _graph.betweenness(graph.out_offsets, graph.out_targets, w, poison=poison.flag)
        except _lib.CozoGpuError as e:
            if e.code == _lib.CZ_E_UNSUPPORTED:
                raise FixedRuleError(str(e)) from e
            raise
        poison.check()
        for i in range(n):
            out.put((indices[i], float(cent[i])))


class DegreeCentrality(FixedRule):
    """algos/degree_centrality.rs:24-76: a scan with three counters per node -- no graph, nothing for the GPU to do;
    mirrored on the host so that the rule family is complete.  Rows: (node, total, out, in)."""

    def arity(self, options, rule_head) -> int:
        return 4

    def run(self, payload, out, poison):
        counter: Dict[Any, list] = {}
        vals: Dict[Any, Any] = {}
        for t in payload.get_input(0).ensure_min_len(2).iter():
            for pos, col in ((1, t[0]), (2, t[1])):
                c = _canon(col)
                vals.setdefault(c, col)
                ent = counter.setdefault(c, [0, 0, 0])
                ent[0] += 1
                ent[pos] += 1
            poison.check()
        try:
            nodes = payload.get_input(1)
        except FixedRuleInputNotFoundError:
            nodes = None
        if nodes is not None:
            for t in nodes.iter():
                c = _canon(t[0])
                vals.setdefault(c, t[0])
                counter.setdefault(c, [0, 0, 0])
                poison.check()
        for c, (tot, o, i) in counter.items():
            out.put((vals[c], tot, o, i))


class LabelPropagation(FixedRule):
    """algos/label_propagation.rs:27-109 -> cz_label_propagation.  The reference shuffles the node order in every iteration and
    picks a random label among the best-scored ones (`thread_rng`, :63-66 and :85): two of ITS runs do not agree, so the GPU
    rule fixes both choices -- colour classes of a deterministic colouring in ascending order (nodes of one class share no
    edge: a class is updated at once, which is the sequential loop over its nodes) and the smallest label on ties -- and
    returns the result of that one execution, which the reference could produce itself (include/cozo_gpu.h).
    Options as the reference: `undirected` (false), `max_iter` (10).  Rows: (label as i64, node)."""

    def arity(self, options, rule_head) -> int:
        return 2

    def run(self, payload, out, poison):
        edges = payload.get_input(0)
        undirected = payload.bool_option("undirected", False)
        max_iter = payload.pos_integer_option("max_iter", 10)
        graph, indices, _ = edges.as_directed_weighted_graph(undirected, True)
        if graph.n == 0:
            return
        labels, _, _ = _graph.label_propagation(graph.out_offsets, graph.out_targets, graph.out_weights, max_iter, poison=poison.flag,
                                                symmetric=bool(undirected))  # mirrored rows: symmetric by construction
        poison.check()
        for i in range(graph.n):
            out.put((int(labels[i]), indices[i]))


# ---- registry (Db::register_fixed_rule, runtime/db.rs:760-784) --------------------------------------------------
class FixedRuleRegistry:
    """The GPU rules are registered under NEW names next to the built-ins (built-ins cannot be replaced or
    unregistered, runtime/db.rs:779-784); a patched build swaps them into DEFAULT_FIXED_RULES instead
    (fixed_rule/mod.rs:799-802) -- see INTEGRATION.md."""

    BUILTIN = ("PageRank", "ShortestPathBFS", "BFS", "BreadthFirstSearch", "ConnectedComponents",
               "StronglyConnectedComponents", "SCC", "ShortestPathDijkstra", "ClusteringCoefficients", "DegreeCentrality", "ClosenessCentrality",
               "BetweennessCentrality", "LabelPropagation")

    def __init__(self):
        self._rules: Dict[str, FixedRule] = {}
        for name, impl in (("PageRankGpu", PageRank()), ("ShortestPathBFSGpu", ShortestPathBFS()), ("BFSGpu", Bfs()),
                           ("ConnectedComponentsGpu", ConnectedComponents()),
                           ("ShortestPathDijkstraGpu", ShortestPathDijkstra()),
                           ("ClusteringCoefficientsGpu", ClusteringCoefficients()),
                           ("DegreeCentralityGpu", DegreeCentrality()),
                           ("ClosenessCentralityGpu", ClosenessCentrality()),
                           ("BetweennessCentralityGpu", BetweennessCentrality()),
                           ("LabelPropagationGpu", LabelPropagation())):
            self._rules[name] = impl

    def register_fixed_rule(self, name: str, impl: FixedRule) -> None:
        if name in self._rules or name in self.BUILTIN:
            raise FixedRuleNameConflict(f"A fixed rule with the name {name} is already registered")
        self._rules[name] = impl

    def unregister_fixed_rule(self, name: str) -> bool:
        if name in self.BUILTIN:
            raise FixedRuleNameConflict(f"Cannot unregister builtin fixed rule {name}")
        return self._rules.pop(name, None) is not None

    def get(self, name: str) -> FixedRule:
        return self._rules[name]

    def run(self, name: str, inputs: Sequence[FixedRuleInputRelation], options: Optional[Dict[str, Any]] = None,
            poison: Optional[Poison] = None) -> List[tuple]:
        """`?[..] <~ Name(inputs.., options..)`: init_options, arity check, run; returns the sorted rows."""
        impl = self._rules[name]
        opts = dict(options or {})
        impl.init_options(opts)
        arity = impl.arity(opts, ())
        out = RegularTempStore()
        impl.run(FixedRulePayload(name, inputs, opts), out, poison or Poison())
        rows = out.rows()
        assert all(len(r) == arity for r in rows)
        return rows
