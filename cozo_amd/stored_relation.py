"""A stored relation (`*edges[...]`) as a fixed rule's input, read off its stored bytes.

`FixedRuleInputRelation` (cozo_amd/fixed_rule.py) mirrors the reference: every row becomes a tuple of values and every
endpoint goes through a map (fixed_rule/mod.rs:136-328).  For a relation that lives in the store that detour is the
slowest step of a whole-graph rule; `StoredInputRelation` hands the key / value bytes of the scan
to libcozo_ingest instead (include/cozo_ingest.h) and only decodes the N node values the rule emits.  Same interface,
same ids, same CSR -- the rules do not know the difference (tests/test_stored_relation.py runs them both ways)."""
from __future__ import annotations

from typing import Any, Iterable, List, Optional, Sequence

import numpy as np

from . import codec
from .fixed_rule import DirectedCsrGraph, FixedRuleInputRelation, _canon
from .ingest import StoredGraph


def _uncanon(c):
    """inverse of fixed_rule._canon"""
    if isinstance(c, tuple):
        if len(c) == 2 and c[0] == "b" and isinstance(c[1], bool):
            return c[1]
        if len(c) == 2 and c[0] == "f" and isinstance(c[1], float):
            return c[1]
        return [_uncanon(x) for x in c]
    return c


class _InvIndices:
    """`inv_indices` (value -> id) answered by the ingest handle's table; keys are fixed_rule._canon forms"""

    def __init__(self, g: StoredGraph, indices: List[Any]):
        self._g, self._indices = g, indices

    def get(self, c, default=None):
        try:
            i = self._g.get_node_idx(_uncanon(c))
        except (TypeError, OverflowError):
            return default
        return default if i is None else i

    def __getitem__(self, c):
        i = self.get(c)
        if i is None:
            raise KeyError(c)
        return i

    def __contains__(self, c):
        return self.get(c) is not None

    def __len__(self):
        return len(self._indices)

    def keys(self):
        return (_canon(v) for v in self._indices)

    def __iter__(self):
        return self.keys()


def _graph_of(g: StoredGraph) -> DirectedCsrGraph:
    d = DirectedCsrGraph.__new__(DirectedCsrGraph)
    d.n = g.n
    ooff, otgt, ow = g.csr(False)
    ioff, isrc, _ = g.csr(True)
    d.out_offsets, d.out_targets, d.out_weights = ooff.astype(np.uint64), otgt, ow
    d.in_offsets, d.in_sources = ioff.astype(np.uint64), isrc
    return d


class StoredInputRelation(FixedRuleInputRelation):
    def __init__(self, rows: codec.StoredRows, bindings: Optional[Sequence[str]] = None, arity: Optional[int] = None):
        self._stored = rows
        self._decoded: Optional[List[tuple]] = None
        self._bindings = list(bindings) if bindings is not None else None
        if arity is None:
            arity = len(codec.decode_tuple_from_kv(*rows.row(0))) if len(rows) else (len(self._bindings) if self._bindings else 0)
        self._arity = arity
        self._prefix_index = None

    @property
    def _rows(self) -> List[tuple]:  # the scan, decoded on first use (only rules that read rows pay for it)
        if self._decoded is None:
            self._decoded = [tuple(t) for t in self._stored.tuples()]
        return self._decoded

    def _ingest(self, **kw):
        from .fixed_rule import NotAnEdgeError
        from .ingest import CZI_E_NOT_AN_EDGE, CozoIngestError
        try:
            g = StoredGraph(self._stored, **kw)
        except CozoIngestError as e:
            if e.code == CZI_E_NOT_AN_EDGE:
                raise NotAnEdgeError() from e
            raise
        return g, g.indices()

    def as_directed_graph(self, undirected: bool):
        g, indices = self._ingest(undirected=undirected)
        return _graph_of(g), indices, _InvIndices(g, indices)

    def as_directed_weighted_graph(self, undirected: bool, allow_negative_weights: bool):
        from .ingest import CZI_E_BAD_WEIGHT, CozoIngestError
        try:
            g, indices = self._ingest(undirected=undirected, weighted=True, allow_negative_weights=allow_negative_weights)
        except CozoIngestError as e:
            if e.code == CZI_E_BAD_WEIGHT:
                return super().as_directed_weighted_graph(undirected, allow_negative_weights)  # raises with the offending value
            raise
        return _graph_of(g), indices, _InvIndices(g, indices)

    def as_ordered_graph(self, extra_nodes: Iterable = ()):
        extra = list(extra_nodes)
        g, indices = self._ingest(ordered_ids=True)
        inv = _InvIndices(g, indices)
        if any(_canon(v) not in inv for v in extra):  # a start / goal with no edge gets an id of its own: the generic path
            return super().as_ordered_graph(extra)
        return _graph_of(g), indices, inv


class _LazyRows:
    """rows[i] of a stored relation, decoded when asked for (a k-NN query touches k rows per parent tuple, not N)"""

    def __init__(self, rows: codec.StoredRows):
        self._rows = rows

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i: int) -> tuple:
        return tuple(codec.decode_tuple_from_kv(*self._rows.row(int(i))))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def stored_base_relation(rows: codec.StoredRows, keys: Sequence[str], non_keys: Sequence[str]):
    """the base relation of an index for HnswSearchRA (cozo_amd/hnsw.py), rows decoded on demand"""
    from .hnsw import BaseRelation
    return BaseRelation(keys=list(keys), non_keys=list(non_keys), rows=_LazyRows(rows))
