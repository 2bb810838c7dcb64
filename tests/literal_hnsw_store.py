"""A literal row store under hnsw_put_vector -- test infrastructure for the oracle's index construction.

The C restatement (oracle/cozo_oracle.c) keeps adjacency lists and a degree field.  This file plays the same insertions on
what the reference actually has: ONE ordered map of `tbl:idx` rows, key (layer, from, to) -> [f64, hash | None, ignore_link],
where (layer, x, x) is the self row of x (its f64 is the degree) and every step is the store_tx.get / put / del of
cozo-core/src/runtime/hnsw.rs.  It exists for the places where the row model and the adjacency model could part ways --
above all extend_candidates, where hnsw_shrink_neighbour selects the target ITSELF and writes a link row onto the target's
self row (:413-433) that hnsw_put_vector then overwrites again (:352-357).  Pure Python loops: small cases only.

`priority_queue::PriorityQueue` is restated as a dict: push on a held key replaces its priority; pop takes the extreme
priority, ties by node id (the crate leaves ties open; the oracle and the kernels use (distance, id) too); iteration is
insertion order (the reference iterates `neighbours` in the crate's internal order, which is just as unspecified)."""
import hashlib

import numpy as np


class MinQueue(dict):  # PriorityQueue<_, Reverse<OrderedFloat>>
    def push(self, key, pri):
        self[key] = pri

    def pop(self):
        key = min(self, key=lambda k: (self[k], k))
        return key, dict.pop(self, key)


class MaxQueue(dict):  # PriorityQueue<_, OrderedFloat>
    def push(self, key, pri):
        self[key] = pri

    def peek(self):
        key = max(self, key=lambda k: (self[k], k))
        return key, self[key]

    def pop(self):
        key, pri = self.peek()
        del self[key]
        return key, pri


class ReferencePanic(Exception):
    """a step at which the reference itself panics (an `unwrap()` on a row that is not there)"""


class LiteralStore:
    def __init__(self, dist, m, ef_construction, extend_candidates=False, keep_pruned_connections=False, row_of=None):
        """dist(a, b) -> f64 on two f32 vectors (the oracle's orc_distance, so that both models see the same bits)"""
        self.dist = dist
        self.m_max, self.m_max0 = m, 2 * m  # runtime/relation.rs:1136-1151
        self.ef_c = ef_construction
        self.extend = extend_candidates
        self.keep_pruned = keep_pruned_connections
        self.row_of = row_of  # base row of every node's vector (None: one vector per row)
        self.rows = {}  # (layer <= 0, from, to) -> [f64, hash | None, bool]
        self.vec = []
        self.self_row_overwrites = 0

    # ---- hnsw_get_neighbours, hnsw.rs:588-629: `key_tup == cand_key.0` skips the self row and every link to another vector
    # of the same base row (:609-610)
    def neighbours(self, cand, layer, include_deleted):
        out = []
        for (la, fr, to) in sorted(k for k in self.rows if k[0] == layer and k[1] == cand):
            if to == cand or (self.row_of is not None and self.row_of[to] == self.row_of[cand]):
                continue
            val = self.rows[(la, fr, to)]
            if include_deleted or not val[2]:
                out.append((to, val[0]))
        return out

    def v_dist(self, q, key):
        return self.dist(q, self.vec[key])

    # ---- hnsw_search_level, :539-587
    def search_level(self, q, ef, layer, found):
        visited = set(found)
        candidates = MinQueue()
        for key, pri in found.items():
            candidates.push(key, pri)
        while candidates:
            cand, cand_dist = candidates.pop()
            if cand_dist > found.peek()[1]:
                break
            for nb, _ in self.neighbours(cand, layer, False):
                if nb in visited:
                    continue
                d = self.v_dist(q, nb)
                if len(found) < ef or d < found.peek()[1]:
                    candidates.push(nb, d)
                    found.push(nb, d)
                    if len(found) > ef:
                        found.pop()
                visited.add(nb)

    # ---- hnsw_select_neighbours_heuristic, :470-538
    def select(self, q, found, m, layer):
        candidates, ret, discarded = MinQueue(), MinQueue(), MinQueue()
        for key, pri in found.items():
            candidates.push(key, pri)
        if self.extend:  # :499-511
            for item in list(found):
                for nb, _ in self.neighbours(item, layer, False):
                    candidates.push(nb, self.v_dist(q, nb))
        while candidates and len(ret) < m:  # :512-529
            cand, cand_dist = candidates.pop()
            add = True
            for existing in ret:
                if self.dist(self.vec[existing], self.vec[cand]) < cand_dist:
                    add = False
                    break
            if add:
                ret.push(cand, cand_dist)
            elif self.keep_pruned:
                discarded.push(cand, cand_dist)
        if self.keep_pruned:  # :530-536
            while discarded and len(ret) < m:
                ret.push(*discarded.pop())
        return ret

    # ---- hnsw_shrink_neighbour, :376-469
    def shrink(self, target, m, layer):
        vec = self.vec[target]
        candidates = MaxQueue()
        for nb, d in self.neighbours(target, layer, False):
            candidates.push(nb, d)
        new = self.select(vec, candidates, m, layer)
        old_set, new_set = set(candidates), set(new)
        for n, d in new.items():  # :413-433
            if n not in old_set:
                if n == target:
                    self.self_row_overwrites += 1
                self.rows[(layer, target, n)] = [d, None, False]
        for o, od in candidates.items():  # :434-466
            if o not in new_set:
                if self.rows[(layer, target, o)][2]:
                    del self.rows[(layer, target, o)]
                else:
                    self.rows[(layer, target, o)] = [od, None, True]
        return len(new)

    # ---- hnsw_put_vector, :155-375, for a key the index does not hold; `level` >= 0 is minus the layer drawn by :46-52
    def put(self, q, level):
        q = np.ascontiguousarray(q, dtype=np.float32)
        node = len(self.vec)
        self.vec.append(q)
        digest = hashlib.sha256(q.astype("<f4").tobytes()).digest()
        target_layer = -int(level)
        if not self.rows:  # :360-373
            for la in range(target_layer, 1):
                self.rows[(la, node, node)] = [0.0, digest, False]
            return node
        ep_row = min(self.rows)  # :184-191 the first row of the relation
        bottom, ep = ep_row[0], ep_row[1]
        found = MaxQueue()
        found.push(ep, self.v_dist(q, ep))
        if target_layer < bottom:  # :206-218
            for la in range(target_layer, bottom):
                self.rows[(la, node, node)] = [0.0, digest, False]
        for la in range(bottom, target_layer):  # :219-229
            self.search_level(q, 1, la, found)
        for la in range(max(target_layer, bottom), 1):  # :242-359
            m_max = self.m_max0 if la == 0 else self.m_max
            self.search_level(q, self.ef_c, la, found)
            nbrs = self.select(q, found, m_max, la)
            self.rows[(la, node, node)] = [float(len(nbrs)), digest, False]
            for nb, d in nbrs.items():
                self.rows[(la, node, nb)] = [d, None, False]
                self.rows[(la, nb, node)] = [d, None, False]
                self_val = list(self.rows[(la, nb, nb)])  # :330-337 read BEFORE the shrink
                degree = int(self_val[0]) + 1
                if degree > m_max:
                    degree = self.shrink(nb, m_max, la)
                self_val[0] = float(degree)
                self.rows[(la, nb, nb)] = self_val  # :352-357
        return node

    # ---- hnsw_knn, :869-1012 (the rows it returns as (node, distance); `accept(node)` stands for the filter bytecode)
    def knn(self, q, k, ef, radius=None, accept=None):
        q = np.ascontiguousarray(q, dtype=np.float32)
        if not self.rows:
            return []
        ep_row = min(self.rows)  # :891-899
        found = MaxQueue()
        found.push(ep_row[1], self.v_dist(q, ep_row[1]))
        for la in range(ep_row[0], 0):  # :919-929
            self.search_level(q, 1, la, found)
        self.search_level(q, ef, 0, found)  # :930-938
        if accept is None:  # :943-947
            while len(found) > k:
                found.pop()
        ret = []
        while found:  # :951-1004 farthest first
            node, d = found.pop()
            if radius is not None and d > radius:
                continue
            if accept is not None and not accept(node):
                continue
            ret.append((node, d))
        ret.reverse()
        return ret[:k]  # :1005-1006

    # ---- hnsw_remove_vec, :754-868 (without the canary row, which this model does not keep: the entry point is read off the
    # first row of the map, as every reader does)
    def remove(self, node):
        layer = 0
        while True:
            if (layer, node, node) not in self.rows:  # :766-778
                break
            del self.rows[(layer, node, node)]
            for nb, _ in self.neighbours(node, layer, True):  # :780-782 soft-deleted rows too; links inside the base row never
                self.rows.pop((layer, node, nb), None)  # :786-795
                self.rows.pop((layer, nb, node), None)  # :796-805 present or not
                self_val = self.rows.get((layer, nb, nb))
                if self_val is None:  # :806-815 `.get(..)?.unwrap()`: a link left dangling by an earlier removal leads here
                    raise ReferencePanic(f"layer {layer}: node {node} links to {nb}, which has no self row")
                self_val = list(self_val)
                self_val[0] -= 1.0  # :816-823
                self.rows[(layer, nb, nb)] = self_val
            layer -= 1

    def dangling(self):
        return sum(1 for (la, fr, to) in self.rows if fr != to and (la, to, to) not in self.rows)

    # ---- views for the comparison with the oracle
    def live_links(self, node, level):
        """what a reader that survives sees: the non-deleted links to nodes that still have their row on this layer"""
        return [to for to, _ in self.neighbours(node, -level, False) if (-level, to, to) in self.rows]

    def degree(self, node, level):
        return self.rows[(-level, node, node)][0]

    def top(self, node):
        return max(-k[0] for k in self.rows if k[1] == node and k[2] == node)

    def n_ignored(self):
        return sum(1 for k, v in self.rows.items() if k[1] != k[2] and v[2])

    def entry(self):
        return min(self.rows)[1]
