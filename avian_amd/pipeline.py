"""Host side of the closed-loop contact pipeline for standalone drivers (tests, demos, benches).

In an Avian integration these structures are Avian's own (``ContactGraph``, ``IdPool``, ``ConstraintGraph``) and the Rust
shim forwards their changes through the ABI; a standalone driver has to keep them itself.  Everything here is integer
bookkeeping over the library's outputs, mirroring the reference:

* ``ContactGraph::add_edge_and_key_with`` + ``IdPool::alloc_id`` (lowest free id first), reference
  ``collision/contact_types/contact_graph.rs:521-566``, ``data_structures/id_pool.rs:31-40``: new broad-phase pairs, in
  emission order;
* the status-change processing of ``NarrowPhase::update`` (``collision/narrow_phase/system_param.rs:141-389``) in
  ascending contact id: removal of pairs whose AABBs separated, ``push_manifold`` / ``pop_manifold`` on the
  ``ConstraintGraph`` when colliders start / stop touching or start generating constraints;
* ``GraphColor::manifold_handles`` -> ``avn_manifold_handles_upload`` whenever a colour list changed.

Per step: ``UPDATE_AABB`` -> ``COLLECT_COLLISION_PAIRS`` -> (add pairs) -> ``NARROW_PHASE`` -> (status changes) -> ``SOLVER``.
Manifold data never leaves the device; the host sees new pairs and status changes only.
"""
from __future__ import annotations

import heapq
from typing import Dict, List

import numpy as np

from . import _ffi as F


class ContactPipeline:
    def __init__(self, world: F.World, lib: F.Library):
        self.w = world
        self.lib = lib
        self.graph = F.ConstraintGraph(lib)
        self.free_ids: List[int] = []          # IdPool: min-heap of freed ids
        self.next_id = 0
        self.pairs: Dict[int, tuple] = {}      # contact id -> (collider1, collider2, body1, body2)
        self.edge_touching: Dict[int, bool] = {}
        self.n_handles: Dict[int, int] = {}    # ContactEdge::constraint_handles.len()
        self.active: List[int] = []            # ContactGraph::active_pairs
        # the ContactGraph's adjacency (StableUnGraph): per collider its outgoing (collider1) / incoming (collider2) edges in INSERTION
        # order -- the reference links a new edge at the head of both lists, so its walks are these lists backwards
        self.out_edges: Dict[int, List[int]] = {}
        self.in_edges: Dict[int, List[int]] = {}
        self.handles_dirty = True
        self.stats = dict(pairs_added=0, pairs_removed=0, pushes=0, pops=0)

    def _alloc_id(self) -> int:
        if self.free_ids:
            return heapq.heappop(self.free_ids)
        i = self.next_id
        self.next_id += 1
        return i

    def add_new_pairs(self, pairs: np.ndarray):
        if len(pairs) == 0:
            return
        ids = np.array([self._alloc_id() for _ in range(len(pairs))], np.uint32)
        self.w.contact_pairs_add(ids, pairs["collider1"], pairs["collider2"], pairs["flags"])
        for i, p in zip(ids.tolist(), pairs):
            self.pairs[i] = (int(p["collider1"]), int(p["collider2"]), int(p["body1"]), int(p["body2"]))
            self.edge_touching[i] = False
            self.n_handles[i] = 0
            self.out_edges.setdefault(int(p["collider1"]), []).append(i)
            self.in_edges.setdefault(int(p["collider2"]), []).append(i)
        self.active.extend(ids.tolist())
        self.w.active_pairs_set(np.asarray(self.active, np.uint32))
        self.stats["pairs_added"] += len(ids)

    def _push(self, cid: int, flags: int):
        _, _, b1, b2 = self.pairs[cid]
        color = self.graph.push(cid, b1, b2, bool(flags & F.CP_STATIC1), bool(flags & F.CP_STATIC2))
        assert color >= 0
        self.n_handles[cid] += 1
        self.handles_dirty = True
        self.stats["pushes"] += 1

    def _pop_all(self, cid: int):
        for _ in range(self.n_handles[cid]):
            self.graph.pop(cid)
            self.stats["pops"] += 1
            self.handles_dirty = True
        self.n_handles[cid] = 0

    def process_status_changes(self):
        """system_param.rs:141-389, ascending contact id (the list arrives sorted)."""
        ch = self.w.contact_changes_get()
        removed = []
        for c in ch:
            cid, flags, dcount, count = int(c["contact_id"]), int(c["flags"]), int(c["manifold_count_change"]), int(c["manifold_count"])
            generates = bool(flags & F.CP_GENERATE_CONSTRAINTS)
            touching = bool(flags & F.CP_TOUCHING)
            if flags & F.CP_DISJOINT_AABB:
                if generates:
                    self._pop_all(cid)
                removed.append(cid)
            elif flags & F.CP_STARTED_TOUCHING:
                self.edge_touching[cid] = True
                if generates:
                    for _ in range(count):
                        self._push(cid, flags)
            elif flags & F.CP_STOPPED_TOUCHING:
                self.edge_touching[cid] = False
                if generates and self.n_handles[cid]:
                    self._pop_all(cid)
            elif touching and (flags & F.CP_STARTED_GENERATING_CONSTRAINTS):
                for _ in range(count):
                    self._push(cid, flags)
            elif touching and generates and dcount > 0:
                for _ in range(dcount):
                    self._push(cid, flags)
            elif touching and generates and dcount < 0:
                for _ in range(-dcount):
                    self.graph.pop(cid); self.n_handles[cid] -= 1; self.handles_dirty = True
        if removed:
            self.w.contact_pairs_remove(np.asarray(removed, np.uint32))
            gone = set(removed)
            self.active = [a for a in self.active if a not in gone]
            self.w.active_pairs_set(np.asarray(self.active, np.uint32))
            for cid in removed:
                c1, c2, _, _ = self.pairs[cid]
                self.out_edges[c1].remove(cid); self.in_edges[c2].remove(cid)
                del self.pairs[cid], self.edge_touching[cid], self.n_handles[cid]
                heapq.heappush(self.free_ids, cid)
            self.stats["pairs_removed"] += len(removed)
        if self.handles_dirty:
            offsets, handles = self.graph.lists()
            self.w.manifold_handles_upload(offsets, handles.astype(np.uint32))
            self.handles_dirty = False
        return len(ch)

    def remove_collider(self, entity: int):
        """``remove_collider`` of the reference (collision/narrow_phase/mod.rs:399-457 over ContactGraph::remove_collider_with,
        contact_types/contact_graph.rs:641-700, and StableUnGraph::remove_node_with, data_structures/stable_graph.rs:251-283): every edge
        of the collider -- outgoing edges newest first, then incoming edges newest first --; a touching pair's constraint handles are
        popped, then the edge leaves the graph, the pair set and the active pairs, and its id returns to the pool.  The caller uploads the
        remaining colliders afterwards (the interval is dropped in place)."""
        ids = list(reversed(self.out_edges.get(entity, []))) + list(reversed(self.in_edges.get(entity, [])))
        for cid in ids:
            if self.edge_touching[cid]:
                self._pop_all(cid)
            c1, c2, _, _ = self.pairs[cid]
            self.out_edges[c1].remove(cid); self.in_edges[c2].remove(cid)
            self.w.contact_pairs_remove(np.asarray([cid], np.uint32))
            self.active.remove(cid)
            del self.pairs[cid], self.edge_touching[cid], self.n_handles[cid]
            heapq.heappush(self.free_ids, cid)
            self.stats["pairs_removed"] += 1
        self.out_edges.pop(entity, None); self.in_edges.pop(entity, None)
        self.w.active_pairs_set(np.asarray(self.active, np.uint32))
        if self.handles_dirty:
            offsets, handles = self.graph.lists()
            self.w.manifold_handles_upload(offsets, handles.astype(np.uint32))
            self.handles_dirty = False

    def step(self):
        w = self.w
        w.run_system("UPDATE_AABB")
        w.run_system("COLLECT_COLLISION_PAIRS")
        self.add_new_pairs(w.pairs_get())
        w.run_system("NARROW_PHASE")
        n_changes = self.process_status_changes()
        w.run_system("SOLVER")
        return n_changes
