"""Island-level sharding of one physics world over ranks (one process per GPU) — host logic, SURVEY.md §8e level 1.

The unit of sharding is the *interaction island*: a connected component of (broad-phase pairs U manifolds U joints)
over non-static bodies (``avn_islands_partition``; static bodies never merge islands, like the reference's
``islands/mod.rs:822-834``, and are replicated on every rank).  Inside an island the greedy colouring, the contact
order and the broad-phase emission order do not depend on any other island, so a rank that owns whole islands
reproduces the single-world results for its bodies BIT FOR BIT and needs no data-path collective.  The only exchange
is :func:`exchange_bounds`: an all-gather of one AABB per rank per step (48 bytes) that detects islands of different
ranks coming into AABB contact — the trigger for a re-partition.

Sub-worlds keep the global relative order of bodies, colliders, manifolds and joints (a stable sub-sequence), which is
what makes the stable SAP sort, the pair emission order and the persistent colouring agree with the global run.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _ffi as F


@dataclass
class ShardPlan:
    world_size: int
    island_of_body: np.ndarray   # [N] int32, -1 = static
    rank_of_body: np.ndarray     # [N] int32, -1 = static (replicated)
    n_islands: int

    def local_bodies(self, rank: int) -> np.ndarray:
        """Global indices (ascending) of the bodies of `rank`'s sub-world: its islands + every static body."""
        return np.flatnonzero((self.rank_of_body == rank) | (self.rank_of_body < 0))

    def owned(self, rank: int) -> np.ndarray:
        return np.flatnonzero(self.rank_of_body == rank)


def plan(lib: F.Library, rb_type, position, edges: np.ndarray, world_size: int) -> ShardPlan:
    """edges: [E, 2] body index pairs (broad-phase pairs, manifolds, joints — any order)."""
    edges = np.asarray(edges).reshape(-1, 2)
    isl, rk, n = lib.islands_partition(rb_type, np.asarray(position, np.float64)[:, 0], edges[:, 0], edges[:, 1], world_size)
    return ShardPlan(world_size, isl, rk, n)


def _take(d: Dict[str, np.ndarray], idx: np.ndarray) -> Dict[str, np.ndarray]:
    return {k: (None if v is None else np.asarray(v)[idx]) for k, v in d.items()}


def split_bodies(p: ShardPlan, rank: int, bodies: Dict[str, np.ndarray]) -> Tuple[Dict[str, np.ndarray], np.ndarray, np.ndarray]:
    """Returns (sub-world body arrays, global index of every local body, global->local map with -1 = absent)."""
    loc = p.local_bodies(rank)
    g2l = np.full(len(p.rank_of_body), -1, np.int64)
    g2l[loc] = np.arange(len(loc))
    return _take(bodies, loc), loc, g2l


def split_colliders(g2l: np.ndarray, colliders: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    body = np.asarray(colliders["body"])
    keep = np.flatnonzero(g2l[body] >= 0)
    out = _take(colliders, keep)
    out["body"] = g2l[body[keep]].astype(np.int32)
    return out                        # entity_index stays GLOBAL: PairKeys are the same on every rank


def split_pairwise(g2l: np.ndarray, p: ShardPlan, rank: int, items: Dict[str, np.ndarray], extra: Optional[Dict[str, np.ndarray]] = None):
    """Manifolds or joints (anything with body1/body2): keep those whose non-static body belongs to `rank`."""
    b1 = np.asarray(items["body1"]); b2 = np.asarray(items["body2"])
    r1 = p.rank_of_body[b1]; r2 = p.rank_of_body[b2]
    owner = np.where(r1 >= 0, r1, r2)
    keep = np.flatnonzero(owner == rank)
    out = _take(items, keep)
    out["body1"] = g2l[b1[keep]].astype(np.int32)
    out["body2"] = g2l[b2[keep]].astype(np.int32)
    ex = None if extra is None else _take(extra, keep)
    return out, keep, ex


def merge_bodies(p: ShardPlan, n_bodies: int, per_rank: List[Tuple[np.ndarray, Dict[str, np.ndarray]]], template: Dict[str, np.ndarray]):
    """per_rank[r] = (global indices of rank r's local bodies, its bodies_download()).  Static bodies are taken
    from `template` (they never move); every other body from its owner."""
    out = {k: np.array(v, copy=True) for k, v in template.items()}
    for r, (loc, d) in enumerate(per_rank):
        mine = p.rank_of_body[loc] == r
        for k in out:
            out[k][loc[mine]] = d[k][mine]
    return out


def bounds_overlap(mn: np.ndarray, mx: np.ndarray) -> List[Tuple[int, int]]:
    """Rank pairs whose dynamic-body bounds intersect (closed intervals, like ColliderAabb::intersects,
    collider/mod.rs:539-544).  mn/mx: [R, 3]."""
    out = []
    R = len(mn)
    for a in range(R):
        for b in range(a + 1, R):
            if np.all(mn[a] <= mx[b]) and np.all(mx[a] >= mn[b]):
                out.append((a, b))
    return out


def exchange_bounds(world: F.World, dist=None, device=None):
    """The per-step exchange: all-gather this rank's dynamic bounds (call after UPDATE_AABB of the step).
    Returns (mins [R,3], maxs [R,3], overlapping rank pairs).  `dist` = torch.distributed (None = single rank)."""
    mn, mx = world.dynamic_bounds()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mn[None], mx[None], []
    import torch
    t = torch.tensor(np.concatenate([mn, mx]), dtype=torch.float64, device=device or "cpu")
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    a = torch.stack(outs).cpu().numpy()
    return a[:, :3], a[:, 3:], bounds_overlap(a[:, :3], a[:, 3:])
