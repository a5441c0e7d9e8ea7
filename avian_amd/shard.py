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

import heapq
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


def plan_from_world(lib: F.Library, world: F.World, rb_type, position, world_size: int) -> ShardPlan:
    """The same plan from the islands the LIBRARY holds (avn_islands_get: connected components of the world's current constraint graph,
    computed on the device): the labels travel as star edges (body -> lowest body of its island), 4 bytes per body instead of the edge list."""
    lab, _ = world.islands_get()
    b = np.flatnonzero(lab != 0xFFFFFFFF)
    return plan(lib, rb_type, position, np.stack([b, lab[b].astype(np.int64)], axis=1), world_size)


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


# ---- level 1, second half: the re-partition that `exchange_bounds` triggers ---------------------------------------------------------------
#
# When two ranks' dynamic bounds intersect, a body of one may be about to touch a body of the other: a pair neither rank's broad phase
# can see.  The islands involved must be co-located BEFORE the frame's pair search.  What has to travel so that the run continues the
# single world bit for bit:
#   * the bodies' state (position, rotation, velocities; mass data) and their colliders;
#   * the contact pairs already known (the pair set that keeps the sweep from re-emitting them, and -- in emission order per island --
#     the input of the colouring);
#   * the persistent interval order of the sweep (reference broad_phase.rs:230-279: a stable insertion sort from LAST frame's order,
#     so ties in min.x are history): the ranks' orders are merged by the key they are sorted by (last sweep's min.x).  Ties between
#     colliders that come from different ranks are broken by ascending entity index -- the single world's order whenever the two have
#     never swapped; otherwise that history lived on no rank and the merged order may differ by a swap of equal keys (pair lists then
#     differ by the order of two pairs, never by content).
# The upload path is the existing ABI: a fresh world whose colliders are uploaded in the merged order starts from exactly that
# interval order (avn_colliders_upload appends new intervals in upload order), `avn_existing_pairs_upload` seeds the pair set.

@dataclass
class RankState:
    gid: np.ndarray                      # global body index of every local body, ascending (static bodies are on every rank)
    bodies: Dict[str, np.ndarray]        # the local bodies' CURRENT upload arrays (bodies_download merged over the original kwargs)
    colliders: Dict[str, np.ndarray]     # local collider arrays; `body` = LOCAL index, entity_index = global
    known: np.ndarray                    # [K, 2] global body pairs found so far, in this rank's emission order
    order: np.ndarray                    # entity index per interval: the rank's persistent sweep order
    order_key: np.ndarray                # min.x each interval had at the rank's last sweep (what `order` is sorted by); nan = never swept


def aabb_cross_pairs(states: List["RankState"], aabb: List[Tuple[np.ndarray, np.ndarray]], rb_static: np.ndarray) -> np.ndarray:
    """Global body pairs (a, b) of DIFFERENT ranks whose collider AABBs intersect now: the candidate edges that merge islands."""
    out = []
    R = len(states)
    for a in range(R):
        for b in range(a + 1, R):
            ca, cb = states[a].colliders, states[b].colliders
            ga = states[a].gid[np.asarray(ca["body"])]; gb = states[b].gid[np.asarray(cb["body"])]
            ka = np.flatnonzero(~rb_static[ga]); kb = np.flatnonzero(~rb_static[gb])
            if not len(ka) or not len(kb):
                continue
            mna, mxa = aabb[a][0][ka], aabb[a][1][ka]; mnb, mxb = aabb[b][0][kb], aabb[b][1][kb]
            hit = np.all(mna[:, None, :] <= mxb[None, :, :], axis=2) & np.all(mxa[:, None, :] >= mnb[None, :, :], axis=2)
            ia, ib = np.nonzero(hit)
            out.append(np.stack([ga[ka[ia]], gb[kb[ib]]], axis=1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros((0, 2), np.int64)


def merge_interval_orders(states: List["RankState"]) -> np.ndarray:
    """The global persistent interval order (entity indices) from the ranks' orders: k-way merge on (last sweep's min.x, entity index)
    that never reorders one rank's own list; colliders present on several ranks (static bodies) are kept once."""
    import heapq
    def stream(r):
        st = states[r]
        for pos, (e, k) in enumerate(zip(st.order.tolist(), st.order_key.tolist())):
            yield ((-np.inf if np.isnan(k) else k), e, r, pos)
    # heapq.merge needs every input sorted by the merge key: inside one rank the key is (min.x) only, so compare on that and break cross-rank
    # ties by entity; a rank's own equal-key run stays in its order because merge is stable per input
    merged, seen = [], set()
    heads = [stream(r) for r in range(len(states))]
    cur = [next(h, None) for h in heads]
    while any(c is not None for c in cur):
        best = None
        for r, c in enumerate(cur):
            if c is None:
                continue
            if best is None or (c[0], c[1]) < (cur[best][0], cur[best][1]):
                best = r
        e = cur[best][1]
        if e not in seen:
            seen.add(e); merged.append(e)
        cur[best] = next(heads[best], None)
    return np.asarray(merged, np.int64)


def repartition(lib: F.Library, states: List["RankState"], aabb: List[Tuple[np.ndarray, np.ndarray]], n_bodies: int, rb_type: np.ndarray,
                world_size: int, joints_edges: np.ndarray = None):
    """Every rank calls this with the SAME gathered inputs and gets the same answer: (plan, [RankState per rank]).  The new states hold the
    bodies' current data; `order` is the merged global order restricted to the rank (upload the colliders in that order)."""
    rb_type = np.asarray(rb_type)
    static = rb_type == F.RB_STATIC
    # global body / collider tables (a body's data comes from any rank that holds it: owners for moving bodies, anyone for static ones)
    fields = [k for k, v in states[0].bodies.items() if v is not None]
    glob_b = {k: None for k in fields}
    for st in states:
        for k in fields:
            v = np.asarray(st.bodies[k])
            if glob_b[k] is None:
                glob_b[k] = np.zeros((n_bodies,) + v.shape[1:], v.dtype)
            glob_b[k][st.gid] = v
    cfields = [k for k, v in states[0].colliders.items() if v is not None and k != "body"]
    n_col = 1 + max(int(np.max(st.colliders["entity_index"])) for st in states)
    glob_c = {k: None for k in cfields}
    col_body = np.full(n_col, -1, np.int64)
    for st in states:
        ent = np.asarray(st.colliders["entity_index"]).astype(np.int64)
        col_body[ent] = st.gid[np.asarray(st.colliders["body"])]
        for k in cfields:
            v = np.asarray(st.colliders[k])
            if glob_c[k] is None:
                glob_c[k] = np.zeros((n_col,) + v.shape[1:], v.dtype)
            glob_c[k][ent] = v
    known = np.concatenate([st.known.reshape(-1, 2) for st in states]).astype(np.int64)
    cross = aabb_cross_pairs(states, aabb, static)
    edges = [known, cross] + ([np.asarray(joints_edges).reshape(-1, 2)] if joints_edges is not None else [])
    pl = plan(lib, rb_type, glob_b["position"], np.concatenate(edges), world_size)
    order = lib.interval_orders_merge([st.order for st in states], [st.order_key for st in states])   # (numpy twin: merge_interval_orders)
    out = []
    for r in range(world_size):
        loc = pl.local_bodies(r)
        g2l = np.full(n_bodies, -1, np.int64); g2l[loc] = np.arange(len(loc))
        ents = order[g2l[col_body[order]] >= 0]                     # the rank's colliders in the merged order
        cols = {k: glob_c[k][ents] for k in cfields}
        cols["entity_index"] = ents.astype(np.uint32)
        cols["body"] = g2l[col_body[ents]].astype(np.int32)
        for k, v in states[0].colliders.items():
            if v is None:
                cols[k] = None
        bodies = {k: (glob_b[k][loc] if k in glob_b else None) for k in states[0].bodies}
        owner = np.where(static[known[:, 0]], pl.rank_of_body[known[:, 1]], pl.rank_of_body[known[:, 0]])
        out.append(RankState(loc, bodies, cols, known[owner == r], ents, np.full(len(ents), np.nan)))
    return pl, out, cross


# ---- x-slab sharding of the broad phase ("each GPU sorts/sweeps its slab (+halo)") -----------------------
#
# The sweep-and-prune emits pairs i-major over the intervals sorted by min.x: pair (i, j) belongs to the EARLIER interval i
# (collision/broad_phase.rs:400-475).  Cut the sorted order into contiguous slabs by min.x and give rank r
#   owned(r) = { c : s_r <= min.x(c) < s_{r+1} }                       -- the pairs whose earlier member lies here are r's
#   halo(r)  = { c : min.x(c) >= s_{r+1} and min.x(c) <= max over owned(r) of max.x }   -- everything an owned interval can reach
# as a sub-world that keeps the global relative order of its colliders (ascending upload index, so the stable sort and the
# candidate order of every owned interval are the global ones).  The rank sweeps the sub-world with the unchanged single-GPU
# broad phase, drops the pairs whose earlier member is a halo collider, and the concatenation of the ranks' lists in slab
# order IS the single-world list, bit for bit -- no merge, one all-gather of pair records per step.
# (A collider that spans the scene, e.g. a ground slab, has min.x in slab 0 and pulls every collider into rank 0's halo:
#  correct, unbalanced -- such colliders are better replicated and swept against each slab separately; not done here.)

@dataclass
class SlabPlan:
    world_size: int
    splits: np.ndarray          # [R + 1] slab boundaries on min.x (splits[0] = -inf, splits[R] = +inf)
    slab_of_collider: np.ndarray  # [C] int32


def slab_plan(aabb_min_x: np.ndarray, world_size: int) -> SlabPlan:
    """Balanced contiguous slabs of the sorted-by-min.x order.  Boundaries sit on values of min.x, and colliders with EQUAL
    min.x always share a slab (a tie never straddles a boundary, so "earlier in sorted order" never depends on the cut)."""
    x = np.asarray(aabb_min_x, np.float64)
    n = len(x)
    xs = np.sort(x, kind="stable")
    splits = np.full(world_size + 1, np.inf)
    splits[0] = -np.inf
    for r in range(1, world_size):
        splits[r] = xs[min(n - 1, (n * r) // world_size)] if n else np.inf
    splits = np.maximum.accumulate(splits)
    slab = (np.searchsorted(splits, x, side="right") - 1).astype(np.int32)   # splits[r] <= x < splits[r + 1]
    return SlabPlan(world_size, splits, np.clip(slab, 0, world_size - 1))


def slab_colliders(p: SlabPlan, rank: int, aabb_min_x: np.ndarray, aabb_max_x: np.ndarray, prev_order: np.ndarray = None) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (local: global collider indices of rank's sub-world = owned + halo, owned_mask over `local`).  `local` is in the
    single world's PERSISTENT interval order when `prev_order` is given (see slab_next_order), else ascending."""
    mn = np.asarray(aabb_min_x, np.float64); mx = np.asarray(aabb_max_x, np.float64)
    owned = p.slab_of_collider == rank
    if not owned.any():
        return np.zeros(0, np.int64), np.zeros(0, bool)
    reach = mx[owned].max()
    halo = (p.slab_of_collider > rank) & (mn <= reach)
    member = owned | halo
    if prev_order is None:
        local = np.flatnonzero(member)
    else:
        po = np.asarray(prev_order, np.int64)
        seen = np.zeros(len(member), bool); seen[po] = True
        local = np.concatenate([po[member[po]], np.flatnonzero(member & ~seen)])   # (colliders new this frame: appended at the end, broad_phase.rs:296-315)
    return local, owned[local]


def slab_next_order(prev_order: np.ndarray, aabb_min_x: np.ndarray, n_colliders: int) -> np.ndarray:
    """The single world's AabbIntervals order AFTER this frame's sort, from the order before it: the reference's insertion sort is
    stable (broad_phase.rs:479-487), so equal min.x keep the PREVIOUS frame's relative order -- not the upload order.  Every rank
    evaluates this on the replicated AABB extents (-0.0 == +0.0; intervals with a non-finite key are dropped like
    update_aabb_intervals drops them, :230-279), so the slab sub-worlds can be seeded with the order the single world would have.
    prev_order = None: first frame (upload order)."""
    x = np.asarray(aabb_min_x, np.float64) + 0.0
    po = np.arange(n_colliders, dtype=np.int64) if prev_order is None else np.asarray(prev_order, np.int64)
    seen = np.zeros(n_colliders, bool); seen[po] = True
    po = np.concatenate([po, np.flatnonzero(~seen)])
    po = po[np.isfinite(x[po])]
    return po[np.argsort(x[po], kind="stable")]


def slab_filter_pairs(pairs: np.ndarray, entity_index: np.ndarray, owned_mask: np.ndarray) -> np.ndarray:
    """Keep the pairs whose EARLIER member (avn_pair.collider1, broad_phase.rs:443) is owned.  entity_index / owned_mask:
    the sub-world's colliders."""
    if len(pairs) == 0:
        return pairs
    order = np.argsort(entity_index, kind="stable")
    pos = order[np.searchsorted(entity_index[order], pairs["collider1"])]
    return pairs[owned_mask[pos]]


def gather_pairs(pairs: np.ndarray, dist=None, device=None) -> np.ndarray:
    """The per-step exchange of the slab-sharded broad phase: all-gather of the ranks' (collider1, collider2, flags) records,
    concatenated in rank (= slab) order.  Returns a [P, 3] uint32 array.  `dist` = torch.distributed (None = single rank)."""
    rec = np.stack([pairs["collider1"], pairs["collider2"], pairs["flags"]], axis=1).astype(np.uint32) if len(pairs) else np.zeros((0, 3), np.uint32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    import torch
    dev = device or "cpu"
    world = dist.get_world_size()
    cnt = torch.tensor([len(rec)], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    sizes = [int(c.item()) for c in cnts]
    cap = max(max(sizes), 1)
    buf = torch.zeros((cap, 3), dtype=torch.int64, device=dev)   # (int64 carrier: gloo and RCCL both move it; values are uint32)
    if len(rec):
        buf[: len(rec)] = torch.from_numpy(rec.astype(np.int64)).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return np.concatenate([o[:s].cpu().numpy().astype(np.uint32) for o, s in zip(outs, sizes)], axis=0)


def slab_subworld(bodies: Dict[str, np.ndarray], colliders: Dict[str, np.ndarray], local: np.ndarray):
    """Body / collider arrays of the sub-world made of the colliders `local` (ascending global collider indices) and the bodies
    they belong to (ascending global body index; ColliderOf.body re-indexed).  Returns (bodies, colliders, global body indices)."""
    body = np.asarray(colliders["body"])[local]
    ub = np.unique(body[body >= 0])
    g2l = np.full(len(np.asarray(bodies["rb_type"])), -1, np.int64)
    g2l[ub] = np.arange(len(ub))
    cols = _take(colliders, local)
    cols["body"] = np.where(body >= 0, g2l[np.maximum(body, 0)], -1).astype(np.int32)
    return _take(bodies, ub), cols, ub


def pair_keys(rec: np.ndarray) -> np.ndarray:
    """PairKey (data_structures/pair_key.rs: smaller entity index in the high word) of [P, >= 2] (collider1, collider2) records."""
    a = rec[:, 0].astype(np.uint64); b = rec[:, 1].astype(np.uint64)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    return (lo << np.uint64(32)) | hi


def slab_broad_phase_step(lib: F.Library, bits: int, bodies: Dict[str, np.ndarray], colliders: Dict[str, np.ndarray], aabb_min_x: np.ndarray,
                          aabb_max_x: np.ndarray, known_keys: np.ndarray, rank: int, world_size: int, dist=None, device=None,
                          prev_order: np.ndarray = None) -> np.ndarray:
    """One frame of the slab-sharded broad phase on this rank: plan the slabs from the (replicated) AABB extents, sweep this
    rank's slab + halo with the single-GPU broad phase, drop halo-owned pairs, all-gather.  known_keys = PairKeys already in
    the contact graph (globally).  Returns the step's NEW pairs of the whole world, [P, 3] uint32, in the single-world order.
    prev_order: the single world's interval order BEFORE this frame (slab_next_order of the previous frame; None on the first
    frame).  The sub-world is uploaded in that order, so ties in min.x break as in the persistent single world -- with None
    they break by upload index, which is the single world's behaviour only on its first frame."""
    # the planner behind the ABI (avn_slab_select, host C++); slab_plan / slab_colliders / slab_next_order above are the same rule in numpy,
    # kept as its independent check (tests/test_planners_cpu.py)
    local, owned, _ = lib.slab_select(aabb_min_x, aabb_max_x, prev_order, world_size, rank, want_next=False)
    mine = np.zeros(0, PAIR_DTYPE_LOCAL)
    if len(local):
        b, c, _ = slab_subworld(bodies, colliders, local)
        w = F.World(lib, F.default_config(bits, substeps=1))
        try:
            w.bodies_upload(**b); w.colliders_upload(**c); w.existing_pairs_upload(np.asarray(known_keys, np.uint64))
            w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
            mine = slab_filter_pairs(w.pairs_get(), np.asarray(c["entity_index"]), owned)
        finally:
            w.close()
    return gather_pairs(mine, dist, device)


PAIR_DTYPE_LOCAL = F.PAIR_DTYPE


# ---- level 2: ONE island split over several worlds (x-slabs) with a global colouring (include/avian_mi355x.h, avn_halo_plan) ----------
#
# Ownership: a non-static body belongs to the x-slab its position falls into; a manifold belongs to the owner of body1 (body2 when body1
# is static).  A world holds its owned manifolds, the bodies they touch, its owned bodies and all static bodies.  After colour c of a
# contact pass the world that solved the (only) manifold of colour c on a shared body sends that body's velocities to the other holders.

@dataclass
class Level2Rank:
    bodies: np.ndarray            # global body indices of this world, ascending (local index = position)
    manifolds: np.ndarray         # global manifold indices (colour-major order of the global set)
    color_offsets: np.ndarray     # [25] local colour offsets
    peers: np.ndarray             # ranks this world exchanges with
    send_offsets: np.ndarray      # [(23 + n_overflow_levels) * n_peers + 1]: exchange SLOT-major (slot c < 23 = colour c, slot 23 + l = overflow level l)
    send_bodies: np.ndarray       # LOCAL body indices
    recv_offsets: np.ndarray
    recv_bodies: np.ndarray
    n_overflow_levels: int = 1    # > 1: the overflow colour is cut into levels of the GLOBAL list (an overflow manifold touches a shared body)
    overflow_level: np.ndarray = None   # [this world's overflow manifolds, local order] level of each
    joints: np.ndarray = None     # global joint indices this world owns (ascending: the global solve order restricted); None: the plan was made without joints
    joint_slot: bool = False      # one more exchange slot behind the colours and levels: the SolverBody records of shared bodies a joint component moved
    global_joints: bool = False   # the unsplit world holds joints: every world runs the XPBD snapshot / velocity projection over its bodies as the unsplit one does

    @property
    def n_slots(self) -> int:
        return F.COLOR_OVERFLOW_INDEX + self.n_overflow_levels + (1 if self.joint_slot else 0)

    @property
    def joint_slot_index(self) -> int:
        return F.COLOR_OVERFLOW_INDEX + self.n_overflow_levels

    def solve_order(self):
        """Exchange slots in solve order: the overflow colour first (solver/plugin.rs:461-467), level by level, then colours 0..22."""
        return [F.COLOR_OVERFLOW_INDEX + l for l in range(self.n_overflow_levels)] + list(range(F.COLOR_OVERFLOW_INDEX))

    def upload(self, world):
        """avn_halo_overflow_levels_upload (when levelled) + avn_halo_plan_upload."""
        if self.n_overflow_levels > 1:
            world.halo_overflow_levels_upload(self.n_overflow_levels, self.overflow_level)
        if self.global_joints or getattr(world, "_halo_joint", False):   # (also back to "no joint slot" on a world that carried one)
            world.halo_joint_slot_set(self.joint_slot, self.global_joints)
        world.halo_plan_upload(self.peers, self.send_offsets, self.send_bodies, self.recv_offsets, self.recv_bodies)


def level2_plan_lib(lib: F.Library, position: np.ndarray, rb_type: np.ndarray, body1: np.ndarray, body2: np.ndarray, color_offsets: np.ndarray,
                    world_size: int, joints=None) -> List[Level2Rank]:
    """The plan from the library's own planner (avn_level2_plan_*: what a host in any language calls).  `level2_plan` below is the same
    rule in numpy, kept as its independent check (tests/test_level2_cpu.py)."""
    out = []
    try:
        ranks = lib.level2_plan(rb_type, np.asarray(position, np.float64)[:, 0], body1, body2, color_offsets, world_size, joints=joints)
    except F.AvnError as e:
        raise ValueError(str(e))
    for k in ranks:
        out.append(Level2Rank(k["bodies"], k["manifolds"], k["color_offsets"], k["peers"], k["send_offsets"], k["send_bodies"], k["recv_offsets"], k["recv_bodies"],
                              k["n_overflow_levels"], k["overflow_level"], k.get("joints"), bool(k.get("joint_slot", False)), bool(k.get("global_joints", False))))
    return out


def level2_plan(position: np.ndarray, rb_type: np.ndarray, body1: np.ndarray, body2: np.ndarray, color_offsets: np.ndarray, world_size: int,
                has_solver_body: np.ndarray = None, joints=None) -> List[Level2Rank]:
    """body1 / body2: the GLOBAL colour-major manifold set; color_offsets: its [25] offsets (the reference's colouring of the whole island).
    joints = (joint body1 [J], joint body2 [J], joint_type [J], damped: bool) or None.  The reference walks the joints of a type serially (xpbd/plugin.rs:145-189,
    joint_damping solver/plugin.rs:756-830), so a joint COMPONENT -- joints linked through non-static bodies, and, with JointDamping, through the type's DUMMY pair
    that stands in for bodies without a SolverBody (plugin.rs:766-767) -- is the unit of ownership: it belongs to the slab of its lowest non-static body, its owner
    holds all its bodies, and after the joint systems of a substep the owner sends the SolverBody records (delta position / rotation, velocities) of the component's
    SHARED bodies to their other holders: the joint slot, one exchange per substep."""
    position = np.asarray(position, np.float64); rb_type = np.asarray(rb_type)
    b1 = np.asarray(body1, np.int64); b2 = np.asarray(body2, np.int64)
    offs = np.asarray(color_offsets, np.int64)
    n = len(rb_type)
    static = rb_type == F.RB_STATIC
    moving = ~static if has_solver_body is None else np.asarray(has_solver_body, bool)
    dyn = np.flatnonzero(~static)
    xs = np.sort(position[dyn, 0], kind="stable")
    cuts = [xs[min(len(xs) - 1, (len(xs) * r) // world_size)] for r in range(1, world_size)]
    owner = np.full(n, -1, np.int64)
    owner[dyn] = np.searchsorted(np.asarray(cuts), position[dyn, 0], side="right")
    m_owner = np.where(static[b1], owner[b2], owner[b1])
    color_of = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
    held = [np.zeros(n, bool) for _ in range(world_size)]
    m_of = []
    for r in range(world_size):
        mine = np.flatnonzero(m_owner == r)
        m_of.append(mine)
        held[r][static] = True; held[r][owner == r] = True
        held[r][b1[mine]] = True; held[r][b2[mine]] = True
    j_owner = None
    jointed = np.zeros(n, bool)
    if joints is not None and len(joints[0]):
        jb1, jb2, jt = np.asarray(joints[0], np.int64), np.asarray(joints[1], np.int64), np.asarray(joints[2], np.int64)
        damped = bool(joints[3])
        parent = np.arange(n + 2 * F.JOINT_TYPE_COUNT)

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        node = np.full((len(jb1), 2), -1, np.int64)
        for j in range(len(jb1)):
            a, b, t = int(jb1[j]), int(jb2[j]), int(jt[j])
            na = (n + 2 * t if damped else -1) if static[a] else a
            nb = (n + 2 * t + 1 if damped else -1) if static[b] else b
            node[j] = (na, nb)
            if na >= 0 and nb >= 0:
                ra, rb_ = find(na), find(nb)
                if ra != rb_:
                    parent[max(ra, rb_)] = min(ra, rb_)
            for x in (a, b):
                if not static[x]:
                    jointed[x] = True
        comp_owner = {}
        for b in np.flatnonzero(jointed).tolist():     # ascending: the first body met is the component's lowest
            comp_owner.setdefault(find(b), int(owner[b]))
        j_owner = np.array([comp_owner.get(find(int(max(node[j]))), 0) if max(node[j]) >= 0 else 0 for j in range(len(jb1))], np.int64)
        for b in np.flatnonzero(jointed).tolist():
            held[comp_owner[find(b)]][b] = True
        body_comp_owner = {b: comp_owner[find(b)] for b in np.flatnonzero(jointed).tolist()}
    held = np.stack(held)                                 # [R, n]
    shared = moving & (held.sum(0) > 1)
    # Exchange slots: colours 0..22, then the overflow colour -- one slot, or (when one of its manifolds touches a shared body) one slot per LEVEL of the global
    # list: the reference walks it serially (solver/plugin.rs:461-467) and only the relative order of manifolds sharing a body matters, so
    # level(m) = the number of overflow manifolds in front of m on the deepest chain through its non-static bodies; manifolds of one level share no body.
    n_col = len(offs) - 2
    o0, o1 = int(offs[n_col]), int(offs[n_col + 1])
    touches = shared[b1] | shared[b2]                     # (only manifolds on a shared body take part in the exchange)
    level = np.zeros(o1 - o0, np.int64)
    n_levels = 1
    if touches[o0:o1].any():
        depth = np.zeros(n, np.int64)
        for m in range(o0, o1):
            bb = [b for b in (int(b1[m]), int(b2[m])) if not static[b]]
            d = max((int(depth[b]) for b in bb), default=0)
            level[m - o0] = d
            for b in bb:
                depth[b] = d + 1
            n_levels = max(n_levels, d + 1)
    joint_slot = bool(j_owner is not None and (shared & jointed).any())
    n_slots = n_col + n_levels + (1 if joint_slot else 0)
    sends = [[{} for _ in range(n_slots)] for _ in range(world_size)]   # sends[s][slot][r] = [global body ...]
    if joint_slot:
        for b in np.flatnonzero(shared & jointed).tolist():
            s = body_comp_owner[b]
            for r in np.flatnonzero(held[:, b]):
                if r != s:
                    sends[s][n_slots - 1].setdefault(int(r), []).append(b)
    for c in range(n_col + 1):
        for m in (np.flatnonzero(touches[offs[c]:offs[c + 1]]) + offs[c]).tolist():
            s = int(m_owner[m])
            slot = c if c < n_col else n_col + int(level[m - o0])
            for b in (int(b1[m]), int(b2[m])):
                if not shared[b]:
                    continue
                for r in np.flatnonzero(held[:, b]):
                    if r != s:
                        sends[s][slot].setdefault(int(r), []).append(b)
    out = []
    for r in range(world_size):
        bodies = np.flatnonzero(held[r])
        g2l = np.full(n, -1, np.int64); g2l[bodies] = np.arange(len(bodies))
        peers = sorted({p for c in range(n_slots) for p in sends[r][c]} | {s for s in range(world_size) if s != r and any(r in sends[s][c] for c in range(n_slots))})
        so, sb, ro, rb = [0], [], [0], []
        for c in range(n_slots):
            for p in peers:
                lst = sorted(sends[r][c].get(p, [])); sb.extend(g2l[lst].tolist()); so.append(len(sb))
                lst = sorted(sends[p][c].get(r, [])); rb.extend(g2l[lst].tolist()); ro.append(len(rb))
        mine = m_of[r]
        local_offs = np.concatenate([[0], np.cumsum(np.bincount(color_of[mine], minlength=len(offs) - 1))])
        out.append(Level2Rank(bodies, mine, local_offs.astype(np.uint32), np.asarray(peers, np.int32), np.asarray(so if peers else [0], np.uint32), np.asarray(sb, np.int32),
                              np.asarray(ro if peers else [0], np.uint32), np.asarray(rb, np.int32), n_levels, level[mine[mine >= o0] - o0].astype(np.uint32),
                              np.flatnonzero(j_owner == r) if j_owner is not None else (np.zeros(0, np.int64) if joints is not None else None), joint_slot, j_owner is not None))
    return out


def level2_local_manifolds(rank: Level2Rank, manifolds: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The rank's rows of a global manifold dict (arrays whose first dimension is the manifold count), bodies re-indexed."""
    n = len(manifolds["body1"])
    bodies = np.asarray(rank.bodies, np.int64)
    g2l = np.full(int(bodies.max()) + 1 if len(bodies) else 1, -1, np.int64); g2l[bodies] = np.arange(len(bodies))
    out = {}
    for k, v in manifolds.items():
        a = np.asarray(v) if v is not None else None
        out[k] = a[rank.manifolds] if a is not None and a.ndim >= 1 and len(a) == n else v
    out["body1"] = g2l[np.asarray(out["body1"], np.int64)].astype(np.int32)
    out["body2"] = g2l[np.asarray(out["body2"], np.int64)].astype(np.int32)
    assert (out["body1"] >= 0).all() and (out["body2"] >= 0).all(), "a rank's manifold names a body the rank does not hold"
    return out


CONTACT_PASSES = ("WARM_START", "SOLVE_CONTACTS_BIAS", "SOLVE_CONTACTS_RELAX", "SOLVE_RESTITUTION")


def level2_exchange_slot(world, rank: Level2Rank, c: int, exchange):
    np_ = len(rank.peers)
    if np_ == 0:
        return
    out = {p: world.halo_pack(c, p) for p in range(np_) if rank.send_offsets[c * np_ + p + 1] > rank.send_offsets[c * np_ + p]}
    need = [p for p in range(np_) if rank.recv_offsets[c * np_ + p + 1] > rank.recv_offsets[c * np_ + p]]
    got = exchange(c, out, need)
    for p in need:
        world.halo_unpack(c, p, got[p])


def level2_pass(world, rank: Level2Rank, system: str, exchange):
    """One contact pass, slot by slot in solve order (the overflow colour first -- level by level when the planner cut it --, then colours 0..22), with the halo
    exchange after every slot.  exchange(slot, {peer_index: records to send}) -> {peer_index: records received}."""
    for c in rank.solve_order():
        world.run_color_pass(system, c)
        if len(rank.peers) == 0:
            continue
        np_ = len(rank.peers)
        out = {p: world.halo_pack(c, p) for p in range(np_) if rank.send_offsets[c * np_ + p + 1] > rank.send_offsets[c * np_ + p]}
        need = [p for p in range(np_) if rank.recv_offsets[c * np_ + p + 1] > rank.recv_offsets[c * np_ + p]]
        got = exchange(c, out, need)
        for p in need:
            world.halo_unpack(c, p, got[p])


def level2_solver(world, rank: Level2Rank, substeps: int, exchange, restitution: bool = True, solver_iterations: int = 1, has_joints: bool = False):
    """SolverSystems in the reference's order (solver/schedule.rs:32-69) with the contact passes split by colour: what avn_step does
    inside the library once a communicator is set; here the transport is the caller's (`exchange`).  `solver_iterations` is the
    declared extension of avn_config: like the library's level2_substeps the biased solve and the relax pass run that many times per
    substep.  The library also repeats the joint pass WITHOUT re-taking the XPBD snapshot, which avn_run_system cannot express: with
    joints and more than one iteration only the library-issued transport (avn_comm_init) is valid."""
    assert solver_iterations >= 1
    assert solver_iterations == 1 or not has_joints, "joints with solver_iterations > 1: use the library transport (avn_comm_init)"
    for s in ("PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"):
        world.run_system(s)
    for _ in range(substeps):
        world.run_system("INTEGRATE_VELOCITIES")
        level2_pass(world, rank, "WARM_START", exchange)
        for _it in range(solver_iterations):
            level2_pass(world, rank, "SOLVE_CONTACTS_BIAS", exchange)
        world.run_system("INTEGRATE_POSITIONS")
        for _it in range(solver_iterations):
            level2_pass(world, rank, "SOLVE_CONTACTS_RELAX", exchange)
        for s in ("XPBD_SOLVE", "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING"):
            world.run_system(s)
        if rank.joint_slot:   # the joint components' shared bodies: SolverBody records from the component's owner to the other holders
            level2_exchange_slot(world, rank, rank.joint_slot_index, exchange)
    world.run_system("CLEAR_VELOCITY_INCREMENTS")
    if restitution:
        level2_pass(world, rank, "SOLVE_RESTITUTION", exchange)
    world.run_system("WRITEBACK_SOLVER_BODIES")
    world.run_system("STORE_CONTACT_IMPULSES")


# ---- the closed loop sharded by islands: replicated integer bookkeeping, sharded physics (round 4) -------------------------------------
class ShardedClosedLoop:
    """One rank of a closed loop whose ISLANDS are spread over ranks and whose results equal the single world's bit for bit.

    Why the integer bookkeeping cannot be per rank: ``IdPool`` hands out the LOWEST free ContactId
    (data_structures/id_pool.rs:31-40) in the broad phase's global emission order, the status loop walks changes in ascending id
    (collision/narrow_phase/system_param.rs:141-145), and ``pop_manifold``'s ``swap_remove`` moves the LAST handle of a colour's list into the
    hole (dynamics/solver/constraint_graph.rs:245-296) -- an island's colours and the order of its overflow manifolds depend on what OTHER
    islands freed and popped.  So every rank replays EVERYTHING that is integer -- the global interval order (a stable sort of all colliders'
    min.x keys), the ContactIds, the ConstraintGraph with all colour lists -- from three small all-gathers per step (4 bytes per collider,
    20 bytes per new pair, 16 bytes per status change), and runs only its own islands' physics: broad phase, narrow phase and solver of its
    sub-world through the low-level ABI.  Its handle lists are the global colour lists restricted to its own pairs (a restriction keeps the
    relative order, which is all the overflow colour's serial solve needs).

    A step is three phases with an exchange after the first two: ``phase1() -> payload``, ``phase2(all payloads) -> payload``,
    ``phase3(all payloads)``; ``step(all_gather)`` runs them over a communicator (``all_gather(obj) -> list of every rank's obj``).
    Bodies of different ranks must not come into AABB contact (the level-1 re-partition handles that: ``exchange_bounds`` / ``repartition``).
    """

    def __init__(self, lib: F.Library, world: F.World, p: ShardPlan, rank: int, colliders: Dict[str, np.ndarray], rb_type: np.ndarray):
        self.lib, self.w, self.plan, self.rank = lib, world, p, rank
        self.loc = p.local_bodies(rank)                    # local body -> global body
        ent = np.asarray(colliders["entity_index"], np.uint32)
        body = np.asarray(colliders["body"])
        self.n_colliders = len(ent)
        self.global_of_entity = {int(e): i for i, e in enumerate(ent)}    # global collider slot (upload order) of an entity
        # every collider's min.x key is contributed by ONE rank: its body's owner, static bodies by rank 0
        owner = p.rank_of_body[body]
        self.my_colliders = np.flatnonzero((owner == rank) | ((owner < 0) & (rank == 0)))
        loc_ent = ent[np.flatnonzero((owner == rank) | (owner < 0))]      # the sub-world's colliders, in its upload order
        self.local_slot_of_entity = {int(e): i for i, e in enumerate(loc_ent)}
        self.entity = ent
        self.order = np.arange(self.n_colliders)           # the global AabbIntervals order (broad_phase.rs:177-185), replicated
        self.graph = F.ConstraintGraph(lib, len(rb_type))  # the global ConstraintGraph, replicated
        self.free_ids: List[int] = []
        self.next_id = 0
        self.pairs: Dict[int, tuple] = {}                  # every rank's pairs: id -> (collider1, collider2, global body1, global body2, owner rank)
        self.n_handles: Dict[int, int] = {}
        self.active: List[int] = []                        # this rank's active pairs
        self.stats = dict(pairs_added=0, pairs_removed=0, pushes=0, pops=0)

    # -- phase 1: local AABBs + local broad phase ---------------------------------------------------------------------------------------
    def phase1(self):
        w = self.w
        w.run_system("UPDATE_AABB")
        w.run_system("COLLECT_COLLISION_PAIRS")
        mn, _, _ = w.aabbs_download()
        mine = self.my_colliders
        keys = np.array([mn[self.local_slot_of_entity[int(self.entity[g])], 0] for g in mine], np.float64)
        pr = w.pairs_get()
        new = np.zeros(len(pr), [("c1", "<u4"), ("c2", "<u4"), ("b1", "<i8"), ("b2", "<i8"), ("flags", "<u4")])
        new["c1"], new["c2"], new["flags"] = pr["collider1"], pr["collider2"], pr["flags"]
        new["b1"], new["b2"] = self.loc[pr["body1"]], self.loc[pr["body2"]]
        return dict(rank=self.rank, colliders=mine, keys=keys, pairs=new)

    # -- phase 2: the global interval order, ids in the global emission order, the local narrow phase -------------------------------------
    def phase2(self, payloads):
        minx = np.zeros(self.n_colliders)
        for pl in payloads:
            minx[pl["colliders"]] = pl["keys"]
        # sweep_and_prune's insertion sort (broad_phase.rs:373-387, 479-487) is a STABLE sort of last frame's order by this frame's min.x
        self.order = self.order[np.argsort(minx[self.order], kind="stable")]
        gpos = np.empty(self.n_colliders, np.int64)
        gpos[self.order] = np.arange(self.n_colliders)
        recs = []
        for pl in payloads:
            for q in pl["pairs"]:
                recs.append((int(gpos[self.global_of_entity[int(q["c1"])]]), int(gpos[self.global_of_entity[int(q["c2"])]]), int(q["c1"]), int(q["c2"]), int(q["b1"]), int(q["b2"]),
                             int(q["flags"]), pl["rank"]))
        recs.sort(key=lambda r: (r[0], r[1]))   # pairs are emitted i-major over the sorted intervals, j ascending (broad_phase.rs:387-388)
        ids, c1, c2, fl = [], [], [], []
        for (p1, p2, a, b, gb1, gb2, flags, owner) in recs:
            assert p1 < p2, "collider1 is the earlier interval"
            cid = heapq.heappop(self.free_ids) if self.free_ids else self._fresh()
            self.pairs[cid] = (a, b, gb1, gb2, owner)
            self.n_handles[cid] = 0
            if owner == self.rank:
                ids.append(cid); c1.append(a); c2.append(b); fl.append(flags)
        self.stats["pairs_added"] += len(recs)
        if ids:
            self.w.contact_pairs_add(np.asarray(ids, np.uint32), np.asarray(c1, np.uint32), np.asarray(c2, np.uint32), np.asarray(fl, np.uint32))
            self.active.extend(ids)
        self.w.active_pairs_set(np.asarray(self.active, np.uint32))
        self.w.run_system("NARROW_PHASE")
        return dict(rank=self.rank, changes=self.w.contact_changes_get().copy())

    def _fresh(self):
        i = self.next_id
        self.next_id += 1
        return i

    # -- phase 3: every rank's status changes in ascending id on the replicated graph; the local handle lists; the local solver ------------
    def phase3(self, payloads):
        ch = np.concatenate([pl["changes"] for pl in payloads]) if payloads else np.zeros(0, F.CHANGE_DTYPE)
        ch = ch[np.argsort(ch["contact_id"], kind="stable")]
        removed, removed_local = [], []
        for c in ch:
            cid, flags, dcount, count = int(c["contact_id"]), int(c["flags"]), int(c["manifold_count_change"]), int(c["manifold_count"])
            generates, touching = bool(flags & F.CP_GENERATE_CONSTRAINTS), bool(flags & F.CP_TOUCHING)
            _, _, gb1, gb2, owner = self.pairs[cid]
            def push(n):
                for _ in range(n):
                    col = self.graph.push(cid, gb1, gb2, bool(flags & F.CP_STATIC1), bool(flags & F.CP_STATIC2))
                    assert col >= 0
                    self.n_handles[cid] += 1; self.stats["pushes"] += 1
            def pop(n):
                for _ in range(n):
                    self.graph.pop(cid); self.n_handles[cid] -= 1; self.stats["pops"] += 1
            if flags & F.CP_DISJOINT_AABB:
                if generates:
                    pop(self.n_handles[cid])
                removed.append(cid)
                if owner == self.rank:
                    removed_local.append(cid)
            elif flags & F.CP_STARTED_TOUCHING:
                if generates:
                    push(count)
            elif flags & F.CP_STOPPED_TOUCHING:
                if generates and self.n_handles[cid]:
                    pop(self.n_handles[cid])
            elif touching and (flags & F.CP_STARTED_GENERATING_CONSTRAINTS):
                push(count)
            elif touching and generates and dcount > 0:
                push(dcount)
            elif touching and generates and dcount < 0:
                pop(-dcount)
        if removed_local:
            self.w.contact_pairs_remove(np.asarray(removed_local, np.uint32))
            gone = set(removed_local)
            self.active = [a for a in self.active if a not in gone]
            self.w.active_pairs_set(np.asarray(self.active, np.uint32))
        for cid in removed:
            del self.pairs[cid], self.n_handles[cid]
            heapq.heappush(self.free_ids, cid)
        self.stats["pairs_removed"] += len(removed)
        # GraphColor::manifold_handles restricted to this rank's pairs, order kept
        offsets, handles = self.graph.lists()
        mine = np.fromiter((self.pairs[int(h)][4] == self.rank for h in handles), bool, len(handles))
        loc_off = np.zeros(F.GRAPH_COLOR_COUNT + 1, np.uint32)
        loc_off[1:] = np.cumsum([int(mine[offsets[c]:offsets[c + 1]].sum()) for c in range(F.GRAPH_COLOR_COUNT)])
        self.w.manifold_handles_upload(loc_off, handles[mine].astype(np.uint32))
        self.global_lists = (offsets, handles.astype(np.uint32))
        self.w.run_system("SOLVER")
        return len(ch)

    def step(self, all_gather):
        return self.phase3(all_gather(self.phase2(all_gather(self.phase1()))))


class ShardedClosedLoopNative:
    """One rank of the closed loop sharded by islands with the replicated bookkeeping in the LIBRARY (``avn_shard_*``, host C++: round 5) -- the same
    three phases as :class:`ShardedClosedLoop` (kept as the readable model and second opinion), but the payloads are FLAT arrays (no pickled objects)
    and no per-pair Python loop runs: ``phase1() -> (key_collider u32[], key_min_x f64[], pairs SHARD_PAIR_DTYPE[])``,
    ``phase2(gathered) -> changes CHANGE_DTYPE[]``, ``phase3(gathered)``.  ``step(gather)`` runs them over ``gather(array) -> every rank's
    arrays concatenated`` (a tensor all-gather with a count exchange in the multi-process drivers, plain concatenation in one process)."""

    def __init__(self, lib: F.Library, world: F.World, p: ShardPlan, rank: int, colliders: Dict[str, np.ndarray]):
        self.lib, self.w, self.plan, self.rank = lib, world, p, rank
        self.loc = p.local_bodies(rank)                    # local body -> global body
        ent = np.asarray(colliders["entity_index"], np.uint32)
        owner = p.rank_of_body[np.asarray(colliders["body"])]
        # every collider's min.x key is contributed by ONE rank: its body's owner, static bodies by rank 0
        self.my_colliders = np.flatnonzero((owner == rank) | ((owner < 0) & (rank == 0))).astype(np.uint32)
        loc_ent = ent[np.flatnonzero((owner == rank) | (owner < 0))]      # the sub-world's colliders, in its upload order
        slot = {int(e): i for i, e in enumerate(loc_ent)}
        self.my_local_slots = np.array([slot[int(ent[g])] for g in self.my_colliders], np.int64)
        self.shard = F.Shard(lib, ent, rank)

    def phase1(self):
        w = self.w
        w.run_system("UPDATE_AABB")
        w.run_system("COLLECT_COLLISION_PAIRS")
        mn, _, _ = w.aabbs_download()
        keys = np.asarray(mn[self.my_local_slots, 0], np.float64)
        pr = w.pairs_get()
        new = np.zeros(len(pr), F.SHARD_PAIR_DTYPE)
        new["collider1"], new["collider2"], new["flags"] = pr["collider1"], pr["collider2"], pr["flags"]
        new["body1"], new["body2"] = self.loc[pr["body1"]], self.loc[pr["body2"]]
        new["owner"] = self.rank
        return self.my_colliders, keys, new

    def phase2(self, key_collider, key_min_x, pairs):
        ids, c1, c2, fl = self.shard.phase2(key_collider, key_min_x, pairs)
        if len(ids):
            self.w.contact_pairs_add(ids, c1, c2, fl)
        self.w.active_pairs_set(self.shard.active())
        self.w.run_system("NARROW_PHASE")
        return self.w.contact_changes_get().copy()

    def phase3(self, changes):
        removed_local = self.shard.phase3(changes)
        if len(removed_local):
            self.w.contact_pairs_remove(removed_local)
            self.w.active_pairs_set(self.shard.active())
        off, handles = self.shard.handles()
        self.w.manifold_handles_upload(off, handles)
        self.w.run_system("SOLVER")
        return len(changes)

    @property
    def global_lists(self):
        return self.shard.handles(global_lists=True)

    def step(self, gather):
        kc, kx, pr = self.phase1()
        ch = self.phase2(gather(kc), gather(kx), gather(pr))
        return self.phase3(gather(ch))


def tensor_gather(dist, torch):
    """``gather(array) -> every rank's arrays concatenated in rank order`` over torch.distributed with FLAT tensors: one all_gather of the byte
    counts, one of the payloads padded to the largest (no pickled objects).  Works on gloo (CPU tensors) and on nccl = RCCL (device tensors)."""
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def gather(a):
        a = np.ascontiguousarray(a)
        raw = a.view(np.uint8).reshape(-1)
        n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        if raw.size: buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
        parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = np.concatenate([parts[r][:counts[r]].cpu().numpy() for r in range(world)]) if sum(counts) else np.zeros(0, np.uint8)
        return out.view(a.dtype)

    return gather


# ---- the DEVICE closed loop sharded by islands (avn_dshard_*, round 6): every world replicates the front of the step, simulates its own bodies, and the bodies' components
#      go round once per step.  With avn_comm_init the library issues that all-gather itself inside avn_step; these two helpers are the host-mediated forms. ----
def dshard_step_in_process(worlds):
    """one step of every rank in ONE process (several worlds on one device, or on the CPU oracle): step, then every rank's bodies to every other world"""
    for w in worlds:
        w.step()
    recs = [w.dshard_bodies_pack() for w in worlds]
    for r, w in enumerate(worlds):
        for q, rec in enumerate(recs):
            if q != r:
                w.dshard_bodies_unpack(q, rec)


def dshard_step_distributed(world, gather):
    """one step of this process's rank; `gather` = tensor_gather(dist, torch): one all-gather of the ranks' body records (rank order), split by the owner table"""
    n_ranks, rank, owner = world._dshard
    world.step()
    mine = world.dshard_bodies_pack()
    allrec = gather(mine.reshape(-1)).reshape(-1, 16)
    at = 0
    for q in range(n_ranks):
        n = int((owner == q).sum())
        if q != rank:
            world.dshard_bodies_unpack(q, allrec[at:at + n])
        at += n


def step_in_process_native(loops: List["ShardedClosedLoopNative"]):
    """One step of every rank in ONE process: the gathers are concatenations in rank order."""
    p1 = [l.phase1() for l in loops]
    kc, kx, pr = (np.concatenate([x[i] for x in p1]) for i in range(3))
    ch = np.concatenate([l.phase2(kc, kx, pr) for l in loops])
    return [l.phase3(ch) for l in loops]


def sharded_closed_loop_worlds(lib: F.Library, bits: int, bodies: Dict[str, np.ndarray], colliders: Dict[str, np.ndarray], p: ShardPlan, substeps: int = 4,
                               friction: float = 0.5, native: bool = False):
    """The sub-worlds of every rank of ``p`` in ONE process (tests: two worlds on one device), each with its ShardedClosedLoop."""
    out = []
    for r in range(p.world_size):
        b, loc, g2l = split_bodies(p, r, bodies)
        c = split_colliders(g2l, colliders)
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**b); w.colliders_upload(**c); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction)
        loop = ShardedClosedLoopNative(lib, w, p, r, colliders) if native else ShardedClosedLoop(lib, w, p, r, colliders, np.asarray(bodies["rb_type"]))
        out.append((w, loop, loc))
    return out


def step_in_process(loops: List["ShardedClosedLoop"]):
    """One step of every rank, the exchanges done by handing the payload lists around (no communicator: all ranks live in this process)."""
    p1 = [l.phase1() for l in loops]
    p2 = [l.phase2(p1) for l in loops]
    return [l.phase3(p2) for l in loops]
