"""Worker of tests/test_shard_gloo.py, launched by `python -m torch.distributed.run --nproc-per-node 2 …`:
every rank builds the same global scene, plans the island partition, runs ONLY its own sub-world through the C ABI,
exchanges per-step bounds, and rank 0 merges the bodies into <out>.npz.  backend lib = the CPU oracle (this is a CPU
test of the N > 1 host path; on a GPU box the same worker runs the HIP product when AVN_SHARD_BACKEND=hip)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from avian_amd import _ffi as F, scenes, shard  # noqa: E402
from helpers import hip_lib, oracle_lib  # noqa: E402


def build_case(name):
    """Global scene + joints (deterministic).  Returns (scene, joints or None, velocity tweak)."""
    if name == "stacks":
        sc = scenes.box_stacks(3, 3, 3, 3, gap=5.0)
        rng = np.random.default_rng(5)
        sc.linear_velocity[1:] = rng.normal(scale=0.3, size=(sc.n - 1, 3))   # make the solve non-trivial
        # a few distance joints inside each stack (27 bodies per stack, bodies 1..81)
        jb1, jb2 = [], []
        for s in range(3):
            base = 1 + 27 * s
            for k in range(6):
                jb1.append(base + k); jb2.append(base + 26 - k)
        J = len(jb1)
        joints = dict(body1=np.array(jb1, np.int32), body2=np.array(jb2, np.int32), local_anchor1=np.zeros((J, 3)),
                      local_anchor2=np.zeros((J, 3)), limit_min=np.full(J, 0.5), limit_max=np.full(J, 2.5),
                      compliance=np.full(J, 1e-5))
        return sc, joints
    if name == "approach":   # two stacks, one body of the left stack is thrown at the right stack
        sc = scenes.box_stacks(2, 2, 2, 2, gap=3.0)
        sc.linear_velocity[8] = [40.0, 2.0, 0.0]
        return sc, None
    raise KeyError(name)


def run_world(lib, sc_bodies, sc_colliders, joints, friction, restitution, steps, substeps, dist_mod=None, rb_type=None):
    """The per-rank flow (also used by the single-world reference run): broad phase -> synthetic manifolds for the new
    pairs -> colour -> upload -> step, `steps` times; returns (bodies_download, list of per-step pair arrays, overlaps)."""
    w = F.World(lib, F.default_config(32, substeps=substeps))
    w.bodies_upload(**sc_bodies)
    w.colliders_upload(**sc_colliders)
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    if joints is not None and len(joints["body1"]):
        w.distance_joints_upload(**joints)
    sub = scenes.Scene(position=np.asarray(sc_bodies["position"]), rotation=np.asarray(sc_bodies["rotation"]),
                       linear_velocity=np.asarray(sc_bodies["linear_velocity"]), angular_velocity=np.asarray(sc_bodies["angular_velocity"]),
                       inv_mass=np.asarray(sc_bodies["inv_mass"]), inv_inertia_local=np.asarray(sc_bodies["inv_inertia_local"]),
                       rb_type=np.asarray(sc_bodies["rb_type"]), half_extents=np.asarray(sc_colliders["half_extents"]),
                       shape=np.asarray(sc_colliders["shape"]))
    all_pairs, overlaps = [], []
    known = np.zeros((0, 2), np.int64)
    for s in range(steps):
        w.run_system("UPDATE_AABB")
        _, _, ov = shard.exchange_bounds(w, dist_mod)
        overlaps.append(ov)
        w.run_system("COLLECT_COLLISION_PAIRS")
        p = w.pairs_get()
        all_pairs.append(np.stack([p["collider1"], p["collider2"]], axis=1).astype(np.int64))
        known = np.concatenate([known, np.stack([p["body1"], p["body2"]], axis=1).astype(np.int64)])
        # the (out of path) narrow phase: face manifolds of the CURRENT positions for every known pair, in pair order
        cur = w.bodies_download()
        sub.position = cur["position"].astype(np.float64); sub.linear_velocity = cur["linear_velocity"].astype(np.float64)
        sub.angular_velocity = cur["angular_velocity"].astype(np.float64)
        mf = scenes.axis_aligned_manifolds(sub, known)
        offs, perm = scenes.color_manifolds(lib, mf, sub.rb_type)
        scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, friction, restitution)
        w.run_system("SOLVER")
    out = w.bodies_download()
    w.close()
    return out, all_pairs, overlaps


def run_world_repartition(lib, sc, rank, world, steps, substeps):
    """The level-1 flow WITH the re-partition: like run_world, but when the per-step bounds exchange reports intersecting ranks every rank
    gathers the ranks' states, re-plans (shard.repartition) and rebuilds its sub-world from the bodies, known pairs and merged interval
    order it now owns.  Returns (global ids, bodies_download, per-step owned counts, steps at which the assignment changed)."""
    edges0 = scenes.brute_force_pairs(sc)
    pl = shard.plan(lib, sc.rb_type, sc.position, edges0, world)
    bodies, loc, g2l = shard.split_bodies(pl, rank, sc.body_kwargs())
    cols = shard.split_colliders(g2l, sc.collider_kwargs())
    st = shard.RankState(loc, bodies, cols, np.zeros((0, 2), np.int64), np.asarray(cols["entity_index"]).astype(np.int64), np.full(len(cols["entity_index"]), np.nan))

    def build(st):
        w = F.World(lib, F.default_config(32, substeps=substeps))
        w.bodies_upload(**st.bodies)
        w.colliders_upload(**st.colliders)
        # one collider per body with entity_index = global body index (scenes.Scene): a known body pair IS its collider pair
        w.existing_pairs_upload(np.array([lib.pair_key(int(a), int(b)) for a, b in st.known], np.uint64))
        return w

    def current(st, w):
        cur = w.bodies_download()
        b = dict(st.bodies)
        for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
            b[k] = cur[k].astype(np.float64)
        return b
    w = build(st)
    owned_hist, changed = [], []
    for s in range(steps):
        w.run_system("UPDATE_AABB")
        _, _, ov = shard.exchange_bounds(w, dist if world > 1 else None)
        if ov:
            st.bodies = current(st, w)
            mn, mx, _ = w.aabbs_download()
            gathered = [None] * world
            dist.all_gather_object(gathered, (st, (mn.astype(np.float64), mx.astype(np.float64))))
            pl2, new_states, cross = shard.repartition(lib, [g[0] for g in gathered], [g[1] for g in gathered], sc.n, sc.rb_type, world)
            if any(not np.array_equal(g[0].gid, n.gid) for g, n in zip(gathered, new_states)):
                w.close()
                st = new_states[rank]
                w = build(st)
                w.run_system("UPDATE_AABB")
                changed.append(s)
        w.run_system("COLLECT_COLLISION_PAIRS")
        p = w.pairs_get()
        st.known = np.concatenate([st.known, np.stack([st.gid[p["body1"]], st.gid[p["body2"]]], axis=1).astype(np.int64)])
        mn, _, ents = w.aabbs_download()
        e2slot = {int(e): i for i, e in enumerate(np.asarray(st.colliders["entity_index"]))}
        st.order = ents.astype(np.int64)
        st.order_key = np.array([mn[e2slot[int(e)], 0] for e in ents], np.float64)
        g2l = np.full(sc.n, -1, np.int64); g2l[st.gid] = np.arange(len(st.gid))
        cur = w.bodies_download()
        sub = sc.subset(st.gid)
        sub.position = cur["position"].astype(np.float64); sub.linear_velocity = cur["linear_velocity"].astype(np.float64)
        sub.angular_velocity = cur["angular_velocity"].astype(np.float64)
        mf = scenes.axis_aligned_manifolds(sub, g2l[st.known])
        offs, perm = scenes.color_manifolds(lib, mf, sub.rb_type)
        scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, sc.restitution)
        w.run_system("SOLVER")
        owned_hist.append(int((sc.rb_type[st.gid] != F.RB_STATIC).sum()))
    out = w.bodies_download()
    w.close()
    return st.gid, out, owned_hist, changed


def slab_scene():
    sc = scenes.sparse_mixed(6000, side=26.0)
    sc.linear_velocity *= 6.0   # colliders cross slab boundaries within a few frames
    return sc


def moved(sc, step, dt=1.0 / 60.0):
    """Bodies of frame `step` (free flight, host-integrated in f64: the broad phase is the system under test)."""
    b = sc.body_kwargs()
    b["position"] = sc.position + sc.linear_velocity * (dt * step)
    return b


def run_slabs(lib, rank, world, steps, out_path, device=None):
    """Slab-sharded broad phase over `steps` frames: every rank sweeps its x-slab + halo, all-gathers the pair records."""
    sc = slab_scene()
    cols = sc.collider_kwargs()
    full = F.World(lib, F.default_config(32, substeps=1))   # replicated AABB update (no sweep): the slab planner's input
    full.bodies_upload(**sc.body_kwargs()); full.colliders_upload(**cols)
    known = np.zeros(0, np.uint64)
    per_step = []
    order = None   # the single world's persistent interval order, replicated on every rank
    for s in range(steps):
        b = moved(sc, s)
        full.bodies_upload(**b); full.run_system("UPDATE_AABB")
        mn, mx, _ = full.aabbs_download()
        rec = shard.slab_broad_phase_step(lib, 32, b, cols, mn[:, 0], mx[:, 0], known, rank, world, dist if world > 1 else None, device, prev_order=order)
        order = shard.slab_next_order(order, mn[:, 0], len(mn))
        known = np.concatenate([known, shard.pair_keys(rec)])
        per_step.append(rec)
    if rank == 0:
        np.savez(out_path, **{f"pairs_s{s}": r for s, r in enumerate(per_step)})


def run_level2(lib, rank, world, steps, out_path, keep=None, joints=False):
    """Level-2 sharding over real ranks: this rank builds ONLY its slab world, steps it with shard.level2_solver and moves the boundary
    records with point-to-point sends (gloo here; the library's own RCCL transport replaces this loop on a multi-GPU node)."""
    from level2_helpers import global_problem, overflow_from
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=7)
    if keep is not None:   # colours >= keep in the overflow colour: its levels travel between the ranks as extra exchange slots
        offs = overflow_from(offs, keep)
    jkw = None
    if joints:   # joints on bodies shared between the two ranks: the joint slot travels after every substep's joint systems
        from level2_helpers import restrict_joints, stack_joints
        jkw = stack_joints(sc, 8, 4, 5, seed=2, damped=True)
    plan = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world, joints=(jkw["body1"], jkw["body2"], jkw["joint_type"], True) if joints else None)
    me = plan[rank]
    w = F.World(lib, F.default_config(32, substeps=3))
    w.bodies_upload(**{k: (np.asarray(v)[me.bodies] if v is not None else None) for k, v in sc.body_kwargs().items()})
    scenes.upload_manifolds(w, shard.level2_local_manifolds(me, pm), me.color_offsets, sc.friction, 0.3)
    if joints and len(me.joints):
        w.joints_upload(**restrict_joints(jkw, me, me.joints))
    me.upload(w)

    def exchange(color, out, need):
        n_p = len(me.peers)
        reqs, bufs = [], {}
        for p in need:
            cnt = int(me.recv_offsets[color * n_p + p + 1] - me.recv_offsets[color * n_p + p])
            bufs[p] = torch.empty((cnt, 16 if me.joint_slot and color == me.joint_slot_index else 8), dtype=torch.float32)
            reqs.append(dist.irecv(bufs[p], src=int(me.peers[p]), tag=color))
        for p, rec in out.items():
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(rec, np.float32)), dst=int(me.peers[p]), tag=color))
        for r in reqs:
            r.wait()
        return {p: bufs[p].numpy() for p in need}
    for _ in range(steps):
        shard.level2_solver(w, me, 3, exchange, restitution=True, has_joints=joints)
    b = w.bodies_download(); imp = w.impulses_download()
    np.savez(out_path + f".rank{rank}.npz", bodies=me.bodies, manifolds=me.manifolds, **{"b_" + k: v for k, v in b.items()}, **{"i_" + k: v for k, v in imp.items()})


def run_level2_problem(lib, rank, world, steps, out_path):
    """Level 2 on a saved closed-loop manifold set (level2_helpers.save_problem): f64, overflow levels as exchange slots."""
    from level2_helpers import load_problem, make_world_from
    body, mf, offs, warm = load_problem(out_path + ".problem.npz")
    plan = shard.level2_plan_lib(lib, body["position"], body["rb_type"], mf["body1"], mf["body2"], offs, world)
    me = plan[rank]
    w = make_world_from(lib, 64, body, mf, offs, warm, 2, rank=me)

    def exchange(slot, out, need):
        n_p = len(me.peers)
        reqs, bufs = [], {}
        for p in need:
            cnt = int(me.recv_offsets[slot * n_p + p + 1] - me.recv_offsets[slot * n_p + p])
            bufs[p] = torch.empty((cnt, 8), dtype=torch.float64)
            reqs.append(dist.irecv(bufs[p], src=int(me.peers[p]), tag=slot))
        for p, rec in out.items():
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(rec, np.float64)), dst=int(me.peers[p]), tag=slot))
        for r in reqs:
            r.wait()
        return {p: bufs[p].numpy() for p in need}
    for _ in range(steps):
        shard.level2_solver(w, me, 2, exchange, restitution=False)
    b = w.bodies_download(); imp = w.impulses_download()
    np.savez(out_path + f".rank{rank}.npz", bodies=me.bodies, manifolds=me.manifolds, n_levels=me.n_overflow_levels, **{"b_" + k: v for k, v in b.items()}, **{"i_" + k: v for k, v in imp.items()})


def main():
    case, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    backend = os.environ.get("AVN_SHARD_BACKEND", "oracle")
    dist.init_process_group(backend="gloo" if backend == "oracle" else "nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = oracle_lib() if backend == "oracle" else hip_lib()
    if case == "level2_problem":
        run_level2_problem(lib, rank, world, steps, out_path)
        dist.barrier()
        dist.destroy_process_group()
        return
    if case in ("level2", "level2_overflow", "level2_joints"):
        run_level2(lib, rank, world, steps, out_path, keep=4 if case == "level2_overflow" else None, joints=case == "level2_joints")
        dist.barrier()
        dist.destroy_process_group()
        return
    if case == "merge":   # the thrown body merges the two ranks' islands: re-partition, then continue as the single world would
        sc, _ = build_case("approach")
        gid, got, owned_hist, changed = run_world_repartition(lib, sc, rank, world, steps, 2)
        gathered = [None] * world
        dist.all_gather_object(gathered, (gid, got, owned_hist, changed))
        if rank == 0:
            merged = {k: np.zeros((sc.n,) + v.shape[1:], v.dtype) for k, v in got.items()}
            holders = np.zeros(sc.n, np.int64)
            for g, d, _, _ in gathered:
                dyn = sc.rb_type[g] != F.RB_STATIC
                holders[g[dyn]] += 1
                for k in merged:
                    merged[k][g] = d[k]          # (static bodies: identical everywhere)
            np.savez(out_path, holders=holders, owned_hist=np.array([g[2] for g in gathered]), changed=np.array(gathered[0][3], np.int64), **merged)
        dist.barrier()
        dist.destroy_process_group()
        return
    if case == "slabs":
        run_slabs(lib, rank, world, steps, out_path, None if backend == "oracle" else f"cuda:{os.environ.get('LOCAL_RANK', '0')}")
        dist.barrier()
        dist.destroy_process_group()
        return
    sc, joints = build_case(case)
    edges = [scenes.brute_force_pairs(sc)]
    if joints is not None:
        edges.append(np.stack([joints["body1"], joints["body2"]], axis=1))
    pl = shard.plan(lib, sc.rb_type, sc.position, np.concatenate(edges), world)
    bodies, loc, g2l = shard.split_bodies(pl, rank, sc.body_kwargs())
    cols = shard.split_colliders(g2l, sc.collider_kwargs())
    jl = None
    if joints is not None:
        jl, _, _ = shard.split_pairwise(g2l, pl, rank, joints)
    got, pairs, overlaps = run_world(lib, bodies, cols, jl, sc.friction, sc.restitution, steps, 2, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (loc, got, pairs, overlaps, int((pl.rank_of_body == rank).sum())))
    if rank == 0:
        template = {"position": sc.position.astype(np.float32), "rotation": sc.rotation.astype(np.float32),
                    "linear_velocity": sc.linear_velocity.astype(np.float32), "angular_velocity": sc.angular_velocity.astype(np.float32)}
        merged = shard.merge_bodies(pl, sc.n, [(g[0], g[1]) for g in gathered], template)
        npairs = np.array([[len(ps) for ps in g[2]] for g in gathered])
        pair_sets = {f"pairs_r{r}_s{s}": ps for r, g in enumerate(gathered) for s, ps in enumerate(g[2])}
        first_overlap = next((s for s in range(steps) if any(len(g[3][s]) for g in gathered)), -1)
        np.savez(out_path, rank_of_body=pl.rank_of_body, island_of_body=pl.island_of_body, n_islands=pl.n_islands,
                 owned=np.array([g[4] for g in gathered]), npairs=npairs, first_overlap=first_overlap, **merged, **pair_sets)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
