"""Shared by the CPU and GPU level-2 sharding tests: build a box stack with its global manifold set and colouring, split it into x-slab
worlds (avian_amd.shard.level2_plan) and step the split worlds with an in-process halo exchange."""
from __future__ import annotations

import numpy as np

from avian_amd import scenes, shard
from helpers import F


def global_problem(lib, nx, ny, nz, seed=0, restitution=0.0):
    sc = scenes.box_stack(nx, ny, nz)
    rng = np.random.default_rng(seed)
    sc.linear_velocity[1:] += rng.normal(scale=0.3, size=(sc.n - 1, 3))   # not a resting lattice: impulses differ everywhere
    sc.angular_velocity[1:] += rng.normal(scale=0.2, size=(sc.n - 1, 3))
    pairs = scenes.brute_force_pairs(sc)
    mf = scenes.axis_aligned_manifolds(sc, pairs)
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    return sc, scenes.permute_manifolds(mf, perm), offs, restitution


def overflow_from(offs, keep):
    """The same colour-major manifold set with colours >= keep moved into the overflow colour (list order = their colour-major order): what a deep pile does to
    the reference's 24-colour graph (constraint_graph.rs:163-196) -- and most of those manifolds then sit on bodies shared between slabs."""
    o = np.asarray(offs).copy()
    o[keep:24] = o[keep]
    return o


def make_single(lib, bits, sc, pm, offs, restitution, substeps):
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    w.bodies_upload(**sc.body_kwargs())
    scenes.upload_manifolds(w, pm, offs, sc.friction, restitution)
    return w


def make_split(lib, bits, sc, pm, offs, restitution, substeps, world_size):
    plan = shard.level2_plan_lib(lib, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world_size)
    worlds = []
    for r in plan:
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**{k: (np.asarray(v)[r.bodies] if v is not None else None) for k, v in sc.body_kwargs().items()})
        scenes.upload_manifolds(w, shard.level2_local_manifolds(r, pm), r.color_offsets, sc.friction, restitution)
        r.upload(w)
        worlds.append(w)
    return plan, worlds


def step_split_in_process(plan, worlds, substeps, restitution):
    """All ranks in one process: every colour of every pass runs on all worlds, then the records travel through Python.  The per-colour
    lock step is emulated by running the generator-free helper once per rank with a mailbox."""
    R = len(worlds)
    # because level2_solver drives ONE world, interleave by colour here instead: replicate its schedule over all worlds
    def all_run(system):
        for w in worlds:
            w.run_system(system)

    def contact_pass(system):
        for c in plan[0].solve_order():
            for w in worlds:
                w.run_color_pass(system, c)
            box = {}
            for r, (w, pl) in enumerate(zip(worlds, plan)):
                n_p = len(pl.peers)
                for p in range(n_p):
                    if pl.send_offsets[c * n_p + p + 1] > pl.send_offsets[c * n_p + p]:
                        box[(r, int(pl.peers[p]))] = w.halo_pack(c, p)
            for r, (w, pl) in enumerate(zip(worlds, plan)):
                n_p = len(pl.peers)
                for p in range(n_p):
                    if pl.recv_offsets[c * n_p + p + 1] > pl.recv_offsets[c * n_p + p]:
                        w.halo_unpack(c, p, box[(int(pl.peers[p]), r)])
    for s in ("PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"):
        all_run(s)
    for _ in range(substeps):
        all_run("INTEGRATE_VELOCITIES")
        contact_pass("WARM_START"); contact_pass("SOLVE_CONTACTS_BIAS")
        all_run("INTEGRATE_POSITIONS")
        contact_pass("SOLVE_CONTACTS_RELAX")
        for s in ("XPBD_SOLVE", "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING"):
            all_run(s)
    all_run("CLEAR_VELOCITY_INCREMENTS")
    if restitution:
        contact_pass("SOLVE_RESTITUTION")
    all_run("WRITEBACK_SOLVER_BODIES"); all_run("STORE_CONTACT_IMPULSES")


def compare_with_single(single, plan, worlds):
    ref = single.bodies_download()
    imp = single.impulses_download()
    for pl, w in zip(plan, worlds):
        got = w.bodies_download()
        for k in ref:
            assert np.array_equal(ref[k][pl.bodies], got[k]), f"bodies.{k} of a slab differ from the single world"
        gi = w.impulses_download()
        for k in imp:
            assert np.array_equal(imp[k][pl.manifolds], gi[k]), f"impulses.{k} of a slab differ from the single world"


def closed_loop_problem(lib, bits, dims, steps, substeps=4, friction=0.5):
    """avian_amd.level2_bench.closed_loop_island (the bench's `--gpus N` leg uses the same set): the device closed loop's own manifolds of a collapsing box stack in
    the host-uploaded form level 2 works on.  Returns (scene with the stepped body state, manifolds, offsets, warm-start impulses)."""
    from avian_amd import level2_bench
    return level2_bench.closed_loop_island(lib, F, scenes, bits, dims, steps, substeps=substeps, friction=friction)


def save_problem(path, sc, mf, offs, warm):
    np.savez(path, offs=offs, wn=warm[0], wt=warm[1], **{"b_" + k: v for k, v in sc.body_kwargs().items() if v is not None}, **{"m_" + k: v for k, v in mf.items()})


def load_problem(path):
    d = np.load(path)
    body = {k[2:]: d[k] for k in d.files if k.startswith("b_")}
    mf = {k[2:]: d[k] for k in d.files if k.startswith("m_")}
    return body, mf, d["offs"], (d["wn"], d["wt"])


def make_world_from(lib, bits, body, mf, offs, warm, substeps, rank=None):
    """A host-manifold world of a closed_loop_problem: the whole set, or one level-2 rank's share of it (plan uploaded)."""
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    if rank is None:
        w.bodies_upload(**body)
        scenes.upload_manifolds(w, mf, offs, mf["friction"], mf["restitution"], warm[0], warm[1])
        return w
    w.bodies_upload(**{k: np.asarray(v)[rank.bodies] for k, v in body.items()})
    lm = shard.level2_local_manifolds(rank, mf)
    scenes.upload_manifolds(w, lm, rank.color_offsets, lm["friction"], lm["restitution"], warm[0][rank.manifolds], warm[1][rank.manifolds])
    rank.upload(w)
    return w


def cfg5_shaped_case(lib_gen, libs_single, lib_split, world_sizes, steps=2, substeps=2, dims=(50, 20, 50), problem=None):
    """VERDICT r5 item 3's case: >= 50 000 bodies in f64 with >= 10 000 overflow-colour manifolds (the closed loop's own colouring of a collapsing stack), over
    2 / 4 slabs; every slab's bodies and impulses == the unsplit world's on each of `libs_single`, every step."""
    sc, mf, offs, warm = problem if problem is not None else closed_loop_problem(lib_gen, 64, dims, 4, substeps=8)
    body = {k: v for k, v in sc.body_kwargs().items() if v is not None}
    assert sc.n > 50_000 and offs[24] - offs[23] >= 10_000
    for R in world_sizes:
        plan = shard.level2_plan_lib(lib_split, sc.position, sc.rb_type, mf["body1"], mf["body2"], offs, R)
        L = plan[0].n_overflow_levels
        n_p = [len(p.peers) for p in plan]
        assert L > 1 and sum(int(p.send_offsets[-1] - p.send_offsets[23 * n]) for p, n in zip(plan, n_p) if n) > 1000, "the overflow levels must carry shared bodies"
        singles = [make_world_from(l, 64, body, mf, offs, warm, substeps) for l in libs_single]
        worlds = [make_world_from(lib_split, 64, body, mf, offs, warm, substeps, rank=r) for r in plan]
        for _ in range(steps):
            for s in singles:
                s.run_system("SOLVER")
            step_split_in_process(plan, worlds, substeps, False)
            for s in singles:
                compare_with_single(s, plan, worlds)
        assert float(np.abs(singles[0].bodies_download()["linear_velocity"]).max()) > 0.05
        for w in singles + worlds:
            w.close()
    return sc, mf, offs, warm


# ---- joints across slabs (round 6: avn_halo_joint_slot_set) -------------------------------------------------------------------------------------------------
def stack_joints(sc, nx, ny, nz, seed=0, damped=True, types=(F.JOINT_DISTANCE, F.JOINT_SPHERICAL, F.JOINT_FIXED)):
    """Joints inside global_problem's box stack (bodies 1 + (j * nz + k) * nx + i): chains along x in several rows -- they cross every slab cut, so their bodies are shared
    between worlds --, a few joints between vertical neighbours, and joints to the static ground (body 0): with JointDamping those run through the type's DUMMY pair
    and are one serial chain in the reference (solver/plugin.rs:766-767).  Returns the avn_joints_upload kwargs (global body indices)."""
    rng = np.random.default_rng(seed)
    idx = lambda i, j, k: 1 + (j * nz + k) * nx + i
    b1, b2, ty = [], [], []
    for (j, k) in [(0, 0), (1, 2), (ny - 1, nz - 1), (2 % ny, 1)]:
        t = types[(j + k) % len(types)]
        for i in range(nx - 1):
            if (i + j + k) % 4 == 3:
                continue   # (gaps: several components per row)
            b1.append(idx(i, j, k)); b2.append(idx(i + 1, j, k)); ty.append(t)
    for i in (1, nx // 2, nx - 2):
        b1.append(idx(i, 0, 1)); b2.append(idx(i, 1, 1)); ty.append(F.JOINT_DISTANCE)
    for i in (0, nx // 2, nx - 1):   # to the static ground
        b1.append(0); b2.append(idx(i, 0, nz - 1)); ty.append(F.JOINT_DISTANCE)
        b1.append(idx(i, 0, 2 % nz)); b2.append(0); ty.append(F.JOINT_SPHERICAL)
    order = np.argsort(np.asarray(ty), kind="stable")   # (the library solves type by type in array order: keep the global array type-major so that a restriction keeps it)
    b1, b2, ty = np.asarray(b1, np.int32)[order], np.asarray(b2, np.int32)[order], np.asarray(ty, np.uint8)[order]
    J = len(b1)
    d = sc.position[b2] - sc.position[b1]
    a1 = 0.5 * d; a2 = -0.5 * d
    ground = (b1 == 0) | (b2 == 0)
    a1[ground] = 0.0; a2[ground] = 0.0
    a1[b1 == 0] = (sc.position[b2] - sc.position[b1])[b1 == 0]; a2[b2 == 0] = (sc.position[b1] - sc.position[b2])[b2 == 0]
    kw = dict(joint_type=ty, body1=b1, body2=b2, local_anchor1=a1, local_anchor2=a2, compliance=np.tile([1e-4, 1e-3, 1e-3], (J, 1)) * rng.uniform(0.5, 2.0, (J, 1)),
              axis=np.tile([1.0, 0, 0], (J, 1)), limit_min=np.where(ty == F.JOINT_DISTANCE, 0.0, -0.4), limit_max=np.where(ty == F.JOINT_DISTANCE, 0.3, 0.4),
              limit_flags=np.where(ty == F.JOINT_SPHERICAL, F.JOINT_HAS_LIMIT1, 0).astype(np.uint8))
    if damped:
        kw["damping_linear"] = rng.uniform(0.5, 3.0, J); kw["damping_angular"] = rng.uniform(0.5, 3.0, J)
    return kw


def restrict_joints(kw, rank, joint_ids):
    """A rank's joints: the rows `joint_ids` of the global arrays, bodies re-indexed to the rank's local numbering."""
    g2l = np.full(int(np.max(rank.bodies)) + 1, -1, np.int64); g2l[np.asarray(rank.bodies, np.int64)] = np.arange(len(rank.bodies))
    out = {k: (np.asarray(v)[joint_ids] if v is not None else None) for k, v in kw.items()}
    out["body1"] = g2l[out["body1"]].astype(np.int32); out["body2"] = g2l[out["body2"]].astype(np.int32)
    assert (out["body1"] >= 0).all() and (out["body2"] >= 0).all(), "a rank's joint names a body the rank does not hold"
    return out


def make_joint_worlds(lib, bits, sc, pm, offs, restitution, substeps, world_size, jkw, planner="lib"):
    """(unsplit world with all joints, plan, split worlds with their own joints and the joint slot)."""
    single = make_single(lib, bits, sc, pm, offs, restitution, substeps)
    single.joints_upload(**jkw)
    jp = (jkw["body1"], jkw["body2"], jkw["joint_type"], "damping_linear" in jkw)
    if planner == "lib":
        plan = shard.level2_plan_lib(lib, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world_size, joints=jp)
    else:
        plan = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world_size, joints=jp)
    worlds = []
    for r in plan:
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**{k: (np.asarray(v)[r.bodies] if v is not None else None) for k, v in sc.body_kwargs().items()})
        scenes.upload_manifolds(w, shard.level2_local_manifolds(r, pm), r.color_offsets, sc.friction, restitution)
        if len(r.joints):
            w.joints_upload(**restrict_joints(jkw, r, r.joints))
        r.upload(w)
        worlds.append(w)
    return single, plan, worlds


def step_split_with_joints(plan, worlds, substeps, restitution):
    """step_split_in_process + the joint slot after the joint systems of every substep."""
    def all_run(system):
        for w in worlds:
            w.run_system(system)

    def exchange_slot(c):
        box = {}
        for r, (w, pl) in enumerate(zip(worlds, plan)):
            n_p = len(pl.peers)
            for p in range(n_p):
                if pl.send_offsets[c * n_p + p + 1] > pl.send_offsets[c * n_p + p]:
                    box[(r, int(pl.peers[p]))] = w.halo_pack(c, p)
        for r, (w, pl) in enumerate(zip(worlds, plan)):
            n_p = len(pl.peers)
            for p in range(n_p):
                if pl.recv_offsets[c * n_p + p + 1] > pl.recv_offsets[c * n_p + p]:
                    w.halo_unpack(c, p, box[(int(pl.peers[p]), r)])

    def contact_pass(system):
        for c in plan[0].solve_order():
            for w in worlds:
                w.run_color_pass(system, c)
            exchange_slot(c)
    for s in ("PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"):
        all_run(s)
    for _ in range(substeps):
        all_run("INTEGRATE_VELOCITIES")
        contact_pass("WARM_START"); contact_pass("SOLVE_CONTACTS_BIAS")
        all_run("INTEGRATE_POSITIONS")
        contact_pass("SOLVE_CONTACTS_RELAX")
        for s in ("XPBD_SOLVE", "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING"):
            all_run(s)
        if plan[0].joint_slot:
            exchange_slot(plan[0].joint_slot_index)
    all_run("CLEAR_VELOCITY_INCREMENTS")
    if restitution:
        contact_pass("SOLVE_RESTITUTION")
    all_run("WRITEBACK_SOLVER_BODIES"); all_run("STORE_CONTACT_IMPULSES")


def compare_joints_with_single(single, plan, worlds):
    ref = single.joints_download()
    for pl, w in zip(plan, worlds):
        if not len(pl.joints):
            continue
        got = w.joints_download()
        for k in ref:
            assert np.array_equal(ref[k][pl.joints], got[k]), f"joints.{k} of a slab differ from the single world"
