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
        w.halo_plan_upload(r.peers, r.send_offsets, r.send_bodies, r.recv_offsets, r.recv_bodies)
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
        for c in [F.COLOR_OVERFLOW_INDEX] + list(range(F.COLOR_OVERFLOW_INDEX)):
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
