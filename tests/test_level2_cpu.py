"""CPU (oracle backend): level-2 sharding -- ONE contact island split into x-slab worlds with a global colouring and a halo exchange of
boundary-body velocities after every colour (include/avian_mi355x.h avn_halo_plan, avian_amd/shard.py level2_*) -- is bit-identical to the
unsplit world: bodies of every slab and impulses of every owned manifold, after several steps, f32 and f64, 2 / 3 / 4 slabs."""
import numpy as np
import pytest

from avian_amd import shard
from helpers import F, oracle_lib
from level2_helpers import compare_with_single, global_problem, make_single, make_split, overflow_from, step_split_in_process


@pytest.mark.parametrize("bits,world_size,restitution", [(32, 2, 0.0), (32, 3, 0.3), (64, 2, 0.3), (32, 4, 0.0)])
def test_split_island_equals_single_world(bits, world_size, restitution):
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=bits + world_size)
    single = make_single(lib, bits, sc, pm, offs, restitution, 3)
    plan, worlds = make_split(lib, bits, sc, pm, offs, restitution, 3, world_size)
    assert sum(len(p.manifolds) for p in plan) == len(pm["body1"]) and all(len(p.manifolds) for p in plan)
    assert sum(len(p.send_bodies) for p in plan) == sum(len(p.recv_bodies) for p in plan) > 0, "the slabs must actually share bodies"
    for step in range(3):
        single.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, restitution > 0)
        compare_with_single(single, plan, worlds)
    v = single.bodies_download()["linear_velocity"]
    assert float(np.abs(v).max()) > 0.05


@pytest.mark.parametrize("bits,world_size,keep,restitution", [(32, 2, 6, 0.0), (32, 3, 0, 0.3), (64, 2, 3, 0.3), (32, 4, 10, 0.0)])
def test_overflow_colour_on_shared_bodies_equals_single_world(bits, world_size, keep, restitution):
    """Round 6: overflow-colour manifolds on bodies shared between slabs.  The planner cuts the GLOBAL overflow list into levels, every level is an exchange slot;
    the split worlds reproduce the single world's serial walk (solver/plugin.rs:461-467) bit for bit."""
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=bits + world_size + keep)
    offs = overflow_from(offs, keep)
    single = make_single(lib, bits, sc, pm, offs, restitution, 3)
    plan, worlds = make_split(lib, bits, sc, pm, offs, restitution, 3, world_size)
    L = plan[0].n_overflow_levels
    assert L > 1 and offs[24] - offs[23] > 100
    n_p = [len(p.peers) for p in plan]
    assert sum(int(p.send_offsets[-1] - p.send_offsets[23 * n]) for p, n in zip(plan, n_p) if n) > 0, "overflow levels must actually exchange bodies"
    for step in range(3):
        single.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, restitution > 0)
        compare_with_single(single, plan, worlds)
    assert float(np.abs(single.bodies_download()["linear_velocity"]).max()) > 0.05


@pytest.fixture(scope="module")
def cfg5_problem():
    from level2_helpers import closed_loop_problem
    return closed_loop_problem(oracle_lib(), 64, (50, 20, 50), 4, substeps=8)


def test_cfg5_shaped_closed_loop_manifolds_over_2_and_4_slabs(cfg5_problem):
    """50 000 cuboids in f64, the closed loop's own manifolds and colouring after 4 steps of the collapsing lattice (about 5 * 10^5 manifolds, 10^5 of them in the
    overflow colour, about 400 levels): level 2 over 2 and 4 slabs == the single world, bit for bit."""
    from level2_helpers import cfg5_shaped_case
    lib = oracle_lib()
    cfg5_shaped_case(lib, [lib], lib, (2, 4), problem=cfg5_problem)


def test_cfg5_shaped_over_gloo_world_size_2(cfg5_problem, tmp_path):
    """The same case over two real processes (torch.distributed, gloo): each rank loads the problem, plans with the library's planner, builds ONLY its slab, steps it
    with shard.level2_solver and exchanges boundary records after every colour and every overflow level (~ 430 slots per pass)."""
    from level2_helpers import make_world_from, save_problem
    from test_shard_gloo import launch
    sc, mf, offs, warm = cfg5_problem
    out = str(tmp_path / "level2_cfg5")
    save_problem(out + ".problem.npz", sc, mf, offs, warm)
    launch("level2_problem", out, 2)
    lib = oracle_lib()
    single = make_world_from(lib, 64, {k: v for k, v in sc.body_kwargs().items() if v is not None}, mf, offs, warm, 2)
    for _ in range(2):
        single.run_system("SOLVER")
    ref, imp = single.bodies_download(), single.impulses_download()
    seen = np.zeros(len(mf["body1"]), bool)
    for r in range(2):
        d = np.load(out + f".rank{r}.npz")
        assert int(d["n_levels"]) > 1
        for k in ref:
            assert np.array_equal(ref[k][d["bodies"]], d["b_" + k]), f"rank {r}: bodies.{k}"
        for k in imp:
            assert np.array_equal(imp[k][d["manifolds"]], d["i_" + k]), f"rank {r}: impulses.{k}"
        seen[d["manifolds"]] = True
    assert seen.all()


def test_plan_properties():
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 9, 3, 4)
    plan = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, 3)
    owned = np.concatenate([p.manifolds for p in plan])
    assert np.array_equal(np.sort(owned), np.arange(len(pm["body1"]))), "every manifold has exactly one owner"
    for r, p in enumerate(plan):
        assert np.all(np.diff(p.bodies) > 0) and 0 in p.bodies, "ascending bodies, the static ground everywhere"
        n_p = len(p.peers)
        for c in range(p.n_slots):
            for k, q in enumerate(p.peers):
                other = plan[int(q)]
                k2 = int(np.flatnonzero(other.peers == r)[0])
                a = p.bodies[p.send_bodies[p.send_offsets[c * n_p + k]:p.send_offsets[c * n_p + k + 1]]]
                b = other.bodies[other.recv_bodies[other.recv_offsets[c * len(other.peers) + k2]:other.recv_offsets[c * len(other.peers) + k2 + 1]]]
                assert np.array_equal(a, b), "send list of one side == receive list of the other, same order (global body ids)"
    assert all(p.n_overflow_levels == 1 for p in plan), "no overflow manifold on a shared body: the plain 24 slots"
    # everything in the overflow colour: round 5 refused it, round 6 cuts the global list into levels -- manifolds of a level share no non-static body, and a
    # manifold's level is above the level of every earlier manifold on one of its bodies
    offs_all = offs.copy(); offs_all[:] = 0; offs_all[24] = len(pm["body1"])   # everything in colour 23
    plan = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs_all, 2)
    L = plan[0].n_overflow_levels
    assert L > 1 and all(p.n_overflow_levels == L for p in plan)
    level = np.full(len(pm["body1"]), -1, np.int64)
    for p in plan:
        level[p.manifolds] = p.overflow_level
    assert level.min() == 0 and level.max() == L - 1
    static = sc.rb_type == F.RB_STATIC
    last = {}
    for m in range(len(level)):
        for b in (int(pm["body1"][m]), int(pm["body2"][m])):
            if static[b]:
                continue
            assert last.get(b, -1) < level[m], "levels rise along every body's chain"
            last[b] = level[m]


def test_library_planner_equals_the_numpy_planner():
    """avn_level2_plan_* of the product (host C++), of the oracle (a differently organised implementation) and shard.level2_plan (numpy)
    give the same plan: bodies, manifolds, colour offsets, peers and every send / receive list."""
    from helpers import hip_lib
    lib = oracle_lib()
    for dims, R in (((9, 3, 4), 3), ((8, 4, 5), 2), ((6, 2, 2), 4), ((5, 2, 2), 1)):
        sc, pm, offs, _ = global_problem(lib, *dims)
        want = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R)
        for L in (hip_lib(), lib):
            got = shard.level2_plan_lib(L, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R)
            assert len(got) == len(want) == R
            for a, b in zip(got, want):
                for f in ("bodies", "manifolds", "color_offsets", "peers", "send_offsets", "send_bodies", "recv_offsets", "recv_bodies", "n_overflow_levels", "overflow_level"):
                    assert np.array_equal(np.asarray(getattr(a, f)).astype(np.int64), np.asarray(getattr(b, f)).astype(np.int64)), (dims, R, L.prefix, f)
    # the overflow colour on shared bodies (levels): the first 8 colours kept, everything behind them moved into the overflow colour; and everything in it
    sc, pm, offs, _ = global_problem(lib, 9, 3, 4)
    for keep in (8, 0):
        offs_o = offs.copy(); offs_o[keep:24] = offs[keep]
        for R in (2, 3):
            want = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs_o, R)
            assert want[0].n_overflow_levels > 1
            for L in (hip_lib(), lib):
                got = shard.level2_plan_lib(L, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs_o, R)
                for a, b in zip(got, want):
                    for f in ("bodies", "manifolds", "color_offsets", "peers", "send_offsets", "send_bodies", "recv_offsets", "recv_bodies", "n_overflow_levels", "overflow_level"):
                        assert np.array_equal(np.asarray(getattr(a, f)).astype(np.int64), np.asarray(getattr(b, f)).astype(np.int64)), (keep, R, L.prefix, f)


@pytest.mark.parametrize("case", ["level2", "level2_overflow", "level2_joints"])
def test_level2_over_gloo_world_size_2(tmp_path, case):
    """Two real processes (torch.distributed, gloo): each builds only its slab, steps it with shard.level2_solver and exchanges the boundary
    records point to point after every colour (level2_overflow: and after every LEVEL of the overflow colour, which then holds the manifolds of
    colours 4..22); merged result == the single world, bit for bit."""
    from test_shard_gloo import launch
    out = str(tmp_path / "level2")
    launch(case, out, 3)
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=7)
    if case == "level2_overflow":
        offs = overflow_from(offs, 4)
    jkw = None
    if case == "level2_joints":
        from level2_helpers import stack_joints
        jkw = stack_joints(sc, 8, 4, 5, seed=2, damped=True)
    single = make_single(lib, 32, sc, pm, offs, 0.3, 3)
    if jkw is not None:
        single.joints_upload(**jkw)
    for _ in range(3):
        single.run_system("SOLVER")
    ref, imp = single.bodies_download(), single.impulses_download()
    seen = np.zeros(len(pm["body1"]), bool)
    for r in range(2):
        d = np.load(out + f".rank{r}.npz")
        for k in ref:
            assert np.array_equal(ref[k][d["bodies"]], d["b_" + k]), f"rank {r}: bodies.{k}"
        for k in imp:
            assert np.array_equal(imp[k][d["manifolds"]], d["i_" + k]), f"rank {r}: impulses.{k}"
        seen[d["manifolds"]] = True
    assert seen.all()


@pytest.mark.parametrize("bits,world_size,damped,keep", [(32, 2, True, None), (32, 3, False, None), (64, 2, True, 6), (32, 4, True, None)])
def test_joints_on_shared_bodies_equal_the_single_world(bits, world_size, damped, keep):
    """Round 6: joints whose bodies are shared between slabs.  A joint component belongs to one world (its owner holds all its bodies); after the joint systems of
    every substep the owner's SolverBody records of the component's shared bodies travel to the other holders (the joint slot).  Chains across every cut, joints to the
    static ground (with JointDamping: one serial chain through the DUMMY pair), three joint types; keep != None: together with overflow levels."""
    from level2_helpers import compare_joints_with_single, make_joint_worlds, stack_joints, step_split_with_joints
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=bits + world_size)
    if keep is not None:
        offs = overflow_from(offs, keep)
    jkw = stack_joints(sc, 8, 4, 5, seed=world_size, damped=damped)
    single, plan, worlds = make_joint_worlds(lib, bits, sc, pm, offs, 0.0, 3, world_size, jkw)
    assert plan[0].joint_slot and plan[0].global_joints and sum(len(p.joints) for p in plan) == len(jkw["body1"])
    n_p = [len(p.peers) for p in plan]
    js = plan[0].joint_slot_index
    assert sum(int(p.send_offsets[(js + 1) * n] - p.send_offsets[js * n]) for p, n in zip(plan, n_p) if n) > 0, "the joint slot must carry bodies"
    for step in range(3):
        single.run_system("SOLVER")
        step_split_with_joints(plan, worlds, 3, False)
        compare_with_single(single, plan, worlds)
        compare_joints_with_single(single, plan, worlds)
    assert float(np.abs(single.joints_download()["total_lagrange"]).max()) > 1e-4, "the joints must have worked"


def test_joint_planners_agree():
    from helpers import hip_lib
    from level2_helpers import stack_joints
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 9, 3, 4)
    for damped in (True, False):
        jkw = stack_joints(sc, 9, 3, 4, seed=1, damped=damped)
        jp = (jkw["body1"], jkw["body2"], jkw["joint_type"], damped)
        for R in (2, 3, 4):
            want = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R, joints=jp)
            for L in (hip_lib(), lib):
                got = shard.level2_plan_lib(L, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R, joints=jp)
                for a, b in zip(got, want):
                    for f in ("bodies", "manifolds", "color_offsets", "peers", "send_offsets", "send_bodies", "recv_offsets", "recv_bodies", "n_overflow_levels", "joints", "joint_slot", "global_joints"):
                        assert np.array_equal(np.asarray(getattr(a, f)).astype(np.int64), np.asarray(getattr(b, f)).astype(np.int64)), (damped, R, L.prefix, f)
            if damped:   # every damped joint with a static body of one type shares the DUMMY pair: one owner
                for t in (F.JOINT_DISTANCE, F.JOINT_SPHERICAL):
                    g = np.flatnonzero((jkw["joint_type"] == t) & ((jkw["body1"] == 0) | (jkw["body2"] == 0)))
                    owners = {r for r, p in enumerate(want) for j in g if j in set(p.joints.tolist())}
                    assert len(owners) == 1, (t, owners)
