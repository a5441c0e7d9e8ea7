"""CPU: the checker of avn_islands_get / avn_sleep_update (oracle/avo_world.hpp) against independent statements of the same thing --
scipy's connected components for the islands, a line-by-line numpy restatement of update_sleeping_states (reference
dynamics/solver/islands/sleeping.rs:184-241) for the timers and the resting decision.  (The reference holds no test of its own for this path: parity is pinned on
the restatement of the cited lines only.)"""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from avian_amd import scenes
from helpers import F, oracle_lib
from level2_helpers import global_problem, make_single

NONE = 0xFFFFFFFF


def scipy_islands(rb_type, b1, b2):
    n = len(rb_type)
    node = rb_type != F.RB_STATIC
    keep = node[b1] & node[b2]
    g = coo_matrix((np.ones(keep.sum()), (b1[keep], b2[keep])), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    lab = np.full(n, NONE, np.uint32)
    first = {}
    for b in range(n):
        if node[b]:
            lab[b] = first.setdefault(comp[b], b)
    return lab


def numpy_sleep_step(timer, lab, lin, ang, dt, tts=0.5, lt=0.15, at=0.15, unit=1.0, scalar=np.float32):
    """update_sleeping_states + sleep_islands' decision, restated (f32 timers; comparisons in the world's Scalar)."""
    lt2 = np.float32(lt) * np.abs(np.float32(lt)); at2 = np.float32(at) * np.abs(np.float32(at))
    u2 = scalar(unit) * scalar(unit)
    lin = lin.astype(scalar); ang = ang.astype(scalar)
    v2 = (lin[:, 0] * lin[:, 0] + lin[:, 1] * lin[:, 1]) + lin[:, 2] * lin[:, 2]
    w2 = (ang[:, 0] * ang[:, 0] + ang[:, 1] * ang[:, 1]) + ang[:, 2] * ang[:, 2]
    rest = (v2 < u2 * scalar(lt2)) & (w2 < scalar(at2))
    node = lab != NONE
    t = np.where(rest, (timer + np.float32(dt)).astype(np.float32), np.float32(0)).astype(np.float32)
    t[~node] = 0
    awake = np.zeros(len(lab), bool)
    awake[lab[node & (t < np.float32(tts))]] = True
    rests = node & ~awake[np.where(node, lab, 0)]
    return t, rests.astype(np.uint8)


@pytest.mark.parametrize("bits", [32, 64])
def test_islands_equal_scipy_and_sleeping_equals_the_restatement(bits):
    lib = oracle_lib()
    sc = scenes.box_stacks(3, 3, 2, 3, gap=4.0)          # three separate stacks on one ground
    pairs = scenes.brute_force_pairs(sc)
    mf = scenes.axis_aligned_manifolds(sc, pairs)
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    pm = scenes.permute_manifolds(mf, perm)
    w = F.World(lib, F.default_config(bits, substeps=2))
    w.bodies_upload(**sc.body_kwargs())
    scenes.upload_manifolds(w, pm, offs, sc.friction, 0.0)
    lab, n = w.islands_get()
    want = scipy_islands(sc.rb_type, pm["body1"], pm["body2"])
    assert np.array_equal(lab, want) and n == 3 == len(set(want[want != NONE].tolist()))
    assert lab[0] == NONE, "the static ground joins no island although every stack stands on it"
    timer = np.zeros(sc.n, np.float32)
    scalar = np.float32 if bits == 32 else np.float64
    slept = False
    for step in range(45):
        w.run_system("SOLVER")
        st = w.sleep_update(delta_secs=1.0 / 60.0)
        got = w.sleep_get()
        sb = w.solver_bodies_download()
        timer, rests = numpy_sleep_step(timer, want, sb["linear_velocity"], sb["angular_velocity"], 1.0 / 60.0, scalar=scalar)
        assert np.array_equal(got["sleep_timer"], timer), f"step {step}: timers"
        assert np.array_equal(got["island_rests"], rests), f"step {step}: resting decision"
        assert np.array_equal(got["island"], want)
        assert st.n_islands == 3 and st.n_island_bodies == sc.n - 1
        assert st.n_resting_bodies == int(rests.sum()) and st.n_awake_bodies == sc.n - 1 and st.n_sleeping_bodies == 0 and st.n_waking_islands == 0
        assert st.n_resting_islands == len(set(want[rests.astype(bool)].tolist()))
        slept |= st.n_resting_islands > 0
    assert slept, "a resting stack must reach TimeToSleep within 45 steps (0.5 s = 30 steps)"
    # waking: SleepTimer = 0 for the island's bodies
    isl = np.flatnonzero(want == want[1])
    w.sleep_reset(isl)
    timer[isl] = 0
    w.run_system("SOLVER")
    w.sleep_update(delta_secs=1.0 / 60.0)
    sb = w.solver_bodies_download()
    timer, rests = numpy_sleep_step(timer, want, sb["linear_velocity"], sb["angular_velocity"], 1.0 / 60.0, scalar=scalar)
    got = w.sleep_get()
    assert np.array_equal(got["sleep_timer"], timer) and np.array_equal(got["island_rests"], rests)
    assert not rests[isl].any() and rests.any()


def test_joints_and_kinematic_bodies_link_islands_static_ones_do_not():
    lib = oracle_lib()
    sc = scenes.box_stacks(2, 1, 1, 1, gap=6.0)            # ground + two single boxes far apart
    assert sc.n == 3
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**sc.body_kwargs())
    assert w.islands_get()[1] == 2
    J = dict(body1=np.array([1], np.int32), body2=np.array([2], np.int32), local_anchor1=np.zeros((1, 3)), local_anchor2=np.zeros((1, 3)),
             limit_min=np.array([0.0]), limit_max=np.array([100.0]), compliance=np.array([0.0]))
    w.distance_joints_upload(**J)
    lab, n = w.islands_get()
    assert n == 1 and lab.tolist() == [NONE, 1, 1]
    J["body2"] = np.array([0], np.int32)                    # a joint to the static ground links nothing
    w.distance_joints_upload(**J)
    assert w.islands_get()[1] == 2
    b = sc.body_kwargs()
    b["rb_type"] = np.array([F.RB_KINEMATIC, F.RB_DYNAMIC, F.RB_DYNAMIC], np.uint8)   # ... a kinematic one does (it has a BodyIslandNode)
    w2 = F.World(lib, F.default_config(32, substeps=1))
    w2.bodies_upload(**b)
    w2.distance_joints_upload(**J)
    lab, n = w2.islands_get()
    assert n == 2 and lab.tolist() == [0, 0, 2]


def test_sleeping_and_disabled_bodies():
    """Bodies the host has put to sleep (AVN_BODY_SLEEPING) are skipped by update_sleeping_states (`Without<Sleeping>`) but keep their island
    node: an island of sleepers is neither resting nor waking; once an awake, moving body is linked to it, it wakes (sleeping.rs:258-261).
    A disabled body has no island node at all (islands/mod.rs:124-135)."""
    lib = oracle_lib()
    n = 7   # ground + 6 boxes in a row, each its own island until joints link them
    sc = scenes.box_stacks(6, 1, 1, 1, gap=3.0)
    assert sc.n == n
    b = sc.body_kwargs()
    flags = np.zeros(n, np.uint8)
    flags[1] = flags[2] = F.BODY_SLEEPING      # island {1, 2}: asleep
    flags[5] = F.BODY_DISABLED                 # no node
    b["body_flags"] = flags
    v = np.zeros((n, 3)); v[3] = [1.0, 0, 0]   # body 3 moves; 4 rests; 6 rests
    b["linear_velocity"] = v
    b["gravity_scale"] = np.zeros(n)
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**b)
    J = dict(body1=np.array([1, 5], np.int32), body2=np.array([2, 6], np.int32), local_anchor1=np.zeros((2, 3)), local_anchor2=np.zeros((2, 3)),
             limit_min=np.zeros(2), limit_max=np.full(2, 100.0), compliance=np.zeros(2))
    w.distance_joints_upload(**J)
    w.run_system("PREPARE_SOLVER_BODIES")
    st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5)
    g = w.sleep_get()
    assert g["island"].tolist() == [NONE, 1, 1, 3, 4, NONE, 6], "the joint to the disabled body links nothing"
    assert (st.n_islands, st.n_island_bodies, st.n_sleeping_bodies, st.n_awake_bodies) == (4, 5, 2, 3)
    assert g["island_rests"].tolist() == [0, 0, 0, 0, 1, 0, 1] and g["island_wakes"].tolist() == [0] * 7
    assert (st.n_resting_islands, st.n_resting_bodies, st.n_waking_islands, st.n_waking_bodies) == (2, 2, 0, 0)
    assert g["sleep_timer"].tolist() == [0, 0, 0, 0, 1, 0, 1], "sleepers' timers are not advanced"
    # a joint now links the moving body 3 to the sleeping island: it has to wake
    J = dict(body1=np.array([1, 2], np.int32), body2=np.array([2, 3], np.int32), local_anchor1=np.zeros((2, 3)), local_anchor2=np.zeros((2, 3)),
             limit_min=np.zeros(2), limit_max=np.full(2, 100.0), compliance=np.zeros(2))
    w.distance_joints_upload(**J)
    st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5)
    g = w.sleep_get()
    assert g["island"].tolist() == [NONE, 1, 1, 1, 4, NONE, 6]
    assert g["island_wakes"].tolist() == [0, 1, 1, 1, 0, 0, 0] and (st.n_waking_islands, st.n_waking_bodies) == (1, 2)


def test_threshold_signs_and_length_unit():
    """`sleep_threshold.linear * sleep_threshold.linear.abs()`: a negative threshold never lets a body rest; the length unit scales the linear one."""
    lib = oracle_lib()
    sc = scenes.box_stacks(1, 1, 1, 1)
    b = sc.body_kwargs()
    b["linear_velocity"] = np.array([[0, 0, 0], [0.2, 0, 0]], float)
    b["inv_mass"] = np.array([0.0, 1.0]); b["gravity_scale"] = np.zeros(2) if "gravity_scale" in b else None
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**{k: v for k, v in b.items() if v is not None})
    w.run_system("PREPARE_SOLVER_BODIES")
    for kw, rests in ((dict(), False), (dict(length_unit=2.0), True), (dict(length_unit=2.0, linear_threshold=-0.15), False), (dict(linear_threshold=0.3), True),
                      (dict(linear_threshold=0.3, angular_threshold=-1.0), False)):
        w.sleep_reset()
        st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5, **kw)
        assert bool(st.n_resting_bodies) == rests, kw


def test_partitioner_fed_by_the_library_islands():
    """shard.plan_from_world (labels from avn_islands_get) == shard.plan (edges from the host): same islands, same ranks."""
    from avian_amd import shard
    lib = oracle_lib()
    sc = scenes.box_stacks(5, 2, 2, 2, gap=4.0)
    pairs = scenes.brute_force_pairs(sc)
    mf = scenes.axis_aligned_manifolds(sc, pairs)
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    pm = scenes.permute_manifolds(mf, perm)
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**sc.body_kwargs())
    scenes.upload_manifolds(w, pm, offs, sc.friction, 0.0)
    a = shard.plan(lib, sc.rb_type, sc.position, np.stack([pm["body1"], pm["body2"]], axis=1), 2)
    b = shard.plan_from_world(lib, w, sc.rb_type, sc.position, 2)
    assert a.n_islands == b.n_islands == 5
    assert np.array_equal(a.island_of_body, b.island_of_body) and np.array_equal(a.rank_of_body, b.rank_of_body)


def test_per_body_thresholds_and_sleeping_disabled():
    """SleepThreshold per body and SleepingDisabled (sleeping.rs:164-241): a disabled body never accumulates time and keeps its whole island
    awake; a body with a larger threshold rests while its equally fast neighbour does not."""
    lib = oracle_lib()
    sc = scenes.box_stacks(4, 1, 1, 1, gap=3.0)     # ground + 4 separate boxes
    b = sc.body_kwargs()
    v = np.zeros((sc.n, 3)); v[1:] = [0.2, 0, 0]
    b["linear_velocity"] = v; b["gravity_scale"] = np.zeros(sc.n)
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**b)
    J = dict(body1=np.array([3], np.int32), body2=np.array([4], np.int32), local_anchor1=np.zeros((1, 3)), local_anchor2=np.zeros((1, 3)),
             limit_min=np.zeros(1), limit_max=np.full(1, 100.0), compliance=np.zeros(1))
    w.distance_joints_upload(**J)
    w.run_system("PREPARE_SOLVER_BODIES")
    lin = np.array([0.15, 0.15, 0.3, 0.3, 0.3], np.float32)      # body 1 keeps the default and is too fast; 2, 3, 4 may rest at |v| = 0.2
    off = np.array([0, 0, 0, 1, 0], np.uint8)                    # body 3 has SleepingDisabled: island {3, 4} stays awake
    st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5, body_linear_threshold=lin, body_sleeping_disabled=off)
    g = w.sleep_get()
    assert g["island"].tolist() == [NONE, 1, 2, 3, 3]
    assert g["sleep_timer"].tolist() == [0, 0, 1, 0, 1] and g["island_rests"].tolist() == [0, 0, 1, 0, 0]
    assert (st.n_resting_islands, st.n_resting_bodies) == (1, 1)
    ang = np.array([0.15, 0.15, -0.15, 0.15, 0.15], np.float32)   # a negative angular threshold never lets body 2 rest
    st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5, body_linear_threshold=lin, body_angular_threshold=ang, body_sleeping_disabled=off)
    assert w.sleep_get()["sleep_timer"].tolist() == [0, 0, 0, 0, 2] and st.n_resting_islands == 0
