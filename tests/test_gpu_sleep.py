"""GPU: avn_islands_get / avn_sleep_update on the HIP backend equal the oracle's -- labels, SleepTimers (f32 bit patterns), the resting
decision and the counters -- with uploaded manifolds (f32 and f64), in the device closed loop (islands of the touching pairs the pipeline
holds, changing every step), on cfg2's single 100 000-body island and on 5 000 separate islands."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
from test_sleep_cpu import NONE, scipy_islands

pytestmark = pytest.mark.gpu


def both(bits, substeps=2):
    return [F.World(lib, F.default_config(bits, substeps=substeps)) for lib in (hip_lib(), oracle_lib())]


def same_sleep_state(h, o, step):
    a, b = h.sleep_get(), o.sleep_get()
    for k in a:
        assert np.array_equal(a[k], b[k]), f"step {step}: {k}"


@pytest.mark.parametrize("bits", [32, 64])
def test_uploaded_manifolds(bits):
    lib = oracle_lib()
    sc = scenes.box_stacks(4, 3, 3, 3, gap=4.0)
    pairs = scenes.brute_force_pairs(sc)
    mf = scenes.axis_aligned_manifolds(sc, pairs)
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    pm = scenes.permute_manifolds(mf, perm)
    h, o = both(bits)
    for w in (h, o):
        w.bodies_upload(**sc.body_kwargs())
        scenes.upload_manifolds(w, pm, offs, sc.friction, 0.0)
    la, na = h.islands_get(); lb, nb = o.islands_get()
    assert na == nb == 4 and np.array_equal(la, lb) and np.array_equal(la, scipy_islands(sc.rb_type, pm["body1"], pm["body2"]))
    rested = 0
    for step in range(40):
        h.run_system("SOLVER"); o.run_system("SOLVER")
        sa, sb = h.sleep_update(linear_threshold=0.3), o.sleep_update(linear_threshold=0.3)   # (frozen manifolds leave |v| ~ g dt = 0.16 > the default 0.15)
        for f, _ in sa._fields_:
            assert getattr(sa, f) == getattr(sb, f), f"step {step}: stats.{f}"
        same_sleep_state(h, o, step)
        rested = max(rested, sa.n_resting_islands)
    assert rested == 4
    h.sleep_reset(np.array([1, 2, 3])); o.sleep_reset(np.array([1, 2, 3]))
    h.sleep_update(); o.sleep_update()
    same_sleep_state(h, o, "after reset")


def test_closed_loop_islands_follow_the_contact_graph():
    """A grid of boxes dropped on the ground: every box is its own island until piles form; labels and decisions stay the oracle's while
    the pipeline's touching set changes."""
    sc = scenes.falling_grid(6, 1.05, 0.6)     # 216 boxes, nearly touching columns
    h, o = both(32, 2)
    for w in (h, o):
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
    counts = []
    for step in range(50):
        h.step(); o.step()
        sa, sb = h.sleep_update(), o.sleep_update()
        for f, _ in sa._fields_:
            assert getattr(sa, f) == getattr(sb, f), f"step {step}: stats.{f}"
        same_sleep_state(h, o, step)
        counts.append(sa.n_islands)
    assert min(counts) < max(counts), "contacts forming and breaking must change the island count"


def test_one_big_island_and_many_small_ones():
    lib = hip_lib()
    # cfg2-sized single island
    sc = scenes.box_stack(50, 40, 50)
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get()
    mf = scenes.axis_aligned_manifolds(sc, np.stack([p["body1"], p["body2"]], axis=1))
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    pm = scenes.permute_manifolds(mf, perm)
    scenes.upload_manifolds(w, pm, offs, sc.friction, 0.0)
    lab, n = w.islands_get()
    assert n == 1 and lab[0] == NONE and np.all(lab[1:] == 1)
    st = w.sleep_update()
    assert st.n_islands == 1 and st.n_island_bodies == 100000
    # 5 000 islands of two bodies
    m = 5000
    rng = np.random.default_rng(1)
    a = rng.permutation(2 * m)[:m] + 1; rest = np.setdiff1d(np.arange(1, 2 * m + 1), a); b = rng.permutation(rest)
    b1 = np.concatenate([a, np.zeros(50, np.int64)]); b2 = np.concatenate([b, rng.integers(1, 2 * m, 50)])     # + manifolds against the ground
    rb = np.zeros(2 * m + 1, np.uint8); rb[0] = F.RB_STATIC
    want = scipy_islands(rb, b1, b2)
    sc2 = scenes.falling_grid(1, 1.0, 1.0)
    k = sc2.body_kwargs()
    n_b = 2 * m + 1
    bodies = {kk: (np.repeat(np.asarray(v)[-1:], n_b, axis=0) if v is not None else None) for kk, v in k.items()}
    bodies["rb_type"] = rb
    bodies["inv_mass"] = np.where(rb == F.RB_STATIC, 0.0, 1.0)
    w2 = F.World(lib, F.default_config(32, substeps=1))
    w2.bodies_upload(**bodies)
    M = len(b1)
    mfs = dict(body1=b1.astype(np.int32), body2=b2.astype(np.int32), normal=np.tile([0.0, 1.0, 0.0], (M, 1)), point_count=np.ones(M, np.uint8),
               anchor1=np.zeros((M, 4, 3)), anchor2=np.zeros((M, 4, 3)), penetration=np.zeros((M, 4)), normal_speed=np.zeros((M, 4)))
    offs2, perm2 = scenes.color_manifolds(lib, {kk: v for kk, v in mfs.items()}, rb)
    scenes.upload_manifolds(w2, scenes.permute_manifolds(mfs, perm2), offs2, 0.5, 0.0)
    lab2, n2 = w2.islands_get()
    assert np.array_equal(lab2, want) and n2 == m


def test_sleeping_and_disabled_bodies_on_hip():
    """The CPU case of tests/test_sleep_cpu.py::test_sleeping_and_disabled_bodies, HIP against the oracle (f32 and f64)."""
    for bits in (32, 64):
        res = []
        for lib in (hip_lib(), oracle_lib()):
            n = 7
            sc = scenes.box_stacks(6, 1, 1, 1, gap=3.0)
            b = sc.body_kwargs()
            flags = np.zeros(n, np.uint8); flags[1] = flags[2] = F.BODY_SLEEPING; flags[5] = F.BODY_DISABLED
            b["body_flags"] = flags
            v = np.zeros((n, 3)); v[3] = [1.0, 0, 0]
            b["linear_velocity"] = v; b["gravity_scale"] = np.zeros(n)
            w = F.World(lib, F.default_config(bits, substeps=1))
            w.bodies_upload(**b)
            out = []
            for jb1, jb2 in (([1, 5], [2, 6]), ([1, 2], [2, 3])):
                J = dict(body1=np.array(jb1, np.int32), body2=np.array(jb2, np.int32), local_anchor1=np.zeros((2, 3)), local_anchor2=np.zeros((2, 3)),
                         limit_min=np.zeros(2), limit_max=np.full(2, 100.0), compliance=np.zeros(2))
                w.distance_joints_upload(**J)
                w.run_system("PREPARE_SOLVER_BODIES")
                st = w.sleep_update(delta_secs=1.0, time_to_sleep=0.5)
                out.append(([getattr(st, f) for f, _ in st._fields_], w.sleep_get()))
            res.append(out)
        for (sa, ga), (sb, gb) in zip(*res):
            assert sa == sb
            for k in ga:
                assert np.array_equal(ga[k], gb[k]), (bits, k)
        assert res[0][1][0][6:] == [1, 2], "the moving body wakes the sleeping island it was linked to"


def test_per_body_thresholds_and_sleeping_disabled_on_hip():
    rng = np.random.default_rng(3)
    sc = scenes.box_stacks(40, 1, 2, 1, gap=3.0)     # 40 islands of two stacked boxes
    b = sc.body_kwargs()
    b["linear_velocity"] = rng.normal(scale=0.1, size=(sc.n, 3)); b["angular_velocity"] = rng.normal(scale=0.06, size=(sc.n, 3)); b["gravity_scale"] = np.zeros(sc.n)
    flags = np.zeros(sc.n, np.uint8); flags[rng.random(sc.n) < 0.15] = F.BODY_SLEEPING
    b["body_flags"] = flags
    pairs = scenes.brute_force_pairs(sc)
    mf = scenes.axis_aligned_manifolds(sc, pairs)
    lin = rng.uniform(-0.05, 0.6, sc.n).astype(np.float32); ang = rng.uniform(-0.02, 0.4, sc.n).astype(np.float32); off = (rng.random(sc.n) < 0.1).astype(np.uint8)
    for bits in (32, 64):
        res = []
        for lib in (hip_lib(), oracle_lib()):
            offs, perm = scenes.color_manifolds(oracle_lib(), mf, sc.rb_type)
            w = F.World(lib, F.default_config(bits, substeps=1))
            w.bodies_upload(**b)
            scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, 0.0)
            w.run_system("PREPARE_SOLVER_BODIES")
            out = []
            for _ in range(3):
                st = w.sleep_update(delta_secs=0.3, time_to_sleep=0.5, body_linear_threshold=lin, body_angular_threshold=ang, body_sleeping_disabled=off)
                out.append(([getattr(st, f) for f, _ in st._fields_], w.sleep_get()))
            res.append(out)
        for (sa, ga), (sb, gb) in zip(*res):
            assert sa == sb, (bits, sa, sb)
            for k in ga:
                assert np.array_equal(ga[k], gb[k]), (bits, k)
        assert res[0][-1][0][4] > 0 and res[0][-1][0][6] > 0, "the scene must produce resting and waking islands"
