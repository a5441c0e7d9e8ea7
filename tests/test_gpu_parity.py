"""Parity of the HIP path against the CPU oracle, through the C ABI, on identical seeded inputs.

Bar (north_star): f32/f64 dynamics within a stated tolerance, broad-phase pair lists bit-exact.  Because the
oracle and the kernels evaluate the same IEEE expressions (both built with -ffp-contract=off, correctly rounded
div/sqrt, one shared deterministic sin/cos algorithm), the dynamics are in fact compared BIT-EXACTLY here
(TOL = 0: a == b elementwise, NaN == NaN); any non-zero tolerance is written next to the assert that uses it.
"""
import numpy as np
import pytest

from helpers import F, assert_same, color_and_upload, compare_dicts, hip_lib, oracle_lib, random_joints, random_world

pytestmark = pytest.mark.gpu
TOL = 0.0  # bit-exact

SUBSTEP_SYSTEMS = ["INTEGRATE_VELOCITIES", "WARM_START", "SOLVE_CONTACTS_BIAS", "INTEGRATE_POSITIONS",
                   "SOLVE_CONTACTS_RELAX", "XPBD_SOLVE", "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING"]


def make_pair(bits, **cfgkw):
    cfg_o = F.default_config(bits, **cfgkw)
    cfg_h = F.default_config(bits, **cfgkw)
    return F.World(oracle_lib(), cfg_o), F.World(hip_lib(), cfg_h)


def compare_all(wo, wh, what, joints=False):
    compare_dicts(wo.solver_bodies_download(), wh.solver_bodies_download(), what + ":solver_bodies", TOL)
    compare_dicts(wo.constraints_download(), wh.constraints_download(), what + ":constraints", TOL)
    compare_dicts(wo.bodies_download(), wh.bodies_download(), what + ":bodies", TOL)
    compare_dicts(wo.impulses_download(), wh.impulses_download(), what + ":impulses", TOL)
    if joints:
        compare_dicts(wo.joints_download(), wh.joints_download(), what + ":joints", TOL)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("seed", [0, 1])
def test_every_system_matches_oracle(bits, seed):
    """Run the reference's schedule system by system (Appendix A order) and compare ALL state after each."""
    wd = random_world(seed=seed, n_bodies=300, n_manifolds=900, n_joints=120, hub_degree=40)
    wo, wh = make_pair(bits, substeps=3)
    color_and_upload(wo, oracle_lib(), wd)
    color_and_upload(wh, oracle_lib(), wd)
    order = ["PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"]
    order += SUBSTEP_SYSTEMS * 3
    order += ["CLEAR_VELOCITY_INCREMENTS", "SOLVE_RESTITUTION", "WRITEBACK_SOLVER_BODIES", "STORE_CONTACT_IMPULSES"]
    for k, name in enumerate(order):
        wo.run_system(name)
        wh.run_system(name)
        compare_all(wo, wh, f"after[{k}] {name}", joints=True)
    assert wh.timers().contact_constraint_count == wo.timers().contact_constraint_count > 0


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_multi_step_matches_oracle(bits, use_graph):
    """avn_step x 6 with device-resident state (no re-upload), eager launches and hipGraph replay."""
    wd = random_world(seed=7, n_bodies=400, n_manifolds=1500, n_joints=200, hub_degree=30)
    wo, wh = make_pair(bits, substeps=4, use_graph=use_graph)
    color_and_upload(wo, oracle_lib(), wd)
    color_and_upload(wh, oracle_lib(), wd)
    for s in range(6):
        wo.step()
        wh.step()
        wh.synchronize()
        compare_all(wo, wh, f"step {s}", joints=True)
    assert wh.timers().kernel_launches > 0


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_all_joint_types_match_oracle_system_by_system(bits, seed):
    """Fixed / revolute / spherical / prismatic / distance joints with random frames, axes, limits and compliances
    (SURVEY.md §8 row a23), plus contacts: every system of the schedule compared bit for bit, then whole steps."""
    wd = random_world(seed=100 + seed, n_bodies=260, n_manifolds=500, n_joints=0, hub_degree=0)
    wd["joints_generic"] = random_joints(np.random.default_rng(seed), 260, 240, with_damping=(seed != 1))
    wo, wh = make_pair(bits, substeps=3)
    color_and_upload(wo, oracle_lib(), wd)
    color_and_upload(wh, oracle_lib(), wd)
    order = ["PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"]
    order += SUBSTEP_SYSTEMS * 3
    order += ["CLEAR_VELOCITY_INCREMENTS", "SOLVE_RESTITUTION", "WRITEBACK_SOLVER_BODIES", "STORE_CONTACT_IMPULSES"]
    for k, name in enumerate(order):
        wo.run_system(name)
        wh.run_system(name)
        compare_all(wo, wh, f"after[{k}] {name}", joints=True)
    jd = wh.joints_download()
    jt = wd["joints_generic"]["joint_type"]
    for t in range(5):   # every type did real work
        assert float(np.abs(jd["total_lagrange"][jt == t]).max()) > 0.0, f"joint type {t} never produced a position impulse"
    assert float(np.abs(jd["total_rotation_lagrange"][jt != F.JOINT_DISTANCE]).max()) > 0.0 and float(np.abs(jd["torque"]).max()) > 0.0
    for s_ in range(2):   # (the inconsistent random joint configuration gains ~1000x velocity per step: stop before overflow)
        wo.step(); wh.step()
        compare_all(wo, wh, f"step {s_}", joints=True)


def test_solver_iterations_extension_matches_oracle():
    """The declared `solver_iterations` extension (outer repeats; OFF for reference parity) stays in lock-step too."""
    wd = random_world(seed=11, n_bodies=150, n_manifolds=500, n_joints=60)
    wo, wh = make_pair(32, substeps=2, solver_iterations=3)
    color_and_upload(wo, oracle_lib(), wd)
    color_and_upload(wh, oracle_lib(), wd)
    for s in range(3):
        wo.step(); wh.step()
        compare_all(wo, wh, f"iters step {s}", joints=True)


def test_overflow_colour_is_serial_and_exact():
    """> 20 dynamic neighbours on one body push manifolds into colour 23, solved serially in list order."""
    wd = random_world(seed=3, n_bodies=120, n_manifolds=100, hub_degree=80, with_odd_features=False)
    wo, wh = make_pair(32, substeps=2)
    offs, _ = color_and_upload(wo, oracle_lib(), wd)
    color_and_upload(wh, oracle_lib(), wd)
    assert offs[24] - offs[23] >= 50, "scene must populate the overflow colour"
    for s in range(4):
        wo.step(); wh.step()
        compare_all(wo, wh, f"overflow step {s}")


def test_host_constraint_graph_matches_oracle_graph():
    """The product's host ConstraintGraph (C++) yields the same colour lists as the oracle's restatement."""
    wd = random_world(seed=5, n_bodies=500, n_manifolds=4000, hub_degree=60)
    from avian_amd import scenes
    rb = np.asarray(wd["bodies"]["rb_type"])
    oo, po = scenes.color_manifolds(oracle_lib(), wd["manifolds"], rb)
    oh, ph = scenes.color_manifolds(hip_lib(), wd["manifolds"], rb)
    assert np.array_equal(oo, oh) and np.array_equal(po, ph)


# ---------------------------------------------------------------------------------------------------------------
def random_colliders(rng, n_bodies, bodies):
    n = n_bodies
    shape = (rng.random(n) < 0.5).astype(np.uint8)
    he = rng.uniform(0.2, 1.2, size=(n, 3))
    return dict(entity_index=(np.arange(n) * 3 + 17).astype(np.uint32), body=np.arange(n, dtype=np.int32), shape=shape,
                half_extents=he, memberships=rng.integers(1, 8, n).astype(np.uint32),
                filters=np.where(rng.random(n) < 0.2, rng.integers(0, 8, n), 0xFFFFFFFF).astype(np.uint32),
                collider_flags=rng.integers(0, 32, n).astype(np.uint8),
                collision_margin=np.where(rng.random(n) < 0.2, rng.uniform(0, 0.1, n), 0.0),
                speculative_margin=np.where(rng.random(n) < 0.2, rng.uniform(0, 0.5, n), -1.0))


@pytest.mark.parametrize("bits", [32, 64])
def test_broadphase_pairs_bit_exact_over_frames(bits):
    """AABBs, interval order and the emitted pair SEQUENCE equal the oracle's, across frames in which bodies move,
    colliders are added (appended unsorted) and removed (retain in place), with existing pairs filtered."""
    rng = np.random.default_rng(42)
    n = 3000
    wd = random_world(seed=9, n_bodies=n, n_manifolds=10, n_static=40, n_kinematic=10)
    wd["bodies"]["position"] = rng.uniform(-25, 25, size=(n, 3))
    wo, wh = make_pair(bits, substeps=2)
    col = random_colliders(rng, n, wd["bodies"])
    alive = np.ones(n, bool)
    alive[rng.choice(n, 300, replace=False)] = False  # start with a subset; the rest are added later
    total_pairs = 0
    for frame in range(5):
        for w in (wo, wh):
            w.bodies_upload(**wd["bodies"])
            sel = np.flatnonzero(alive)
            w.colliders_upload(**{k: v[sel] for k, v in col.items()})
            if frame == 0:
                w.existing_pairs_upload(np.zeros(0, np.uint64))
        wo.run_system("UPDATE_AABB"); wh.run_system("UPDATE_AABB")
        wo.run_system("COLLECT_COLLISION_PAIRS"); wh.run_system("COLLECT_COLLISION_PAIRS")
        mo, xo, eo = wo.aabbs_download(); mh, xh, eh = wh.aabbs_download()
        assert_same(mo, mh, f"frame {frame} aabb.min"); assert_same(xo, xh, f"frame {frame} aabb.max")
        assert np.array_equal(eo, eh), f"frame {frame}: interval order differs"
        po, ph = wo.pairs_get(), wh.pairs_get()
        assert len(po) == len(ph), f"frame {frame}: {len(po)} vs {len(ph)} pairs"
        assert np.array_equal(po, ph), f"frame {frame}: pair sequence differs"
        if frame == 0:
            assert len(po) > 1000
        total_pairs += len(po)
        # move things, wake/kill some colliders
        wd["bodies"]["position"] = wd["bodies"]["position"] + rng.normal(scale=0.4, size=(n, 3))
        wd["bodies"]["linear_velocity"] = rng.normal(scale=3.0, size=(n, 3))
        wd["bodies"]["rotation"] = wd["bodies"]["rotation"] + rng.normal(scale=0.05, size=(n, 4))
        wd["bodies"]["rotation"] /= np.linalg.norm(wd["bodies"]["rotation"], axis=1, keepdims=True)
        flip = rng.choice(n, 150, replace=False)
        alive[flip] = ~alive[flip]
        if frame == 2:  # the host forgets some pairs (narrow phase removed them): they must be re-emitted
            keys = np.array([hip_lib().pair_key(int(a), int(b)) for a, b in zip(po["collider1"], po["collider2"])], np.uint64)
            for w in (wo, wh):
                w.existing_pairs_upload(keys[: len(keys) // 2])
    assert total_pairs > 1500


def test_broadphase_long_interval_path():
    """An AABB that spans the scene (ground) has > SW_CAP (8192) sweep candidates: its tail is swept by the
    workgroup-cooperative kernel; the pair sequence must still be the reference's (i asc, j asc) order."""
    from avian_amd import scenes
    sc = scenes.box_stack(30, 6, 60)   # 10 800 cubes on one big static ground, plus a second huge static slab
    wo, wh = make_pair(32)
    for w in (wo, wh):
        w.bodies_upload(**sc.body_kwargs())
        w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    po, ph = wo.pairs_get(), wh.pairs_get()
    assert len(po) == len(ph) and np.array_equal(po, ph)
    ground_pairs = (po["body1"] == 0) | (po["body2"] == 0)
    assert ground_pairs.sum() == 30 * 60, "every bottom cube pairs with the ground"
    # second frame: nothing new
    for w in (wo, wh):
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        assert len(w.pairs_get()) == 0


def test_broadphase_edge_cases():
    """Empty world, a single collider, identical min.x (stability), -0.0 vs +0.0 keys, touching AABBs, non-finite AABB."""
    wo, wh = make_pair(32)
    n = 8
    pos = np.array([[0, 0, 0], [0, 0, 0], [-0.0, 0.5, 0], [0.0, 1.0, 0], [2.01, 0, 0], [1.0, 0, 0], [np.inf, 0, 0], [5, 5, 5]], float)
    bodies = dict(position=pos, rotation=np.tile([0, 0, 0, 1.0], (n, 1)), linear_velocity=np.zeros((n, 3)),
                  angular_velocity=np.zeros((n, 3)), inv_mass=np.ones(n), inv_inertia_local=np.tile([6, 0, 0, 6, 0, 6.0], (n, 1)),
                  rb_type=np.zeros(n, np.uint8))
    col = dict(entity_index=np.arange(n, dtype=np.uint32)[::-1].copy(), body=np.arange(n, dtype=np.int32),
               shape=np.zeros(n, np.uint8), half_extents=np.full((n, 3), 0.5))
    for w in (wo, wh):
        w.bodies_upload(**bodies)
        w.colliders_upload(**{k: v[:0] for k, v in col.items()})
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        assert len(w.pairs_get()) == 0
        w.colliders_upload(**{k: v[:1] for k, v in col.items()})
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        assert len(w.pairs_get()) == 0
        w.colliders_upload(**col)
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    po, ph = wo.pairs_get(), wh.pairs_get()
    assert np.array_equal(po, ph) and len(po) >= 5
    _, _, eo = wo.aabbs_download(); _, _, eh = wh.aabbs_download()
    assert np.array_equal(eo, eh) and len(eo) == n - 1, "the non-finite AABB interval is dropped"


@pytest.mark.parametrize("bits", [32, 64])
def test_sweep_batch_cull_with_non_finite_yz_and_ragged_group_count(bits):
    """The sweep culls whole groups of 8 sorted records against their y/z bounds (k_batch_bounds): the last group is ragged when the interval
    count is not a multiple of 8, and intervals with a non-finite y / z (dropped by update_aabb_intervals, sorted to the end) sit in the
    padded tail the bounds kernel reads.  1 003 colliders in a slab, six of them non-finite: same pair list and interval order as the oracle."""
    n = 1003
    rng = np.random.default_rng(11)
    pos = np.stack([rng.uniform(0, 6, n), rng.uniform(0, 6, n), rng.uniform(0, 6, n)], 1)
    pos[5, 1] = np.nan; pos[77, 2] = np.nan; pos[130, 1] = np.inf; pos[131, 2] = -np.inf; pos[999, 1] = np.nan; pos[1002, 2] = np.inf
    bodies = dict(position=pos, rotation=np.tile([0, 0, 0, 1.0], (n, 1)), linear_velocity=np.zeros((n, 3)), angular_velocity=np.zeros((n, 3)),
                  inv_mass=np.ones(n), inv_inertia_local=np.tile([6, 0, 0, 6, 0, 6.0], (n, 1)), rb_type=np.zeros(n, np.uint8))
    col = dict(entity_index=np.arange(n, dtype=np.uint32), body=np.arange(n, dtype=np.int32), shape=np.zeros(n, np.uint8), half_extents=np.full((n, 3), 0.35))
    wo, wh = make_pair(bits)
    for w in (wo, wh):
        w.bodies_upload(**bodies); w.colliders_upload(**col)
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    po, ph = wo.pairs_get(), wh.pairs_get()
    assert len(po) > 1000 and np.array_equal(po, ph)
    touched = set(po["collider1"].tolist()) | set(po["collider2"].tolist())
    assert not ({5, 77, 130, 131, 999, 1002} & touched), "an interval whose AABB is not finite is dropped before the sweep (broad_phase.rs:230-279)"
    _, _, eo = wo.aabbs_download(); _, _, eh = wh.aabbs_download()
    assert np.array_equal(eo, eh) and len(eo) == n - 6


def test_long_interval_chunk_overflow_grows_and_retries(monkeypatch):
    """A wall of 9 600 boxes sharing one x extent: every interval has > 8 192 sweep candidates and is cut into chunks -- more chunks
    than the (here artificially small) chunk arrays hold.  The library must grow them and run the count pass again instead of reading the
    unwritten items: same pair list as the oracle."""
    monkeypatch.setenv("AVN_SWEEP_LONG_CAP", "1024")
    ny, nz = 120, 80
    n = ny * nz
    yy, zz = np.meshgrid(np.arange(ny) * 3.0, np.arange(nz) * 3.0, indexing="ij")
    rng = np.random.default_rng(7)
    pos = np.stack([rng.uniform(-0.2, 0.2, n), yy.ravel(), zz.ravel()], 1)
    pos[::7, 1] += 2.2   # some boxes reach their neighbours
    bodies = dict(position=pos, rotation=np.tile([0, 0, 0, 1.0], (n, 1)), linear_velocity=np.zeros((n, 3)), angular_velocity=np.zeros((n, 3)),
                  inv_mass=np.ones(n), inv_inertia_local=np.tile([6, 0, 0, 6, 0, 6.0], (n, 1)), rb_type=np.zeros(n, np.uint8))
    col = dict(entity_index=np.arange(n, dtype=np.uint32), body=np.arange(n, dtype=np.int32), shape=np.zeros(n, np.uint8), half_extents=np.full((n, 3), 0.5))
    from helpers import hip_measure_lib   # (AVN_SWEEP_LONG_CAP is a test hook of the `make measure` build)
    wo, wh = F.World(oracle_lib(), F.default_config(32)), F.World(hip_measure_lib(), F.default_config(32))
    for w in (wo, wh):
        w.bodies_upload(**bodies); w.colliders_upload(**col)
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    po, ph = wo.pairs_get(), wh.pairs_get()
    assert len(po) > 500 and np.array_equal(po, ph)


def test_bad_arguments_are_reported_not_crashed():
    wh = F.World(hip_lib(), F.default_config(32))
    with pytest.raises(F.AvnError) as e:
        wh.step()
    assert e.value.status == 6  # AVN_ERR_STATE: no bodies
    wd = random_world(seed=1, n_bodies=20, n_manifolds=10)
    wh.bodies_upload(**wd["bodies"])
    mf = dict(wd["manifolds"]); mf["body1"] = mf["body1"].copy(); mf["body1"][0] = 10_000
    from avian_amd import scenes
    offs = np.zeros(25, np.uint32); offs[1:] = len(mf["body1"])
    with pytest.raises(F.AvnError) as e:
        scenes.upload_manifolds(wh, mf, offs, 0.5, 0.0)
    assert e.value.status == 1
    bad = F.default_config(32); bad.scalar_bits = 16
    with pytest.raises(F.AvnError):
        F.World(hip_lib(), bad)


@pytest.mark.parametrize("threshold", [None, "16"])
def test_reuploading_manifolds_every_step_keeps_the_substep_graph_honest(threshold, monkeypatch):
    """HostNarrowPhase mode: the host re-sends the manifold set before every step.  The substep graph survives an upload whose overflow colour keeps its captured launch
    parameters (round 6) and must be re-captured when they change: the same set twice, then hub manifolds leaving / returning (other level sizes, other component
    counts), in the per-component form and -- measure build, AVN_OVERFLOW_LEVEL_THRESHOLD -- in the per-level form whose level sizes are launch parameters."""
    from helpers import hip_measure_lib
    lib = hip_lib()
    if threshold is not None:
        monkeypatch.setenv("AVN_OVERFLOW_LEVEL_THRESHOLD", threshold)
        lib = hip_measure_lib()
    wd = random_world(seed=31, n_bodies=260, n_manifolds=500, hub_degree=70, with_odd_features=False)
    m_all = len(wd["manifolds"]["body1"])
    wo = F.World(oracle_lib(), F.default_config(32, substeps=3, use_graph=1))
    wh = F.World(lib, F.default_config(32, substeps=3, use_graph=1))
    rng = np.random.default_rng(2)
    keep_sets = [np.arange(m_all)] * 3                                                         # the same set three times: the graph is replayed
    keep_sets += [np.sort(rng.choice(m_all, m_all - 25, replace=False)) for _ in range(3)]     # manifolds leave (the hub's among them): other levels
    keep_sets += [np.arange(m_all)] * 2 + [np.arange(m_all - 70)] + [np.arange(m_all)]         # back; the hub's 70 manifolds gone altogether (no overflow colour); back
    for s, keep in enumerate(keep_sets):
        sub = dict(wd)
        sub["manifolds"] = {k: np.asarray(v)[keep] for k, v in wd["manifolds"].items()}
        for k in ("friction", "restitution", "warm_n", "warm_t"): sub[k] = wd[k][keep]
        for w in (wo, wh):
            if s == 0: color_and_upload(w, oracle_lib(), sub)
            else:
                from avian_amd import scenes
                offsets, perm = scenes.color_manifolds(oracle_lib(), sub["manifolds"], np.asarray(wd["bodies"]["rb_type"]))
                pm = scenes.permute_manifolds(sub["manifolds"], perm)
                scenes.upload_manifolds(w, pm, offsets, sub["friction"][perm], sub["restitution"][perm], warm_n=sub["warm_n"][perm], warm_t=sub["warm_t"][perm])
            w.step()
        wh.synchronize()
        compare_all(wo, wh, f"upload {s}")
