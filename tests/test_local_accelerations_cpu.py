"""apply_local_acceleration (reference dynamics/rigid_body/forces/plugin.rs:207-241) in the oracle: the substep system that turns AccumulatedLocalAcceleration
into velocity, in front of integrate_velocities.  The reference holds no test vector for it; the oracle is checked against a float64 derivation of a free
flight written from the reference's text, and its stated edge behaviour (kinematic bodies included, CustomVelocityIntegration excluded, the translation locks
masking BOTH vectors, the values dropped by another body count) is pinned here."""
import numpy as np
import pytest

from helpers import F, assert_same, color_and_upload, oracle_lib, random_world
from local_acceleration_helpers import free_flight_float64, random_local_accelerations, single_body_world

ROT = np.array([0.18257419, 0.36514837, 0.54772256, 0.73029674])
ROT = ROT / np.linalg.norm(ROT)


@pytest.mark.parametrize("locked", [0, 0x20, 0x18, 0x38])
def test_free_flight_matches_float64_derivation(locked):
    acc_l, acc_a = [3.0, -1.0, 0.5], [0.4, 0.2, -0.7]
    w = single_body_world(oracle_lib(), 64, [1, 2, 3], ROT, [0.5, 0, -0.25], [0.3, -0.9, 0.6], locked=locked)
    w.local_accelerations_upload(np.array([acc_l]), np.array([acc_a]))
    for _ in range(30):
        w.step()
    o = w.bodies_download()
    p, r, v, om = free_flight_float64([1, 2, 3], ROT, [0.5, 0, -0.25], [0.3, -0.9, 0.6], acc_l, acc_a, locked, 1.0 / 60.0, 5, 30)
    # 1e-6: Time<Substeps> is a Duration (whole nanoseconds: 1/60 s / 5 is off by 2e-8 relative) and the oracle's sin / cos are the library's own
    tol = dict(rtol=0, atol=1e-6)
    assert np.allclose(o["position"][0], p, **tol) and np.allclose(o["linear_velocity"][0], v, **tol)
    assert np.allclose(o["angular_velocity"][0], om, **tol)
    assert np.allclose(o["rotation"][0], r, **tol) or np.allclose(o["rotation"][0], -r, **tol)
    if locked & 0x20: assert o["linear_velocity"][0][0] == 0.5 and o["angular_velocity"][0][0] == 0.3   # the translation lock masks the ANGULAR x too


def test_a_spinning_rocket_flies_a_circle():
    """A body that spins about z at omega and pushes along its own x axis with a: the velocity vector rotates with the body, |v| stays bounded by 2 a / omega
    (a world-space push of the same size would reach a t = 10 after 2 s)."""
    omega, a = 2.0 * np.pi, 5.0
    w = single_body_world(oracle_lib(), 64, [0, 0, 0], [0, 0, 0, 1.0], [0, 0, 0], [0, 0, omega], substeps=8)
    w.local_accelerations_upload(np.array([[a, 0, 0]]), None)
    vmax = 0.0
    for _ in range(120):
        w.step()
        vmax = max(vmax, float(np.linalg.norm(w.bodies_download()["linear_velocity"][0])))
    assert 0.9 * 2 * a / omega < vmax < 1.1 * 2 * a / omega


def test_kinematic_included_custom_integration_excluded():
    for rb, flags, moved in ((F.RB_KINEMATIC, 0, True), (F.RB_DYNAMIC, F.BODY_CUSTOM_VEL, False), (F.RB_STATIC, 0, False)):
        w = single_body_world(oracle_lib(), 32, [0, 0, 0], [0, 0, 0, 1.0], [0, 0, 0], [0, 0, 0], rb_type=rb, flags=flags)
        w.local_accelerations_upload(np.array([[6.0, 0, 0]]), np.array([[0, 0, 1.0]]))
        w.step()
        o = w.bodies_download()
        assert (abs(o["linear_velocity"][0][0] - 0.1) < 1e-6) == moved, (rb, flags, o["linear_velocity"])
        assert (abs(o["angular_velocity"][0][2] - 1.0 / 60.0) < 1e-6) == moved


def test_values_stay_until_replaced_and_are_dropped_by_another_body_count():
    wd = random_world(seed=5, n_bodies=60, n_manifolds=90)
    lin, ang = random_local_accelerations(1, 60)

    def run(upload_each_step, clear_after=None):
        w = F.World(oracle_lib(), F.default_config(32, substeps=3))
        color_and_upload(w, oracle_lib(), wd)
        w.local_accelerations_upload(lin, ang)
        for s in range(4):
            if upload_each_step and s: w.local_accelerations_upload(lin, ang)
            if clear_after is not None and s == clear_after: w.local_accelerations_upload()
            w.step()
        return w.bodies_download()
    base = run(False)
    again = run(True)
    for k in base: assert_same(base[k], again[k], k)
    cleared = run(False, clear_after=0)
    w0 = F.World(oracle_lib(), F.default_config(32, substeps=3))
    color_and_upload(w0, oracle_lib(), wd)
    for _ in range(4): w0.step()
    none = w0.bodies_download()
    for k in none: assert_same(none[k], cleared[k], k)
    assert not np.array_equal(base["linear_velocity"], none["linear_velocity"])
    # wrong count refused; another body count drops the values
    w = F.World(oracle_lib(), F.default_config(32, substeps=3))
    color_and_upload(w, oracle_lib(), wd)
    with pytest.raises(Exception):
        w.local_accelerations_upload(lin[:10], ang[:10])
    w.local_accelerations_upload(lin, ang)
    wd2 = random_world(seed=5, n_bodies=61, n_manifolds=90)
    w1 = F.World(oracle_lib(), F.default_config(32, substeps=3)); color_and_upload(w1, oracle_lib(), wd2)
    color_and_upload(w, oracle_lib(), wd2)
    w.step(); w1.step()
    a, b = w.bodies_download(), w1.bodies_download()
    for k in a: assert_same(a[k], b[k], k)


def test_despawn_drops_the_values():
    """After avn_despawn of bodies the oracle holds no local acceleration until the host uploads them for the remaining bodies (header)."""
    from pipeline_scenes import dropped_boxes
    from test_despawn_cpu import subset
    bodies, colliders = dropped_boxes(seed=43, n=24)
    n = len(bodies["inv_mass"])
    lin, ang = random_local_accelerations(11, n, fraction=0.8)

    def world(b, c):
        w = F.World(oracle_lib(), F.default_config(32, substeps=4))
        w.bodies_upload(**b); w.colliders_upload(**c)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        return w
    w = world(bodies, colliders)
    w.local_accelerations_upload(lin, ang)
    for _ in range(5): w.step()
    state = w.bodies_download()
    gone = [n - 1]   # (the last body: the remaining indices keep their numbers)
    mask = np.ones(n, bool); mask[gone] = False
    cmask = mask[np.asarray(colliders["body"])]
    w.despawn(bodies=gone, collider_entities=())
    nb = subset(bodies, mask)
    for k in ("position", "rotation", "linear_velocity", "angular_velocity"): nb[k] = state[k][mask].astype(np.float64)
    nc = {k: (np.asarray(v)[cmask] if isinstance(v, np.ndarray) and len(v) == len(cmask) else v) for k, v in colliders.items()}
    w.bodies_upload(**nb); w.colliders_upload(**nc); w.collider_materials_upload(friction=0.5)
    before = w.bodies_download()
    w.step()
    after = w.bodies_download()
    # a body in free flight (far above the pile) with a thruster but no upload after the despawn: only gravity acts on it
    dv = after["linear_velocity"] - before["linear_velocity"]
    free = np.argmax(before["position"][:, 1])
    assert np.any(lin[mask][free] != 0), "pick a seed whose topmost body has a thruster"
    assert abs(dv[free][0]) < 1e-6 and abs(dv[free][2]) < 1e-6 and abs(dv[free][1] + 9.81 / 60.0) < 1e-4
    with pytest.raises(Exception):
        w.local_accelerations_upload(lin, ang)   # the old count is refused
    w.local_accelerations_upload(lin[mask], ang[mask])
    w.step()


def _split_with_thrusters(lib, ref_lib, bits, world_size):
    """One island over `world_size` slab worlds (level 2): every slab uploads the local accelerations of ITS bodies (shared bodies on both sides, like every other body
    component) and must stay bit-identical to the unsplit world."""
    from level2_helpers import compare_with_single, global_problem, make_single, make_split, step_split_in_process
    sc, pm, offs, _ = global_problem(oracle_lib(), 8, 4, 5, seed=3 + world_size)
    lin, ang = random_local_accelerations(17, sc.n, fraction=0.6)
    single = make_single(lib, bits, sc, pm, offs, 0.0, 3)
    ref = make_single(ref_lib, bits, sc, pm, offs, 0.0, 3)
    plan, worlds = make_split(lib, bits, sc, pm, offs, 0.0, 3, world_size)
    single.local_accelerations_upload(lin, ang); ref.local_accelerations_upload(lin, ang)
    for pl, w in zip(plan, worlds):
        w.local_accelerations_upload(lin[pl.bodies], ang[pl.bodies])
    assert sum(len(p.send_bodies) for p in plan) > 0
    for _ in range(3):
        single.run_system("SOLVER"); ref.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, False)
        compare_with_single(single, plan, worlds)
        compare_with_single(ref, plan, worlds)


@pytest.mark.parametrize("bits,world_size", [(32, 2), (64, 3)])
def test_level2_slabs_carry_local_accelerations(bits, world_size):
    _split_with_thrusters(oracle_lib(), oracle_lib(), bits, world_size)


def _dshard_with_thrusters(lib, bits, steps):
    """The DEVICE closed loop sharded by islands: every rank holds every body (and every body's local acceleration), simulates its own; against the single world every step."""
    import dshard_helpers as D
    from avian_amd import shard
    from test_dshard_cpu import owner_by_pile
    from test_sharded_closed_loop_cpu import piles
    bodies, colliders = piles(2, 24)
    owner = owner_by_pile(bodies, 2, 24)
    lin, ang = random_local_accelerations(23, len(bodies["inv_mass"]), fraction=0.5)
    lin *= 0.5; ang *= 0.5
    ref, ranks = D.make_worlds(lib, bits, bodies, colliders, owner, 2)
    for w in [ref] + ranks: w.local_accelerations_upload(lin, ang)
    for s in range(steps):
        ref.step()
        shard.dshard_step_in_process(ranks)
        D.compare(s, ref, ranks, owner)
    return ref


def test_device_sharded_closed_loop_carries_local_accelerations():
    ref = _dshard_with_thrusters(oracle_lib(), 32, 50)
    assert ref.pipeline_stats().manifolds_pushed > 0
