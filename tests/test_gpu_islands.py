"""GPU: the island-block substep kernel (k_island_substeps: whole substep loop of a block of contact islands in one workgroup,
SolverBody records staged in LDS) against the CPU oracle and against the device-wide colour launches, bit for bit.

The path is chosen by the library (f32, no joints, <= 65 536 manifolds, every island <= 512 bodies); avn_timers.island_blocks
says which one ran.  AVN_ISLAND_BLOCKS=0 (read at world creation) forces the device-wide path."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, color_and_upload, compare_dicts, hip_lib, hip_measure_lib, oracle_lib, random_world

pytestmark = pytest.mark.gpu
TOL = 0.0  # bit-exact


def compare_all(wo, wh, what):
    compare_dicts(wo.solver_bodies_download(), wh.solver_bodies_download(), what + ":solver_bodies", TOL)
    compare_dicts(wo.constraints_download(), wh.constraints_download(), what + ":constraints", TOL)
    compare_dicts(wo.bodies_download(), wh.bodies_download(), what + ":bodies", TOL)
    compare_dicts(wo.impulses_download(), wh.impulses_download(), what + ":impulses", TOL)


def clustered(wd, cluster):
    """Rewire the random manifolds so that both bodies fall in the same group of `cluster` consecutive bodies: many islands."""
    mf = wd["manifolds"]
    b1 = mf["body1"].astype(np.int64)
    base = (b1 // cluster) * cluster
    off = (mf["body2"].astype(np.int64) % (cluster - 1)) + 1
    n = len(np.asarray(wd["bodies"]["rb_type"]))
    b2 = base + (b1 - base + off) % cluster
    b2 = np.minimum(b2, n - 1)
    b2 = np.where(b2 == b1, base, b2)
    keep = b2 != b1
    mf["body2"] = b2.astype(np.int32)
    for k in list(mf):
        mf[k] = mf[k][keep]
    for k in ("friction", "restitution", "warm_n", "warm_t"):
        wd[k] = wd[k][keep]
    return wd


@pytest.mark.parametrize("case", ["one_island_with_overflow", "clusters", "clusters_small_blocks", "two_iterations"])
def test_island_blocks_match_oracle_and_device_wide_path(case, monkeypatch):
    kw = dict(substeps=4)
    if case == "one_island_with_overflow":
        wd = random_world(seed=11, n_bodies=300, n_manifolds=900, n_joints=0, hub_degree=40)   # hub: > 23 contacts -> overflow colour
    elif case == "two_iterations":
        wd = random_world(seed=12, n_bodies=200, n_manifolds=500, n_joints=0, hub_degree=30)
        kw["solver_iterations"] = 2
    else:
        wd = clustered(random_world(seed=13, n_bodies=900, n_manifolds=2400, n_joints=0, hub_degree=0), 24)
        if case == "clusters_small_blocks":
            monkeypatch.setenv("AVN_ISLAND_PACK_BODIES", "40")
    wo = F.World(oracle_lib(), F.default_config(32, **kw))
    # (environment switches exist in the `make measure` build only: the worlds that need one are created on it, the plain island-block world on the release build)
    wi = F.World(hip_measure_lib() if case == "clusters_small_blocks" else hip_lib(), F.default_config(32, **kw))
    monkeypatch.setenv("AVN_ISLAND_BLOCKS", "0")
    wd_ = F.World(hip_measure_lib(), F.default_config(32, **kw))
    monkeypatch.delenv("AVN_ISLAND_BLOCKS")
    for w in (wo, wi, wd_):
        color_and_upload(w, oracle_lib(), wd)
    for s in range(5):
        for w in (wo, wi, wd_):
            w.step()
        compare_all(wo, wi, f"{case}: island blocks vs oracle, step {s}")
        compare_all(wo, wd_, f"{case}: device-wide vs oracle, step {s}")
    ti, td = wi.timers(), wd_.timers()
    assert ti.island_blocks > 0 and td.island_blocks == 0
    if case == "clusters":
        assert ti.island_blocks >= 3
    if case == "clusters_small_blocks":
        assert ti.island_blocks >= 20
    assert ti.kernel_launches < td.kernel_launches


def test_individual_systems_still_run_device_wide_and_match():
    """run_system drives single systems (no island kernel): mixing them with whole steps must stay consistent."""
    wd = random_world(seed=14, n_bodies=150, n_manifolds=400, n_joints=0, hub_degree=0)
    wo, wh = F.World(oracle_lib(), F.default_config(32, substeps=2)), F.World(hip_lib(), F.default_config(32, substeps=2))
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
    for w in (wo, wh):
        w.step()
    compare_all(wo, wh, "after a whole step")
    for name in ["PREPARE_SOLVER_BODIES", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS", "INTEGRATE_VELOCITIES", "WARM_START",
                 "SOLVE_CONTACTS_BIAS", "INTEGRATE_POSITIONS", "SOLVE_CONTACTS_RELAX"]:
        wo.run_system(name); wh.run_system(name)
        compare_all(wo, wh, "after " + name)
    for w in (wo, wh):
        w.step()
    compare_all(wo, wh, "after another whole step")


def test_ineligible_worlds_fall_back_to_colour_launches():
    # an island of > 512 bodies
    wd = random_world(seed=15, n_bodies=900, n_manifolds=3000, n_joints=0, hub_degree=0)
    wh = F.World(hip_lib(), F.default_config(32, substeps=2)); wo = F.World(oracle_lib(), F.default_config(32, substeps=2))
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
        w.step()
    assert wh.timers().island_blocks == 0
    compare_all(wo, wh, "big island")
    # f64
    wd = random_world(seed=16, n_bodies=100, n_manifolds=200, n_joints=0, hub_degree=0)
    wh = F.World(hip_lib(), F.default_config(64, substeps=2)); wo = F.World(oracle_lib(), F.default_config(64, substeps=2))
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
        w.step()
    assert wh.timers().island_blocks == 0
    compare_all(wo, wh, "f64")
    # joints
    wd = random_world(seed=17, n_bodies=100, n_manifolds=200, n_joints=30, hub_degree=0)
    wh = F.World(hip_lib(), F.default_config(32, substeps=2)); wo = F.World(oracle_lib(), F.default_config(32, substeps=2))
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
        w.step()
    assert wh.timers().island_blocks == 0
    compare_all(wo, wh, "joints")


def test_closed_loop_many_pyramids_on_island_blocks():
    """The reference's Many Pyramids bench shape, closed loop (device broad + narrow phase, library-side constraint graph):
    contacts appear and colours drift while the island blocks are rebuilt from the live handle lists."""
    sc = scenes.many_pyramids(6, 3, 3)
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(32, substeps=4))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        worlds.append(w)
    wo, wh = worlds
    for s in range(20):
        wo.step(); wh.step()
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo:
            assert np.array_equal(bo[k], bh[k]), f"step {s}: bodies.{k}"
    assert wh.timers().island_blocks >= 2, "nine pyramids that never touch: nine islands, packed a few per block"
