"""GPU: apply_local_acceleration (reference dynamics/rigid_body/forces/plugin.rs:207-241; avn_local_accelerations_upload) on HIP == oracle bit for bit,
through every kernel that carries integrate_velocities: the stand-alone system, the fused body-centric warm start (lane and quad forms), the island
blocks' LDS loop, eager launches and hipGraph replay, f32 and f64."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, assert_same, color_and_upload, hip_lib, oracle_lib, random_joints, random_world
from local_acceleration_helpers import random_local_accelerations, single_body_world
from pipeline_scenes import dropped_boxes
from test_gpu_parity import SUBSTEP_SYSTEMS, compare_all, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [32, 64])
def test_every_system_matches_oracle_with_local_accelerations(bits):
    wd = random_world(seed=21, n_bodies=300, n_manifolds=900, n_joints=100, hub_degree=30)
    lin, ang = random_local_accelerations(3, 300)
    wo, wh = make_pair(bits, substeps=3)
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
        w.local_accelerations_upload(lin, ang)
    order = ["PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"]
    order += SUBSTEP_SYSTEMS * 3
    order += ["CLEAR_VELOCITY_INCREMENTS", "SOLVE_RESTITUTION", "WRITEBACK_SOLVER_BODIES", "STORE_CONTACT_IMPULSES"]
    for k, name in enumerate(order):
        wo.run_system(name); wh.run_system(name)
        compare_all(wo, wh, f"after[{k}] {name}", joints=True)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_whole_steps_with_changing_local_accelerations(bits, use_graph):
    """The host's pattern: what it accumulated is uploaded before each step, nothing once nothing is left (steps 4, 5), only the linear half in step 3."""
    wd = random_world(seed=22, n_bodies=400, n_manifolds=1400, n_joints=150, hub_degree=30)
    wo, wh = make_pair(bits, substeps=4, use_graph=use_graph)
    for w in (wo, wh):
        color_and_upload(w, oracle_lib(), wd)
    none = None
    for s in range(7):
        lin, ang = random_local_accelerations(100 + s, 400, fraction=0.5)
        for w in (wo, wh):
            if s in (4, 5): w.local_accelerations_upload()
            elif s == 3: w.local_accelerations_upload(lin, None)
            else: w.local_accelerations_upload(lin, ang)
            w.step()
        wh.synchronize()
        compare_all(wo, wh, f"step {s}", joints=True)
    # and the values did something: the same world without them ends elsewhere
    w0 = F.World(hip_lib(), F.default_config(bits, substeps=4, use_graph=use_graph))
    color_and_upload(w0, oracle_lib(), wd)
    for s in range(7): w0.step()
    assert not np.array_equal(w0.bodies_download()["linear_velocity"], wh.bodies_download()["linear_velocity"])


@pytest.mark.parametrize("bits", [32, 64])
def test_closed_loop_pile_with_thrusters(bits):
    """Device closed loop (quad warm start fused with integrate_velocities): boxes that push along their own axes while they fall, land and tumble."""
    bodies, colliders = dropped_boxes(seed=12, n=64)
    n = len(bodies["position"])
    lin, ang = random_local_accelerations(9, n, fraction=0.5)
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(bits, substeps=4))
        w.bodies_upload(**bodies); w.colliders_upload(**colliders)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        w.local_accelerations_upload(lin, ang)
        worlds.append(w)
    wo, wh = worlds
    for s in range(40):
        wo.step(); wh.step()
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo: assert_same(bo[k], bh[k], f"step {s}: bodies.{k}")
    assert wh.timers().contact_constraint_count > 0


def test_island_blocks_carry_local_accelerations():
    """Many Pyramids in the closed loop runs its substeps inside k_island_substeps (bodies in LDS): delta_rotation is read from the block's copy."""
    sc = scenes.many_pyramids(5, 3, 3)
    n = len(sc.body_kwargs()["position"])
    lin, ang = random_local_accelerations(4, n, fraction=0.3)
    lin *= 0.25; ang *= 0.25
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(32, substeps=4))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        w.local_accelerations_upload(lin, ang)
        worlds.append(w)
    wo, wh = worlds
    for s in range(20):
        wo.step(); wh.step()
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo: assert_same(bo[k], bh[k], f"step {s}: bodies.{k}")
    assert wh.timers().island_blocks >= 2


def test_kinematic_custom_and_errors_on_the_device():
    for rb, flags in ((F.RB_KINEMATIC, 0), (F.RB_DYNAMIC, F.BODY_CUSTOM_VEL), (F.RB_STATIC, 0), (F.RB_DYNAMIC, 0)):
        out = []
        for lib in (oracle_lib(), hip_lib()):
            w = single_body_world(lib, 32, [0, 0, 0], [0.1, 0.2, 0.3, 0.9273618], [0.5, 0, 0], [0, 1.0, 0], rb_type=rb, flags=flags, locked=0x10)
            w.local_accelerations_upload(np.array([[6.0, 2.0, 0]]), np.array([[0, 0.5, 1.0]]))
            w.step(); w.step()
            out.append(w.bodies_download())
        for k in out[0]: assert_same(out[0][k], out[1][k], f"rb {rb} flags {flags}: {k}")
    w = single_body_world(hip_lib(), 32, [0, 0, 0], [0, 0, 0, 1.0], [0, 0, 0], [0, 0, 0])
    with pytest.raises(Exception):
        w.local_accelerations_upload(np.zeros((2, 3)), np.zeros((2, 3)))
    # another body count drops the values (as the oracle does)
    wd = random_world(seed=5, n_bodies=60, n_manifolds=90); wd2 = random_world(seed=5, n_bodies=61, n_manifolds=90)
    lin, ang = random_local_accelerations(1, 60)
    outs = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(32, substeps=3)); color_and_upload(w, oracle_lib(), wd)
        w.local_accelerations_upload(lin, ang); w.step()
        color_and_upload(w, oracle_lib(), wd2); w.step()
        outs.append(w.bodies_download())
    for k in outs[0]: assert_same(outs[0][k], outs[1][k], k)


def test_despawn_drops_the_values_and_the_host_uploads_them_for_what_remains():
    """avn_despawn renumbers the bodies: the library drops the local accelerations (header) and the host brings them back for the compacted set."""
    from test_gpu_despawn import despawn_both, make_pair as make_loop_pair
    bodies, colliders = dropped_boxes(seed=43, n=48)
    n = len(bodies["inv_mass"])
    lin, ang = random_local_accelerations(11, n, fraction=0.5)
    wo, wh = make_loop_pair(bodies, colliders)
    for w in (wo, wh): w.local_accelerations_upload(lin, ang)
    for s in range(20):
        wo.step(); wh.step()
    gone = [5, 17, 18, 30]
    keep = np.ones(n, bool); keep[gone] = False
    despawn_both((wo, wh), bodies, colliders, gone)
    for s in range(3):   # nothing uploaded yet: the remaining bodies coast without their thrusters, on both sides alike
        wo.step(); wh.step()
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo: assert_same(bo[k], bh[k], f"after despawn, step {s}: bodies.{k}")
    for w in (wo, wh): w.local_accelerations_upload(lin[keep], ang[keep])
    for s in range(15):
        wo.step(); wh.step()
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo: assert_same(bo[k], bh[k], f"values back, step {s}: bodies.{k}")


@pytest.mark.parametrize("bits,world_size", [(32, 2), (64, 3)])
def test_level2_slabs_carry_local_accelerations_on_hip(bits, world_size):
    from test_local_accelerations_cpu import _split_with_thrusters
    _split_with_thrusters(hip_lib(), oracle_lib(), bits, world_size)


def test_device_sharded_closed_loop_carries_local_accelerations_on_hip():
    from test_local_accelerations_cpu import _dshard_with_thrusters
    _dshard_with_thrusters(hip_lib(), 32, 50)
