"""CPU (oracle backend): child colliders -- colliders on CHILD entities of their rigid body with a ColliderTransform (reference
collision/collider/collider_transform/plugin.rs:62-91 update_child_collider_position; collider/backend.rs:569-586 the swept AABB's velocity at the collider's offset;
narrow_phase/system_param.rs:540-575 collider_offset).  The library computes the child's pose from its body's wherever it reads one (avn_collider_transforms_upload)."""
import numpy as np
import pytest

from helpers import F, oracle_lib
from compound_helpers import assert_same_compound_step, collider_poses_f64, compound_scene, compound_world
from host_shape_helpers import HostShapes


@pytest.mark.parametrize("bits", [32, 64])
def test_child_collider_boxes_sit_where_an_independent_derivation_puts_them(bits):
    """speculative margin 0: ColliderAabb = shape AABB at the collider's pose, grown by contact_tolerance.  For balls the AABB centre IS the collider's position; for cuboids
    the box is centre +- |R| half_extents: both against float64 textbook algebra."""
    lib = oracle_lib()
    bodies, colliders, tf = compound_scene(seed=3, n_bodies=12)
    w = compound_world(lib, bits, bodies, colliders, tf, speculative_margin=0.0)
    w.pipeline_enable()
    for step in range(5):
        before = w.bodies_download()   # update_aabb reads the poses the step STARTS with
        w.step()
        mn, mx, _ = w.aabbs_download()
        pos, rot, rotm = collider_poses_f64(before, colliders, tf)
        tol = 0.005
        for i in range(len(colliders["shape"])):
            he = colliders["half_extents"][i]
            ext = np.full(3, he[0]) if colliders["shape"][i] == F.SHAPE_BALL else np.abs(rotm(rot[i])) @ he
            assert np.allclose(mn[i], pos[i] - ext - tol, atol=2e-5) and np.allclose(mx[i], pos[i] + ext + tol, atol=2e-5), (step, i)
    assert tf["is_child"].sum() > 12


def test_a_child_at_the_identity_transform_behaves_like_a_collider_on_the_body():
    """translation 0, rotation identity: the child's pose is its body's up to one rounding of the quaternion normalisation -- trajectories agree to 1e-4 over a second."""
    lib = oracle_lib()
    bodies, colliders, tf = compound_scene(seed=5, n_bodies=10)
    keep = tf["is_child"] == 0
    cols = {k: v[keep] for k, v in colliders.items()}
    n = int(keep.sum())
    on_body = dict(is_child=np.zeros(n, np.uint8), translation=np.zeros((n, 3)), rotation=np.tile([0.0, 0, 0, 1], (n, 1)))
    as_child = dict(on_body, is_child=np.r_[0, np.ones(n - 1)].astype(np.uint8))
    a = compound_world(lib, 64, bodies, cols, on_body); b = compound_world(lib, 64, bodies, cols, as_child)
    a.pipeline_enable(); b.pipeline_enable()
    for _ in range(60):
        a.step(); b.step()
    x, y = a.bodies_download(), b.bodies_download()
    assert np.abs(x["position"] - y["position"]).max() < 1e-4 and float(np.abs(x["position"][1:, 1]).max()) < 20


@pytest.mark.parametrize("bits,seed", [(32, 1), (64, 2)])
def test_compound_bodies_come_to_rest_and_their_children_carry_them(bits, seed):
    lib = oracle_lib()
    bodies, colliders, tf = compound_scene(seed=seed, n_bodies=20)
    w = compound_world(lib, bits, bodies, colliders, tf)
    w.pipeline_enable()
    child_touching = 0
    for step in range(300):
        w.step()
        if step % 50 == 49:
            offs, handles = w.pipeline_handles()
            child_touching = max(child_touching, len(handles))
    b = w.bodies_download()
    speed = np.linalg.norm(b["linear_velocity"][1:], axis=1)
    assert float(np.percentile(speed, 80)) < 0.2 and float(speed.max()) < 5.0, "the pile settles (a compound with ball children may still be rolling)"
    assert float(b["position"][1:, 1].min()) > 0.15, "nothing sank into the slab (top at y = 0): the children hold the bodies up"
    assert child_touching > 20
    # no pair between two colliders of the same body was ever created (broad_phase.rs:409-416: collider_of1 == collider_of2)
    ps = w.pipeline_stats()
    assert ps.active_pairs > 0


def test_children_whose_shape_lives_on_the_host_equal_native_children():
    """Host shapes + child colliders: the host callbacks are asked at the CHILD's pose (aabb queries with the velocity at its offset, manifold queries with both colliders' poses)."""
    lib = oracle_lib()
    bodies, colliders, tf = compound_scene(seed=7, n_bodies=16)
    rng = np.random.default_rng(7)
    host = rng.random(len(colliders["shape"])) < 0.4
    native = compound_world(lib, 32, bodies, colliders, tf)
    hosted = compound_world(lib, 32, bodies, dict(colliders, shape=np.where(host, F.SHAPE_HOST, colliders["shape"]).astype(np.uint8)), tf)
    hs = HostShapes(F.World(lib, F.default_config(32)), colliders["entity_index"], colliders["shape"], colliders["half_extents"])
    hosted.host_shapes_set(hs.aabb, hs.manifolds)
    native.pipeline_enable(); hosted.pipeline_enable()
    for step in range(70):
        native.step(); hosted.step()
        assert not hosted.host_shape_errors()
        assert_same_compound_step(native, hosted, step)
    assert hs.manifold_queries > 200 and (host & (tf["is_child"] == 1)).sum() > 5


def test_a_colliders_upload_puts_every_collider_back_on_its_body():
    lib = oracle_lib()
    bodies, colliders, tf = compound_scene(seed=9, n_bodies=6)
    w = compound_world(lib, 32, bodies, colliders, tf, speculative_margin=0.0)
    plain = F.World(lib, F.default_config(32, substeps=4))
    plain.bodies_upload(**bodies); plain.colliders_upload(**dict(colliders, speculative_margin=np.zeros(len(colliders["shape"]))))
    w.colliders_upload(**dict(colliders, speculative_margin=np.zeros(len(colliders["shape"]))))   # (no transforms upload after it)
    for x in (w, plain):
        x.run_system("UPDATE_AABB")
    a, b = w.aabbs_download(), plain.aabbs_download()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    import ctypes as C
    ch, tr, ro = np.zeros(3, np.uint8), np.zeros((3, 3), np.float32), np.zeros((3, 4), np.float32)
    bad = F.avn_collider_transforms(3, ch.ctypes.data, tr.ctypes.data, ro.ctypes.data)   # (a count that is not the last colliders_upload's)
    assert w.lib.fn("collider_transforms_upload")(w.handle, C.byref(bad)) != 0
