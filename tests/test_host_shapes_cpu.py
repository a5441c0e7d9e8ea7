"""CPU (oracle backend): host shapes -- colliders whose AnyCollider::aabb_with_context / contact_manifolds_with_context are answered by host callbacks
(avn_host_shapes_set) while everything else of update_aabb / update_contacts stays in the library.  A world with a third of its colliders flagged AVN_SHAPE_HOST
(the callbacks hold their real Ball / Cuboid geometry) must equal the all-native world bit for bit, step after step, in the closed loop."""
import numpy as np
import pytest

from helpers import F, oracle_lib
from host_shape_helpers import assert_same_closed_loop_step, make_pair
from pipeline_scenes import dropped_boxes


@pytest.mark.parametrize("bits,seed", [(32, 1), (64, 2)])
def test_host_flagged_colliders_equal_the_native_world(bits, seed):
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=seed, n=48)
    rng = np.random.default_rng(seed)
    host = rng.random(len(colliders["shape"])) < 0.35
    host[0] = seed % 2 == 0   # (the static ground too, in one of the cases: every pair against it is the host's)
    native, hosted, hs = make_pair(lib, lib, bits, bodies, colliders, host)
    native.pipeline_enable(); hosted.pipeline_enable()
    touched = 0
    for step in range(30):
        native.step(); hosted.step()
        assert_same_closed_loop_step(native, hosted, step)
        st = hosted.host_shape_stats()
        assert st.host_colliders == int(host.sum()) and st.last_aabb_queries == st.host_colliders
        touched = max(touched, st.last_manifolds_with_points)
    assert hs.manifold_queries > 100 and touched > 5, "host pairs must actually have been queried and have touched"
    assert float(np.abs(native.bodies_download()["linear_velocity"]).max()) > 0.01


def test_host_shapes_without_callbacks_fail_loudly():
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=3, n=8)
    cols = dict(colliders); cols["shape"] = np.full(len(colliders["shape"]), F.SHAPE_HOST, np.uint8)
    w = F.World(lib, F.default_config(32))
    w.bodies_upload(**bodies); w.colliders_upload(**cols); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.pipeline_enable()
    with pytest.raises(F.AvnError):
        w.step()
    with pytest.raises(F.AvnError):
        w.run_system("UPDATE_AABB")


def test_host_shapes_in_the_host_bookkeeping_mode():
    """An Avian integration's mode: the host keeps ContactGraph / ConstraintGraph (avian_amd.pipeline.ContactPipeline) and drives UPDATE_AABB /
    COLLECT_COLLISION_PAIRS / NARROW_PHASE / SOLVER through avn_run_system; the host pairs' status changes arrive in the same change list."""
    from avian_amd.pipeline import ContactPipeline
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=4, n=40)
    rng = np.random.default_rng(4)
    host = rng.random(len(colliders["shape"])) < 0.4
    native, hosted, hs = make_pair(lib, lib, 32, bodies, colliders, host)
    pa, pb = ContactPipeline(native, lib), ContactPipeline(hosted, lib)
    for step in range(20):
        pa.step(); pb.step()
        assert not hosted.host_shape_errors()
        a, b = native.bodies_download(), hosted.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        assert sorted(pa.pairs) == sorted(pb.pairs) and list(pa.active) == list(pb.active)
        ids = np.array(sorted(pa.pairs), np.uint32)
        ra, rb = native.contacts_download(ids), hosted.contacts_download(ids)
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), f"step {step}: contact rows.{k}"
    assert hs.manifold_queries > 50


def test_capsules_a_shape_the_library_has_no_kernel_for_come_to_rest_on_the_ground():
    """24 capsules, implemented ONLY in the host callbacks (tests/host_shape_helpers.py CapsuleShapes: AABB, capsule-ground and capsule-capsule manifolds), tumble onto
    a native cuboid ground inside the library's closed loop: they end up lying on it (centre at one radius above the slab), nothing tunnels, nothing blows up."""
    from host_shape_helpers import capsule_world
    w, cs, top = capsule_world(oracle_lib(), 32)
    for _ in range(240):
        w.step()
    assert not w.host_shape_errors()
    b = w.bodies_download()
    y = b["position"][1:, 1]
    assert np.isfinite(b["position"]).all() and cs.queries > 1000
    assert (y > top + 0.25 - 0.03).all(), "no capsule sank into the ground"
    assert (y < top + 0.25 + 0.45).all(), "every capsule lies on the ground or leans on a neighbour"
    assert np.median(y) < top + 0.25 + 0.05
    assert float(np.abs(b["linear_velocity"][1:, 1]).max()) < 0.05, "at rest vertically (a lying capsule may still roll: there is no rolling friction)"


def _combo(lib, query_lib, seed=11, steps=160):
    """Host-flagged colliders with sleeping enabled and a despawn in the middle: hosted world == native world every step."""
    bodies, colliders = dropped_boxes(seed=seed, n=30)
    rng = np.random.default_rng(seed)
    host = rng.random(len(colliders["shape"])) < 0.4
    host[5] = True   # (the body despawned below carries a host shape)
    native, hosted, hs = make_pair(lib, query_lib, 32, bodies, colliders, host)
    for w in (native, hosted):
        w.pipeline_enable(); w.sleeping_enable()
    slept = 0
    for step in range(steps):
        if step == 60:   # despawn body 5 with its collider, re-upload what remains
            keep = np.arange(len(bodies["inv_mass"])) != 5
            for w, cols in ((native, colliders), (hosted, dict(colliders, shape=np.where(host, F.SHAPE_HOST, colliders["shape"]).astype(np.uint8)))):
                st = w.bodies_download()
                w.despawn(bodies=[5])
                nb = {k: np.asarray(v)[keep] for k, v in bodies.items()}
                nb.update({k: v[keep] for k, v in st.items()})
                nc = {k: np.asarray(v)[keep] for k, v in cols.items()}
                nc["body"] = np.arange(int(keep.sum()), dtype=np.int32)
                w.bodies_upload(**nb); w.colliders_upload(**nc)
        native.step(); hosted.step()
        assert not hosted.host_shape_errors()
        a, b = native.bodies_download(), hosted.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        (oa, ha), (ob, hb) = native.pipeline_handles(), hosted.pipeline_handles()
        assert np.array_equal(oa, ob) and np.array_equal(ha, hb), f"step {step}: colour lists"
        sa, sb = native.sleeping_state(), hosted.sleeping_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), f"step {step}: sleeping state.{k}"
        slept = max(slept, int(sa["sleeping"].sum()))
    return slept, hs


def test_host_shapes_with_sleeping_and_a_despawn():
    lib = oracle_lib()
    slept, hs = _combo(lib, lib)
    assert hs.manifold_queries > 200 and slept > 0, "bodies must have fallen asleep with host shapes in the loop"
