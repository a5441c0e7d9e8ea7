"""GPU: child colliders on the HIP backend (include/avian_mi355x.h "child colliders").  A collider on a child entity gets its Position / Rotation from its body's
wherever the kernels read a collider pose (k_update_aabb, k_host_aabb_queries, np_update_pair's HS instantiations: reference collider_transform/plugin.rs:62-91,
collider/backend.rs:569-586, narrow_phase/system_param.rs:540-575).  Compound bodies on HIP == the oracle bit for bit: ColliderAabbs, new pairs with order, colour lists,
contact rows, bodies; with host-shaped children and with hooks on top; a world WITHOUT children is untouched by the feature (its kernels are the plain instantiations)."""
import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from compound_helpers import assert_same_compound_step, compound_scene, compound_world
from hook_helpers import Hooks
from host_shape_helpers import HostShapes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,seed,n,steps", [(32, 1, 24, 80), (64, 2, 24, 80), (32, 3, 300, 50), (64, 4, 200, 40)])
def test_compound_bodies_equal_the_oracle(bits, seed, n, steps):
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders, tf = compound_scene(seed=seed, n_bodies=n)
    dev = compound_world(hip, bits, bodies, colliders, tf)
    ref = compound_world(orc, bits, bodies, colliders, tf)
    dev.pipeline_enable(); ref.pipeline_enable()
    most = 0
    for step in range(steps):
        dev.step(); ref.step()
        assert_same_compound_step(dev, ref, step)
        most = max(most, dev.pipeline_stats().manifolds)
    assert most > 20, "compounds must have landed on the slab and on each other"


def test_no_speculative_margin_and_the_host_bookkeeping_mode():
    """speculative margin 0 (the un-swept AABB branch) through AVN_SYS_UPDATE_AABB / COLLECT_COLLISION_PAIRS / NARROW_PHASE / SOLVER of a host that keeps its own graphs."""
    from avian_amd.pipeline import ContactPipeline
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders, tf = compound_scene(seed=6, n_bodies=30)
    dev = compound_world(hip, 32, bodies, colliders, tf, speculative_margin=0.0)
    ref = compound_world(orc, 32, bodies, colliders, tf, speculative_margin=0.0)
    pd, pr = ContactPipeline(dev, hip), ContactPipeline(ref, orc)
    for step in range(60):
        pd.step(); pr.step()
        a, b = dev.bodies_download(), ref.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        mna, mxa, _ = dev.aabbs_download(); mnb, mxb, _ = ref.aabbs_download()
        assert np.array_equal(mna, mnb) and np.array_equal(mxa, mxb), f"step {step}: ColliderAabb"
        assert sorted(pd.pairs) == sorted(pr.pairs)
    assert len(pd.pairs) > 20


def test_host_shaped_children_with_hooks_on_top():
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders, tf = compound_scene(seed=8, n_bodies=40)
    rng = np.random.default_rng(8)
    host = rng.random(len(colliders["shape"])) < 0.35
    flags = np.where(rng.random(len(colliders["shape"])) < 0.4, F.COLLIDER_FILTER_PAIRS | F.COLLIDER_MODIFY_CONTACTS, 0).astype(np.uint8)
    worlds, hooks = [], []
    for lib in (hip, orc):
        cols = dict(colliders, shape=np.where(host, F.SHAPE_HOST, colliders["shape"]).astype(np.uint8), collider_flags=flags)
        w = compound_world(lib, 32, bodies, cols, tf)
        hs = HostShapes(F.World(orc, F.default_config(32)), colliders["entity_index"], colliders["shape"], colliders["half_extents"])
        w.host_shapes_set(hs.aabb, hs.manifolds)
        h = Hooks(reject_mod=11)
        w.collision_hooks_set(h.filter, h.modify)
        w._keep = (hs, h)
        w.pipeline_enable()
        worlds.append(w); hooks.append(h)
    dev, ref = worlds
    for step in range(70):
        dev.step(); ref.step()
        assert not dev.host_shape_errors() and not ref.host_shape_errors()
        assert_same_compound_step(dev, ref, step)
    assert hooks[0].filter_log == hooks[1].filter_log and hooks[0].modify_log == hooks[1].modify_log and len(hooks[0].modify_log) > 50


def test_a_world_without_children_is_untouched():
    """All-zero is_child flags == never uploading transforms == the plain world (bit for bit, HIP)."""
    hip = hip_lib()
    bodies, colliders, tf = compound_scene(seed=10, n_bodies=30)
    none = dict(tf, is_child=np.zeros_like(tf["is_child"]))
    a = compound_world(hip, 32, bodies, colliders, none)
    b = F.World(hip, F.default_config(32, substeps=4))
    b.bodies_upload(**bodies); b.colliders_upload(**colliders); b.existing_pairs_upload(np.zeros(0, np.uint64)); b.collider_materials_upload(friction=0.6, restitution=0.0)
    a.pipeline_enable(); b.pipeline_enable()
    for step in range(40):
        a.step(); b.step()
        assert_same_compound_step(a, b, step)
