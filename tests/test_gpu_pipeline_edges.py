"""GPU: edge cases of the narrow-phase / closed-loop ABI — empty and degenerate worlds, error returns, mode switches —
each against the oracle driven identically."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
from pipeline_scenes import dropped_boxes

pytestmark = pytest.mark.gpu


def both(bits=32, substeps=3):
    return [F.World(lib, F.default_config(bits, substeps=substeps)) for lib in (oracle_lib(), hip_lib())]


def upload(w, bodies, colliders):
    w.bodies_upload(**bodies); w.colliders_upload(**colliders)
    w.existing_pairs_upload(np.zeros(0, np.uint64))


def same_bodies(wo, wh, what):
    bo, bh = wo.bodies_download(), wh.bodies_download()
    for k in bo:
        assert np.array_equal(bo[k], bh[k]), f"{what}: bodies.{k}"


def test_pipeline_with_no_contacts_at_all():
    """Two bodies far apart: no pairs, no rows, no manifolds; the step is pure integration."""
    bodies, colliders = dropped_boxes(seed=1, n=2, balls=False)
    bodies["position"][1] = [50.0, 30.0, 0.0]; bodies["position"][2] = [-50.0, 60.0, 0.0]
    wo, wh = both()
    for w in (wo, wh):
        upload(w, bodies, colliders); w.pipeline_enable()
        for _ in range(5):
            w.step()
        st = w.pipeline_stats()
        assert st.pairs_added == 0 and st.manifolds == 0 and st.active_pairs == 0
    same_bodies(wo, wh, "no contacts")


def test_pipeline_all_static_world():
    bodies, colliders = dropped_boxes(seed=2, n=6, balls=False)
    bodies["rb_type"][:] = F.RB_STATIC; bodies["inv_mass"][:] = 0.0; bodies["inv_inertia_local"][:] = 0.0
    bodies["linear_velocity"][:] = 0.0; bodies["angular_velocity"][:] = 0.0
    wo, wh = both()
    for w in (wo, wh):
        upload(w, bodies, colliders); w.pipeline_enable()
        for _ in range(3):
            w.step()
        assert w.pipeline_stats().manifolds == 0
    same_bodies(wo, wh, "all static")


def test_error_returns_of_the_contact_table_calls():
    bodies, colliders = dropped_boxes(seed=3, n=4, balls=False)
    for w in both():
        upload(w, bodies, colliders)
        ents = colliders["entity_index"]
        with pytest.raises(F.AvnError):
            w.contact_pairs_add([0], [ents[1]], [999999], [F.PAIR_GENERATE_CONSTRAINTS])   # unknown collider
        w.contact_pairs_add([3], [ents[1]], [ents[2]], [F.PAIR_GENERATE_CONSTRAINTS])
        with pytest.raises(F.AvnError):
            w.contact_pairs_add([3], [ents[1]], [ents[3]], [F.PAIR_GENERATE_CONSTRAINTS])   # id in use
        with pytest.raises(F.AvnError):
            w.contact_pairs_remove([2])                                                          # no such row
        with pytest.raises(F.AvnError):
            w.active_pairs_set([7])
        with pytest.raises(F.AvnError):
            w.manifold_handles_upload(np.array([0] * 24 + [1], np.uint32), np.array([9], np.uint32))
        w.active_pairs_set([3])
        w.run_system("UPDATE_AABB"); w.run_system("NARROW_PHASE")
        w.contact_pairs_remove([3])
        w.active_pairs_set(np.zeros(0, np.uint32))
        w.manifold_handles_upload(np.zeros(25, np.uint32), np.zeros(0, np.uint32))   # an empty ConstraintGraph is valid
        w.run_system("SOLVER")
        import ctypes as C
        bad = F.avn_collider_materials(2, None, None, None, None)                                # wrong count
        assert w.lib.fn("collider_materials_upload")(w.handle, C.byref(bad)) == 1                # AVN_ERR_BAD_ARG


def test_switching_between_handle_mode_and_uploaded_manifolds():
    """A world can leave the closed loop and go back to host-uploaded manifolds (and its captured substep graph follows)."""
    sc = scenes.box_stack(4, 3, 4)
    wo, wh = both(substeps=4)
    results = []
    for w, lib in ((wo, oracle_lib()), (wh, hip_lib())):
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        for _ in range(3):
            w.step()
        w.pipeline_enable(False)
        # host manifolds again: the synthetic face manifolds of the same pairs
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        p = w.pairs_get()
        mf = scenes.axis_aligned_manifolds(sc, np.stack([p["body1"], p["body2"]], axis=1))
        offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
        scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, sc.restitution)
        for _ in range(3):
            w.step()
        w.pipeline_enable()
        for _ in range(2):
            w.step()
        results.append((w.bodies_download(), w.pipeline_handles()))
    (bo, ho), (bh, hh) = results
    for k in bo:
        assert np.array_equal(bo[k], bh[k]), k
    assert np.array_equal(ho[0], hh[0]) and np.array_equal(ho[1], hh[1])


def test_f64_box_stack_closed_loop_matches_oracle():
    sc = scenes.box_stack(5, 4, 5)
    wo, wh = both(bits=64, substeps=4)
    for w in (wo, wh):
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5, restitution=0.3)
        w.pipeline_enable()
    for s in range(10):
        wo.step(); wh.step()
        same_bodies(wo, wh, f"f64 step {s}")
    assert wh.pipeline_stats().manifolds > 100
