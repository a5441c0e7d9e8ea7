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


def _respawn(w, bodies, colliders, new_pos, new_he, entity, collider_first):
    """Append one dynamic cuboid to a running closed-loop world: every body is re-uploaded with its CURRENT state (what an ECS host does
    each step), the collider list gains one entry — at the end, or at the front so that every existing collider changes its slot."""
    cur = w.bodies_download()
    n = len(bodies["inv_mass"])
    vol = 8.0 * float(np.prod(new_he))
    d = [vol / 3 * (new_he[1] ** 2 + new_he[2] ** 2), vol / 3 * (new_he[0] ** 2 + new_he[2] ** 2), vol / 3 * (new_he[0] ** 2 + new_he[1] ** 2)]
    nb = dict(position=np.vstack([cur["position"], [new_pos]]), rotation=np.vstack([cur["rotation"], [[0.0, 0, 0, 1]]]),
              linear_velocity=np.vstack([cur["linear_velocity"], [[0.0, -1.0, 0.0]]]), angular_velocity=np.vstack([cur["angular_velocity"], [[0.3, 0.0, 0.2]]]),
              inv_mass=np.append(bodies["inv_mass"], 1.0 / vol), inv_inertia_local=np.vstack([bodies["inv_inertia_local"], [[1 / d[0], 0, 0, 1 / d[1], 0, 1 / d[2]]]]),
              rb_type=np.append(bodies["rb_type"], F.RB_DYNAMIC).astype(np.uint8))
    ent = np.append(colliders["entity_index"], entity).astype(np.uint32); body = np.append(colliders["body"], n).astype(np.int32)
    shape = np.append(colliders["shape"], F.SHAPE_CUBOID).astype(np.uint8); he = np.vstack([colliders["half_extents"], [new_he]])
    if collider_first:
        ent, body, shape, he = np.roll(ent, 1), np.roll(body, 1), np.roll(shape, 1), np.roll(he, 1, axis=0)
    nc = dict(entity_index=ent, body=body, shape=shape, half_extents=he)
    w.bodies_upload(**nb); w.colliders_upload(**nc); w.collider_materials_upload(friction=0.6, restitution=0.0)
    return nb, nc


def test_bodies_and_colliders_spawned_inside_the_device_closed_loop():
    """`pg.bcol` and `ent2slot` were sized once at avn_pipeline_enable.  Two spawns while contacts are live — one appended, one whose
    collider is uploaded FIRST (every live row's collider slots are renumbered) — and the run stays equal to the oracle's, step by step."""
    from test_gpu_graph import compare_step
    bodies, colliders = dropped_boxes(seed=11, n=40)
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(32, substeps=4))
        upload(w, bodies, colliders); w.collider_materials_upload(friction=0.6, restitution=0.0); w.pipeline_enable()
        worlds.append(w)
    wo, wh = worlds
    s = 0
    for _ in range(25):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    assert wh.pipeline_stats().manifolds > 20
    state = [(bodies, colliders), (bodies, colliders)]
    for k, (pos, first) in enumerate((([1.5, 6.0, 1.5], False), ([2.5, 7.0, 2.0], True))):
        state = [_respawn(w, st[0], st[1], pos, [0.4, 0.3, 0.5], 5000 + k, first) for w, st in zip((wo, wh), state)]
        for _ in range(30):
            wo.step(); wh.step(); compare_step(s, wo, wh, check_rows=(s % 10 == 0)); s += 1
    # the spawned boxes have landed on the pile: they are part of the contact graph
    ids = wh.pipeline_handles()[1]
    assert len(ids) and wh.bodies_download()["position"][-1, 1] < 6.5


def test_shrinking_uploads_without_avn_despawn_are_refused_not_ignored():
    """Fewer bodies, or a collider with live contact rows missing from the upload, WITHOUT avn_despawn: AVN_ERR_STATE and an unchanged world
    (the library used to switch the closed loop off silently).  The way to remove bodies is avn_despawn (tests/test_gpu_despawn.py)."""
    bodies, colliders = dropped_boxes(seed=12, n=20)
    w = F.World(hip_lib(), F.default_config(32, substeps=4))
    upload(w, bodies, colliders); w.pipeline_enable()
    for _ in range(20):
        w.step()
    st = w.pipeline_stats()
    assert st.manifolds > 5
    with pytest.raises(F.AvnError):
        w.bodies_upload(**{k: np.asarray(v)[:-1] for k, v in bodies.items()})
    # the ground carries contact rows: dropping its collider orphans them
    with pytest.raises(F.AvnError):
        w.colliders_upload(**{k: np.asarray(v)[1:] for k, v in colliders.items()})
    w.step(); w.synchronize()
    assert w.pipeline_stats().manifolds > 5 and np.isfinite(w.bodies_download()["position"]).all()


def test_contacts_upload_to_a_dead_row_is_skipped_and_reported():
    """in the device closed loop liveness is a row flag; an upload to a free ContactId must not create a phantom pair."""
    bodies, colliders = dropped_boxes(seed=13, n=12)
    w = F.World(hip_lib(), F.default_config(32, substeps=4))
    upload(w, bodies, colliders); w.pipeline_enable()
    for _ in range(15):
        w.step()
    ids = np.unique(w.pipeline_handles()[1]).astype(np.uint32)
    rows = w.contacts_download(ids[:1])
    dead = np.array([w.pipeline_stats().pairs_added + 7], np.uint32)   # beyond every id handed out so far, inside the table's capacity
    with pytest.raises(F.AvnError, match="k_pack_contacts"):
        w.contacts_upload(dead, rows)
    z = w.contacts_download(dead)
    assert int(z["flags"][0]) == 0 and int(z["point_count"][0]) == 0
    w.contacts_upload(ids[:1], rows)          # a live row is fine, and the error word was cleared
    w.step(); w.synchronize()
