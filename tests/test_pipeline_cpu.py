"""CPU: the closed-loop contact pipeline (device-side ContactGraph rows + host status processing) on the oracle —
broad phase -> narrow phase -> status changes -> ConstraintGraph -> solver.  Checks the bookkeeping invariants and
that a pile of boxes comes to rest on the ground instead of sinking or exploding."""
import numpy as np

from avian_amd.pipeline import ContactPipeline
from helpers import F, oracle_lib
from pipeline_scenes import dropped_boxes


def make(lib, bits, bodies, colliders, substeps=4):
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    w.bodies_upload(**bodies)
    w.colliders_upload(**colliders)
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.collider_materials_upload(friction=0.6, restitution=0.0)
    return w, ContactPipeline(w, lib)


def test_pile_comes_to_rest_and_bookkeeping_is_consistent():
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=3, n=40)
    w, pl = make(lib, 32, bodies, colliders)
    for s in range(150):
        pl.step()
        # every handle the ConstraintGraph holds belongs to a live, touching, constraint-generating pair
        off, handles = pl.graph.lists()
        assert len(set(handles.tolist())) == len(handles)
        if len(handles):
            c = w.contacts_download(handles.astype(np.uint32))
            assert np.all(c["point_count"] >= 1) and np.all(c["flags"] & F.CP_TOUCHING) and np.all(c["flags"] & F.CP_GENERATE_CONSTRAINTS)
            assert np.all(np.abs(np.linalg.norm(c["normal"], axis=1) - 1) < 1e-3)
    b = w.bodies_download()
    he = colliders["half_extents"]
    lowest = b["position"][1:, 1] - np.where(colliders["shape"][1:] == F.SHAPE_BALL, he[1:, 0], np.linalg.norm(he[1:], axis=1))
    assert np.isfinite(b["position"]).all() and lowest.min() > -0.7, "nothing tunnels through the ground slab"
    assert b["position"][1:, 1].max() < 8.0 and np.abs(b["linear_velocity"]).max() < 3.0, "the pile settles"
    assert pl.stats["pushes"] > 40 and pl.stats["pops"] > 0 and pl.stats["pairs_added"] > 40
    # warm starting is alive: persisting contacts carry impulses over
    off, handles = pl.graph.lists()
    c = w.contacts_download(handles.astype(np.uint32))
    assert float(c["warm_start_normal_impulse"].max()) > 0.0


def test_pairs_are_removed_when_aabbs_separate_and_ids_are_reused():
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=5, n=8, balls=False)
    bodies["linear_velocity"][1:] = [[6.0, 0.0, 0.0]] * 4 + [[-6.0, 0, 0]] * 4   # fly apart
    w, pl = make(lib, 64, bodies, colliders)
    ids_seen = set()
    for s in range(60):
        pl.step()
        ids_seen |= set(pl.pairs)
    assert pl.stats["pairs_removed"] > 0
    assert max(ids_seen) < pl.stats["pairs_added"], "freed ContactIds are reused lowest-first"


def test_library_pipeline_equals_the_python_driver():
    """avn_pipeline_enable (the library's own host bookkeeping, C++) gives the same colour lists and body state as the
    Python driver over the low-level calls — two independent implementations of the status-change loop."""
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=9, n=30)
    wa, pa = make(lib, 32, bodies, colliders)
    wb, _ = make(lib, 32, bodies, colliders)
    wb.pipeline_enable()
    for s in range(60):
        pa.step(); wb.step()
        offa, ha = pa.graph.lists()
        offb, hb = wb.pipeline_handles()
        assert np.array_equal(offa, offb) and np.array_equal(ha.astype(np.uint32), hb), f"step {s}"
        ba, bb = wa.bodies_download(), wb.bodies_download()
        for k in ba:
            assert np.array_equal(ba[k], bb[k]), f"step {s}: {k}"
    st = wb.pipeline_stats()
    assert st.pairs_added == pa.stats["pairs_added"] and st.manifolds_pushed == pa.stats["pushes"] and st.manifolds_popped == pa.stats["pops"]
    assert st.active_pairs == len(pa.active) and st.manifolds == len(ha)


def test_a_world_continues_on_a_fresh_world_when_its_contact_rows_move_with_it():
    """avn_contacts_upload (the reference holds no test of its own: its ContactGraph never leaves the process).  A pile is stepped,
    moved onto a fresh world — bodies, colliders in interval order, pair set, contact rows, colour lists — and both are stepped
    further under ONE host pipeline: every broad-phase pair list, every status-change list, the bodies and the rows stay
    bit-identical.  Without the rows (negative control) the continuation differs: the warm-start impulses and the touching flags
    are state."""
    import migration_helpers as M
    lib = oracle_lib()
    for bits in (32, 64):
        bodies, colliders = dropped_boxes(seed=11, n=30)
        a, b, pl, _ = M.run_migration(lib, bits, bodies, colliders, steps_before=45, steps_after=40)
        assert len(pl.pairs) > 20 and pl.graph.lists()[1].size > 10
        M.assert_same_world(a, b, pl)
    bodies, colliders = dropped_boxes(seed=11, n=30)
    a, b, pl, mirror = M.run_migration(lib, 32, bodies, colliders, steps_before=45, steps_after=5, rows=False)
    ba, bb = a.bodies_download(), b.bodies_download()
    assert mirror.differences > 0 or not np.array_equal(ba["linear_velocity"], bb["linear_velocity"])


def test_contacts_upload_round_trip_and_errors():
    import pytest
    import migration_helpers as M
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=2, n=12)
    w = M.new_world(lib, 64, bodies, colliders)
    pl = ContactPipeline(w, lib)
    for _ in range(30):
        pl.step()
    ids = np.array(sorted(pl.pairs), np.uint32)
    rows = w.contacts_download(ids)
    assert rows["point_count"].max() >= 1
    scrambled = {k: np.roll(v, 1, axis=0) for k, v in rows.items()}
    w.contacts_upload(ids, scrambled)
    got = w.contacts_download(ids)
    live = np.arange(4)[None, :] < scrambled["point_count"][:, None]
    for k, v in scrambled.items():
        if k in ("flags", "point_count"):
            assert np.array_equal(got[k], v), k
        elif v.ndim == 1 or k == "normal":                  # per-manifold fields: zero when the row has no manifold
            m = scrambled["point_count"] > 0
            assert np.array_equal(got[k][m], v[m]), k
        else:                                             # per-point fields: points beyond point_count read as zero
            assert np.array_equal(got[k][live], v[live]), k
            assert not np.any(got[k][~live]), k
    w.contacts_upload(ids, rows)
    with pytest.raises(F.AvnError):
        w.contacts_upload(np.array([10_000], np.uint32), {k: v[:1] for k, v in rows.items()})
    bad = {k: v[:1].copy() for k, v in rows.items()}; bad["point_count"][0] = 5
    with pytest.raises(F.AvnError):
        w.contacts_upload(ids[:1], bad)


def test_new_pair_ids_and_status_changes_describe_the_contact_graph():
    """avn_pipeline_new_pair_ids_get + avn_contact_changes_get are enough to mirror the loop's ContactGraph on a host (what the Rust layer builds
    CollisionStart / CollisionEnd / CollidingEntities from): replaying them reproduces the set of touching pairs the colour lists hold."""
    bodies, colliders = dropped_boxes(seed=5, n=60)
    w, _ = make(oracle_lib(), 32, bodies, colliders, 4)
    w.pipeline_enable()
    pair_of, touching = {}, set()
    for s in range(70):
        w.step()
        ids, pairs = w.pipeline_new_pair_ids(), w.pairs_get()
        assert len(ids) == len(pairs)
        for cid, pr in zip(ids, pairs):
            assert int(cid) not in pair_of, "an id handed out while still live"
            pair_of[int(cid)] = (int(pr["collider1"]), int(pr["collider2"]))
        ch = w.contact_changes_get()
        assert np.all(np.diff(ch["contact_id"].astype(np.int64)) > 0), "ascending ContactId"
        for c in ch:
            cid, fl = int(c["contact_id"]), int(c["flags"])
            assert cid in pair_of
            if fl & F.CP_STARTED_TOUCHING: touching.add(cid)
            if fl & F.CP_STOPPED_TOUCHING: touching.discard(cid)
            if fl & F.CP_DISJOINT_AABB: touching.discard(cid); del pair_of[cid]
        _, handles = w.pipeline_handles()
        assert set(int(h) for h in handles) <= touching, f"step {s}: a constraint handle of a pair the events do not show as touching"
    assert len(touching) > 50
