"""GPU: the closed-loop contact pipeline — device narrow phase (k_narrow_phase over the contact table), host status
processing, manifolds gathered by handle, impulses scattered back — against the CPU oracle driven by the SAME host code,
bit for bit, every step: status changes, table rows, colour lists, body state."""
import numpy as np
import pytest

from avian_amd import scenes
from avian_amd.pipeline import ContactPipeline
from helpers import F, hip_lib, oracle_lib
from pipeline_scenes import dropped_boxes
from test_pipeline_cpu import make

pytestmark = pytest.mark.gpu


def run_both(bits, bodies, colliders, steps, substeps=4):
    wo, po = make(oracle_lib(), bits, bodies, colliders, substeps)
    wh, ph = make(hip_lib(), bits, bodies, colliders, substeps)
    for s in range(steps):
        for w, p in ((wo, po), (wh, ph)):
            w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
            p.add_new_pairs(w.pairs_get())
            w.run_system("NARROW_PHASE")
        co, ch = wo.contact_changes_get(), wh.contact_changes_get()
        assert np.array_equal(co, ch), f"step {s}: contact status changes differ"
        ids = np.asarray(sorted(po.pairs), np.uint32)
        assert sorted(ph.pairs) == ids.tolist()
        if len(ids):
            ro, rh = wo.contacts_download(ids), wh.contacts_download(ids)
            for k in ro:
                assert np.array_equal(ro[k], rh[k]), f"step {s}: contact table {k} differs"
        po.process_status_changes(); ph.process_status_changes()
        oo, oh = po.graph.lists(), ph.graph.lists()
        assert np.array_equal(oo[0], oh[0]) and np.array_equal(oo[1], oh[1]), f"step {s}: colour lists differ"
        wo.run_system("SOLVER"); wh.run_system("SOLVER")
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo:
            assert np.array_equal(bo[k], bh[k]), f"step {s}: bodies.{k} differs"
    return wo, wh, po, ph


@pytest.mark.parametrize("bits", [32, 64])
def test_dropped_pile_matches_oracle_every_step(bits):
    bodies, colliders = dropped_boxes(seed=11, n=64)
    wo, wh, po, ph = run_both(bits, bodies, colliders, steps=40)
    assert ph.stats == po.stats and ph.stats["pushes"] > 60 and ph.stats["pops"] > 0
    ids = np.asarray(sorted(ph.pairs), np.uint32)
    c = wh.contacts_download(ids)
    assert float(c["warm_start_normal_impulse"].max()) > 0.0 and int(c["point_count"].max()) == 4


def test_separating_bodies_remove_pairs_like_the_oracle():
    bodies, colliders = dropped_boxes(seed=5, n=8, balls=False)
    bodies["linear_velocity"][1:] = [[6.0, 0.0, 0.0]] * 4 + [[-6.0, 0, 0]] * 4
    wo, wh, po, ph = run_both(32, bodies, colliders, steps=50)
    assert ph.stats["pairs_removed"] == po.stats["pairs_removed"] > 0


def test_box_stack_1000_closed_loop_settles():
    """A 10 x 10 x 10 stack driven entirely on the device (no manifold upload): stays a stack."""
    sc = scenes.box_stack(10, 10, 10)
    w = F.World(hip_lib(), F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.collider_materials_upload(friction=0.5)
    pl = ContactPipeline(w, hip_lib())
    for _ in range(30):
        pl.step()
    b = w.bodies_download()
    assert np.isfinite(b["position"]).all()
    assert float(np.abs(b["position"][1:] - sc.position[1:]).max()) < 0.2 and float(np.abs(b["linear_velocity"]).max()) < 1.0
    assert w.n_manifolds > 2000


@pytest.mark.parametrize("bits", [32, 64])
def test_library_pipeline_hip_matches_oracle_and_python_driver(bits):
    """avn_pipeline_enable + avn_step on the HIP product == the oracle's own pipeline == the Python driver, bit for bit."""
    bodies, colliders = dropped_boxes(seed=21, n=80)
    wo, _ = make(oracle_lib(), bits, bodies, colliders); wo.pipeline_enable()
    wh, _ = make(hip_lib(), bits, bodies, colliders); wh.pipeline_enable()
    wp, pp = make(hip_lib(), bits, bodies, colliders)
    for s in range(50):
        wo.step(); wh.step(); pp.step()
        oo, oh = wo.pipeline_handles(), wh.pipeline_handles()
        assert np.array_equal(oo[0], oh[0]) and np.array_equal(oo[1], oh[1]), f"step {s}: colour lists"
        op = pp.graph.lists()
        assert np.array_equal(op[0], oh[0]) and np.array_equal(op[1].astype(np.uint32), oh[1])
        bo, bh, bp = wo.bodies_download(), wh.bodies_download(), wp.bodies_download()
        for k in bo:
            assert np.array_equal(bo[k], bh[k]) and np.array_equal(bp[k], bh[k]), f"step {s}: bodies.{k}"
    so, sh = wo.pipeline_stats(), wh.pipeline_stats()
    for f in ("pairs_added", "pairs_removed", "manifolds_pushed", "manifolds_popped", "active_pairs", "manifolds", "last_status_changes"):
        assert getattr(so, f) == getattr(sh, f), f
    assert sh.manifolds_pushed > 80 and sh.manifolds_popped > 0


def test_overflow_colour_per_level_launches_match_oracle(monkeypatch):
    """A pile dense enough to overflow the 23 colours, with the per-level launch path forced (threshold 0) — the same
    bits as the oracle's serial overflow loop."""
    monkeypatch.setenv("AVN_OVERFLOW_LEVEL_THRESHOLD", "0")
    sc = scenes.box_stack(7, 7, 7)
    worlds = []
    from helpers import hip_measure_lib   # (AVN_OVERFLOW_LEVEL_THRESHOLD is a test hook of the `make measure` build)
    for lib in (oracle_lib(), hip_measure_lib()):
        w = F.World(lib, F.default_config(32, substeps=4))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable(host_bookkeeping=True)   # (the device bookkeeping solves colour 23 with k_overflow_flow instead: tests/test_gpu_graph.py)
        worlds.append(w)
    wo, wh = worlds
    seen_overflow = 0
    for s in range(8):
        wo.step(); wh.step()
        seen_overflow = max(seen_overflow, wh.pipeline_stats().last_overflow_manifolds)
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo:
            assert np.array_equal(bo[k], bh[k]), f"step {s}: bodies.{k}"
    assert seen_overflow > 50, "the scene must actually use colour 23"


@pytest.mark.parametrize("bits", [32, 64])
def test_contact_rows_move_to_a_fresh_world_and_it_continues_bit_identically(bits):
    """avn_contacts_upload on the device: the pile of tests/test_pipeline_cpu.py is moved onto a fresh HIP world after 45 steps and
    both continue under one host pipeline for 40 more — pairs, status changes, bodies, rows and interval order stay equal; the
    migrated HIP world also equals the oracle's migrated world (same procedure there), so the upload kernel writes what the
    restatement writes."""
    import migration_helpers as M
    bodies, colliders = dropped_boxes(seed=11, n=30)
    ah, bh, plh, _ = M.run_migration(hip_lib(), bits, bodies, colliders, steps_before=45, steps_after=40)
    assert len(plh.pairs) > 20 and plh.graph.lists()[1].size > 10
    M.assert_same_world(ah, bh, plh)
    ao, bo, plo, _ = M.run_migration(oracle_lib(), bits, bodies, colliders, steps_before=45, steps_after=40)
    assert sorted(plo.pairs) == sorted(plh.pairs)
    M.assert_same_world(bo, bh, plh)


def test_contacts_upload_hip_round_trip_and_errors():
    import migration_helpers as M
    bodies, colliders = dropped_boxes(seed=2, n=12)
    w = M.new_world(hip_lib(), 32, bodies, colliders)
    pl = ContactPipeline(w, hip_lib())
    for _ in range(30):
        pl.step()
    ids = np.array(sorted(pl.pairs), np.uint32)
    rows = w.contacts_download(ids)
    scrambled = {k: np.roll(v, 1, axis=0) for k, v in rows.items()}
    w.contacts_upload(ids, scrambled)
    got = w.contacts_download(ids)
    live = np.arange(4)[None, :] < scrambled["point_count"][:, None]
    assert np.array_equal(got["flags"], scrambled["flags"]) and np.array_equal(got["point_count"], scrambled["point_count"])
    for k in ("anchor1", "anchor2", "penetration", "warm_start_normal_impulse", "warm_start_tangent_impulse", "normal_impulse", "feature_id1", "feature_id2"):
        assert np.array_equal(got[k][live], scrambled[k][live]) and not np.any(got[k][~live]), k
    with pytest.raises(F.AvnError):
        w.contacts_upload(np.array([1 << 20], np.uint32), {k: v[:1] for k, v in rows.items()})
