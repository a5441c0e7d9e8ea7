"""GPU: BASELINE.json's configurations in the CLOSED LOOP (device broad phase -> narrow phase -> ContactGraph / ConstraintGraph
bookkeeping -> solver, avn_pipeline_enable(1)) against the oracle, tolerance 0, in the regimes the bench reports:

* cfg2 carried to the steady window the bench line's `closed_loop` leg is measured in (steps 100..119): ids reused through the IdPool, an
  overflow colour of a few hundred manifolds, ~21 colours;
* cfg3 with the chains swinging into / lowered onto the stack: joints + contacts + `collision_disabled` pairs + status changes together
  (reference order: relax -> joints, xpbd/plugin.rs:30-40,145-189; status loop narrow_phase/system_param.rs:141-389);
* cfg1 exactly as SURVEY.md §8(d) writes it: 1 000 falling cuboids, spacing 1.5, lowest layer at y = 2, 1 substep, 300 steps.

After EVERY step: colour lists with their order, pipeline counters, all bodies (and joints); contact rows on a sample."""
import os

import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, compare_dicts, hip_lib, oracle_lib
from test_gpu_graph import compare_step

pytestmark = pytest.mark.gpu


def closed_loop_pair(sc, bits=32, substeps=4, joints=None, friction=0.5):
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        if joints is not None:
            w.distance_joints_upload(**joints)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction)
        w.pipeline_enable()
        worlds.append(w)
    return worlds


def threads():
    return str(max(1, min(64, os.cpu_count() or 1)))


def test_cfg2_closed_loop_to_the_steady_window_steps_100_119(monkeypatch):
    """BASELINE.json config 2 (100 000 cuboids, 4 substeps), device bookkeeping, 120 steps: the collapse of the lattice (~2e5 status
    changes per step, an overflow colour 1.8e5 strong and hundreds of levels deep), the decay, and the steady regime bench.py times."""
    monkeypatch.setenv("AVO_THREADS", threads())
    wo, wh = closed_loop_pair(scenes.box_stack(50, 40, 50))
    reused = False
    for s in range(120):
        wo.step(); wh.step()
        compare_step(s, wo, wh)
        st = wh.pipeline_stats()
        reused = reused or (st.pairs_removed > 1000 and st.pairs_added > 1_244_836 + 1000)
        if s in (21, 100, 110, 119):
            ids = np.unique(wh.pipeline_handles()[1])[::97]
            ro, rh = wo.contacts_download(ids), wh.contacts_download(ids)
            for k in ro:
                assert np.array_equal(ro[k], rh[k]), f"step {s}: contact rows {k} differ"
    st = wh.pipeline_stats()
    assert reused, "the run must free ContactIds and hand them out again (IdPool, lowest id first)"
    assert 150_000 < st.manifolds < 300_000 and 10 < st.last_overflow_manifolds < 5000 and st.last_status_changes > 5000
    assert st.last_host_ms < 5.0


def test_cfg3_closed_loop_chains_swing_into_the_stack(monkeypatch):
    """cfg3's bodies (50k cuboids + 100 chains x 100 links, 9 900 DistanceJoints, first links kinematic) with the chains hitting the
    stack: 24 closed-loop steps, bodies + joints + colour lists every step."""
    monkeypatch.setenv("AVO_THREADS", threads())
    sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100, swing=True)
    joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
    wo, wh = closed_loop_pair(sc, joints=joints)
    n0 = 50 * 20 * 50 + 1
    for s in range(24):
        wo.step(); wh.step()
        compare_step(s, wo, wh, check_rows=(s in (5, 23)))
        compare_dicts(wo.joints_download(), wh.joints_download(), f"cfg3 closed loop step {s}: joints")
    st = wh.pipeline_stats()
    b = wh.bodies_download()
    assert st.manifolds > 50_000
    assert float(np.abs(wh.joints_download()["total_lagrange"]).max()) > 0.0
    assert np.isfinite(b["position"]).all()
    # the side chains started 0.19 m off the +x face moving at up to 4 m/s: after 24 steps their lower links have been stopped by it
    side_low = b["position"][n0 + 60:n0 + 100, 0]
    assert side_low.min() > 24.9, "links must not tunnel into the stack"
    assert side_low.min() < 25.2, "the lower links must have reached the face"


def test_cfg1_thousand_falling_cuboids_300_steps_closed_loop():
    """cfg1: 10 x 10 x 10 unit cuboids, spacing 1.5, lowest layer centred at y = 2, ground cuboid(200, 1, 200) centred
    y = -0.5, 1 substep, 300 steps: free fall, landing, the layers piling onto each other."""
    sc = scenes.falling_grid(10, 1.5, 2.0)
    assert sc.n == 1001
    wo, wh = closed_loop_pair(sc, substeps=1)
    peak = 0
    for s in range(300):
        wo.step(); wh.step()
        compare_step(s, wo, wh, check_rows=(s % 50 == 49))
        peak = max(peak, wh.pipeline_stats().manifolds)
    st = wh.pipeline_stats()
    assert peak >= 1000 and st.manifolds_pushed > 1000, "the boxes must land (100 ground contacts + 900 box-box contacts at least)"
    y = wh.bodies_download()["position"][1:, 1]
    assert y.min() > 0.45 and y.max() < 16.0
