"""GPU: avn_diagnostics_get fills the fields of the reference's SolverDiagnostics (dynamics/solver/diagnostics.rs:13-37) and
CollisionDiagnostics (collision/diagnostics.rs:13-19) from events on the world's stream."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib

pytestmark = pytest.mark.gpu
SUBSTEP_FIELDS = ("warm_start_ms", "solve_constraints_ms", "integrate_positions_ms", "relax_velocities_ms")


def closed_loop_world(lib, use_graph):
    sc = scenes.box_stack(12, 10, 12)
    cfg = F.default_config(32, substeps=4)
    cfg.use_graph = use_graph
    w = F.World(lib, cfg)
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5, restitution=0.2)
    w.pipeline_enable()
    for _ in range(4):
        w.step()
    w.synchronize()
    return w


def test_direct_launches_fill_every_solver_diagnostics_field():
    w = closed_loop_world(hip_lib(), 0)
    d, t = w.diagnostics(), w.timers()
    assert d.per_system_valid == 1
    for f in SUBSTEP_FIELDS + ("prepare_constraints_ms", "update_velocity_increments_ms", "apply_restitution_ms", "finalize_ms", "store_impulses_ms",
                               "substeps_ms", "broad_phase_ms", "narrow_phase_ms"):
        assert getattr(d, f) > 0.0, f
    assert d.integrate_velocities_ms == 0.0 and d.swept_ccd_ms == 0.0   # fused into the warm-start launch / outside the path
    parts = sum(getattr(d, f) for f in SUBSTEP_FIELDS)
    assert 0.6 * d.substeps_ms <= parts <= 1.05 * d.substeps_ms, (parts, d.substeps_ms)
    assert abs(d.substeps_ms - t.substeps_ms) < 1e-6
    whole = d.broad_phase_ms + d.narrow_phase_ms + d.prepare_constraints_ms + d.update_velocity_increments_ms + d.substeps_ms + d.apply_restitution_ms + d.finalize_ms + d.store_impulses_ms
    assert 0.8 * t.step_ms <= whole <= 1.05 * t.step_ms, (whole, t.step_ms)
    st = w.pipeline_stats()
    assert d.contact_constraint_count == t.contact_constraint_count > 1000 and d.contact_count == st.active_pairs


def test_graph_replay_reports_the_loop_total_only():
    w = closed_loop_world(hip_lib(), 1)
    d = w.diagnostics()
    assert d.per_system_valid == 0 and all(getattr(d, f) == 0.0 for f in SUBSTEP_FIELDS)
    assert d.substeps_ms > 0.0 and d.narrow_phase_ms > 0.0 and d.finalize_ms > 0.0


def test_oracle_fills_the_same_struct():
    w = closed_loop_world(oracle_lib(), 0)
    d = w.diagnostics()
    assert d.per_system_valid == 1 and d.integrate_velocities_ms > 0.0 and d.solve_constraints_ms > 0.0 and d.narrow_phase_ms > 0.0 and d.contact_count > 1000
