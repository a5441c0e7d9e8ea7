"""The CPU baseline's threads (oracle/avo_parallel.hpp; AVO_THREADS at world creation) restate the reference's own
parallelism — crate::utils::par_for_each over the constraints of a graph colour (src/utils.rs:57-87, solver/plugin.rs:476,564,
662), over colours in prepare_contact_constraints (plugin.rs:387-388), Query::par_iter_mut over bodies — and must not change
a single bit: every colour's constraints touch disjoint bodies."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, color_and_upload, compare_dicts, oracle_lib, random_joints, random_world


def world(threads, monkeypatch, bits=32, **kw):
    monkeypatch.setenv("AVO_THREADS", str(threads))
    w = F.World(oracle_lib(), F.default_config(bits, **kw))
    monkeypatch.delenv("AVO_THREADS")
    return w


@pytest.mark.parametrize("bits", [32, 64])
def test_threaded_oracle_is_bit_identical_to_serial(bits, monkeypatch):
    wd = random_world(seed=21, n_bodies=1500, n_manifolds=6000, n_joints=0, hub_degree=40)
    wd["joints_generic"] = random_joints(np.random.default_rng(3), 1500, 300)
    ws = [world(t, monkeypatch, bits, substeps=3) for t in (1, 4, 7)]
    for w in ws:
        color_and_upload(w, oracle_lib(), wd)
    for s in range(3):
        for w in ws:
            w.step()
        for w in ws[1:]:
            compare_dicts(ws[0].solver_bodies_download(), w.solver_bodies_download(), f"step {s}: solver bodies")
            compare_dicts(ws[0].constraints_download(), w.constraints_download(), f"step {s}: constraints")
            compare_dicts(ws[0].bodies_download(), w.bodies_download(), f"step {s}: bodies")
            compare_dicts(ws[0].impulses_download(), w.impulses_download(), f"step {s}: impulses")
            compare_dicts(ws[0].joints_download(), w.joints_download(), f"step {s}: joints")
            assert ws[0].timers().contact_constraint_count == w.timers().contact_constraint_count > 0


def test_threaded_oracle_closed_loop(monkeypatch):
    sc = scenes.many_pyramids(5, 2, 2)
    ws = []
    for t in (1, 5):
        w = world(t, monkeypatch, substeps=4)
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        ws.append(w)
    for s in range(12):
        for w in ws:
            w.step()
    a, b = ws[0].bodies_download(), ws[1].bodies_download()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert ws[0].pipeline_stats().manifolds == ws[1].pipeline_stats().manifolds > 0
