"""CPU: persistent islands and sleeping ACTUATION in the oracle's closed loop (avn_sleeping_enable; oracle/avo_islands.hpp, the reference's
islands/mod.rs + islands/sleeping.rs restated): physics-level checks no parity test can give -- a settled stack falls asleep, its bodies
stop being integrated and its manifolds leave the ConstraintGraph, a box dropped on it wakes it, the island structures stay consistent."""
import numpy as np

from helpers import F, oracle_lib
from pipeline_scenes import stack_and_projectile


def make(lib, bits=32, height=32.0):
    sc = stack_and_projectile(3, 3, 3, height=height)
    w = F.World(lib, F.default_config(bits, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable(); w.sleeping_enable()
    return sc, w


def test_stack_sleeps_projectile_wakes_it():
    sc, w = make(oracle_lib())
    n = sc.n
    asleep_steps, woke_after_impact, slept_positions = 0, False, None
    impact_step = None
    for s in range(200):
        before = w.bodies_download()["position"].copy()
        sleeping_before = w.sleeping_state()["sleeping"].copy()
        w.step()
        st, ps, state = w.sleeping_stats(), w.pipeline_stats(), w.sleeping_state()
        pos = w.bodies_download()["position"]
        # a body that slept through the whole step did not move (no SolverBody: not integrated, not solved)
        still = (sleeping_before == 1) & (state["sleeping"] == 1)
        assert np.array_equal(pos[still], before[still]), f"step {s}: a sleeping body moved"
        # islands: ids of bodies with a node are live, every list is walkable (next pointers form chains that cover each island once)
        isl, nxt = state["island"], state["next_in_island"]
        assert isl[0] == 0xFFFFFFFF and (isl[1:] != 0xFFFFFFFF).all()
        heads = set(range(1, n)) - set(int(x) for x in nxt if x != 0xFFFFFFFF)
        seen = 0
        for h in heads:
            b = h
            while b != 0xFFFFFFFF:
                assert isl[b] == isl[h]; seen += 1; b = int(nxt[b])
        assert seen == n - 1 and len(heads) == st.islands.n_islands
        if st.islands.n_sleeping_islands and st.n_awake_bodies == 1:
            asleep_steps += 1
            assert ps.manifolds == 0, "a sleeping island's manifolds are out of the ConstraintGraph"
        if impact_step is None and pos[-1, 1] < 4.0:
            impact_step = s
        if impact_step is not None and s > impact_step and st.n_awake_bodies == n - 1:
            woke_after_impact = True
    assert asleep_steps > 20, "the stack must sleep while the projectile falls"
    assert impact_step is not None and woke_after_impact, "the landing box must wake the stack (add_contact -> WakeIslands)"
    assert w.sleeping_stats().islands.splits >= 1 and w.sleeping_stats().islands.merges >= 27


def test_sleeping_disabled_body_keeps_its_island_awake_and_switching_off_wakes_everything():
    sc = stack_and_projectile(2, 2, 2, height=3.0)
    dis = np.zeros(sc.n, np.uint8); dis[3] = 1
    w = F.World(oracle_lib(), F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.pipeline_enable(); w.sleeping_enable(body_sleeping_disabled=dis)
    for _ in range(150):
        w.step()
        assert w.sleeping_state()["sleeping"].sum() == 0 or w.sleeping_state()["island"][3] not in set(w.sleeping_state()["island"][w.sleeping_state()["sleeping"] == 1])
    assert w.sleeping_state()["sleep_timer"][3] == 0.0
    w2 = make(oracle_lib(), height=60.0)[1]
    for _ in range(120):
        w2.step()
    assert w2.sleeping_state()["sleeping"].sum() > 0
    w2.sleeping_enable(False)
    w2.step()
    assert w2.pipeline_stats().manifolds > 20
