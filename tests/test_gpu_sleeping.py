"""GPU: persistent islands and sleeping ACTUATION in the device closed loop (avn_sleeping_enable: host island manager + device op
pipeline, world/sleeping.hpp) against the oracle (the reference's linked-list structures restated, oracle/avo_islands.hpp) -- after EVERY
step: colour lists with their order (SleepIslands pops in body-list x edge-list order: the order decides where swap_remove moves handles;
WakeIslands pushes in that order: it decides the colours), pipeline counters, bodies, island ids (slab keys), body-list order, Sleeping
flags, SleepTimers (bit patterns), the counters of the Sleeping set.  Reference: islands/mod.rs:513-1280, islands/sleeping.rs:164-540."""
import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from pipeline_scenes import dropped_boxes, stack_and_projectile
from test_gpu_graph import compare_step

pytestmark = pytest.mark.gpu

SLEEP_STATS = ("n_awake_bodies", "last_islands_slept", "last_islands_woken", "last_manifolds_popped", "last_manifolds_pushed")
ISLAND_STATS = ("n_islands", "n_sleeping_islands", "n_bodies", "n_sleeping_bodies", "merges", "splits", "split_candidate", "sleeping_pairs")


def compare_sleeping(s, wo, wh):
    so, sh = wo.sleeping_state(), wh.sleeping_state()
    for k in so:
        a, b = so[k], sh[k]
        if k == "sleep_timer":
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), f"step {s}: sleeping state {k} differs at {np.flatnonzero(a != b)[:8]}: oracle {so[k][a != b][:8]} device {sh[k][a != b][:8]}"
    to, th = wo.sleeping_stats(), wh.sleeping_stats()
    for f in SLEEP_STATS:
        assert getattr(to, f) == getattr(th, f), f"step {s}: sleeping stats.{f}: oracle {getattr(to, f)} device {getattr(th, f)}"
    for f in ISLAND_STATS:
        assert getattr(to.islands, f) == getattr(th.islands, f), f"step {s}: islands.{f}: oracle {getattr(to.islands, f)} device {getattr(th.islands, f)}"


def pair_of_worlds(bodies_kw, colliders_kw, bits=32, substeps=4, **sleep_kw):
    out = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**bodies_kw); w.colliders_upload(**colliders_kw)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable(); w.sleeping_enable(**sleep_kw)
        out.append(w)
    return out


@pytest.mark.parametrize("bits", [32, 64])
def test_stack_sleeps_and_a_dropped_box_wakes_it_260_steps(bits):
    """27 boxes settle, the island is split (constraints were removed while it settled), falls asleep (~90 manifolds popped), wakes itself when
    a still-active non-touching pair starts touching, sleeps again; a box dropped from 32 m lands at step ~146: its island merges with the
    sleeping one, WakeIslands pushes the manifolds back; the pile settles and sleeps again."""
    sc = stack_and_projectile(3, 3, 3, height=32.0)
    wo, wh = pair_of_worlds(sc.body_kwargs(), sc.collider_kwargs(), bits=bits)
    slept = woken_by_impact = 0
    for s in range(260):
        wo.step(); wh.step()
        compare_step(s, wo, wh, check_rows=(s % 40 == 39))
        compare_sleeping(s, wo, wh)
        st = wh.sleeping_stats()
        slept += st.last_islands_slept
        if s > 140 and st.n_awake_bodies == sc.n - 1:
            woken_by_impact = 1
    assert slept >= 3 and woken_by_impact and wh.sleeping_stats().islands.splits >= 3
    assert wh.sleeping_stats().last_host_ms < 50.0


def test_tumbling_pile_with_balls_sleeps_in_pieces():
    """60 tumbling boxes / balls: islands merge as bodies land on each other, pieces that come to rest split off and sleep while others still
    roll; per-body thresholds and one SleepingDisabled body."""
    bodies, colliders = dropped_boxes(seed=17, n=60)
    n = len(bodies["inv_mass"])
    rng = np.random.default_rng(3)
    lin = np.where(rng.random(n) < 0.3, 0.4, 0.15).astype(np.float32); ang = np.where(rng.random(n) < 0.3, 0.5, 0.15).astype(np.float32)
    dis = np.zeros(n, np.uint8); dis[7] = 1
    wo, wh = pair_of_worlds(bodies, colliders, time_to_sleep=0.3, body_linear_threshold=lin, body_angular_threshold=ang, body_sleeping_disabled=dis)
    slept = 0
    for s in range(240):
        wo.step(); wh.step()
        compare_step(s, wo, wh)
        compare_sleeping(s, wo, wh)
        slept += wh.sleeping_stats().last_islands_slept
    assert slept >= 2 and wh.sleeping_stats().islands.merges > 20


def test_wake_bodies_and_switching_sleeping_off():
    sc = stack_and_projectile(3, 3, 3, height=80.0)
    wo, wh = pair_of_worlds(sc.body_kwargs(), sc.collider_kwargs())
    for s in range(110):
        wo.step(); wh.step()
    compare_sleeping(110, wo, wh)
    assert wh.sleeping_state()["sleeping"].sum() >= 27
    for w in (wo, wh):
        w.wake_bodies([5])
    compare_step(111, wo, wh); compare_sleeping(111, wo, wh)
    assert wh.sleeping_state()["sleeping"].sum() == 0 and wh.pipeline_stats().manifolds > 50
    for s in range(112, 150):
        wo.step(); wh.step()
        compare_step(s, wo, wh); compare_sleeping(s, wo, wh)
    for w in (wo, wh):
        w.sleeping_enable(False)
        w.step()
    compare_step(151, wo, wh)


def test_large_stack_sleeps_and_wakes_in_thousands_of_ops():
    """2 197 boxes (13 x 13 x 13) with a SHORT time_to_sleep and generous thresholds: the lattice flip-flops -- the whole island
    falls asleep every few steps (5 000 - 8 500 manifolds popped at once, in body-list x edge-list order) and the status loop wakes it in the
    next step (as many pushed back): batches of the workgroup replay (k_pg_replay_wide) full of pops whose fillers are popped later in the same
    batch, whole colour lists emptied and refilled, cuts.  Every step against the oracle: colour lists with order, counters, bodies, island
    state."""
    sc = stack_and_projectile(13, 13, 13, height=12.0)
    wo, wh = pair_of_worlds(sc.body_kwargs(), sc.collider_kwargs(), bits=32, time_to_sleep=0.05, linear_threshold=3.0, angular_threshold=3.0)
    popped = events = 0
    for s in range(70):
        wo.step(); wh.step()
        compare_step(s, wo, wh, check_rows=(s % 35 == 34))
        compare_sleeping(s, wo, wh)
        st = wh.sleeping_stats()
        popped += st.last_manifolds_popped; events += 1 if st.last_manifolds_popped else 0
    assert popped > 40000 and events >= 8, (popped, events)


def test_a_world_entirely_asleep_steps_as_the_identity_and_can_be_woken():
    """Once every body sleeps nothing can change: the device closed loop returns from avn_step without a launch (world/sleeping.hpp), the oracle
    runs the whole step -- colour lists, counters, bodies, island state and timers must stay equal for 60 such steps; then WakeBody, and a spawn
    into the sleeping world (the upload ends the shortcut)."""
    sc = stack_and_projectile(3, 3, 3, height=3.6)
    wo, wh = pair_of_worlds(sc.body_kwargs(), sc.collider_kwargs(), time_to_sleep=0.3, linear_threshold=0.4, angular_threshold=0.8)
    s = 0
    all_asleep_since = None
    for _ in range(400):
        wo.step(); wh.step()
        compare_step(s, wo, wh); compare_sleeping(s, wo, wh); s += 1
        if wh.sleeping_state()["sleeping"].sum() == sc.n - 1:
            all_asleep_since = all_asleep_since if all_asleep_since is not None else s
            if s - all_asleep_since >= 60:
                break
        else:
            all_asleep_since = None
    assert all_asleep_since is not None and s - all_asleep_since >= 60, "the world must fall asleep as a whole and stay asleep"
    assert wh.timers().kernel_launches == 0, "a fully sleeping world costs no launch"
    for w in (wo, wh):
        w.wake_bodies([3])
    compare_step(s, wo, wh); compare_sleeping(s, wo, wh)
    for k in range(30):
        wo.step(); wh.step(); compare_step(s, wo, wh); compare_sleeping(s, wo, wh); s += 1
        if k == 0:
            assert wh.timers().kernel_launches > 0, "WakeBody ends the shortcut"
