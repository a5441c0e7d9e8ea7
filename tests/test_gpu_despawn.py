"""GPU: despawn INSIDE the device closed loop (avn_despawn, world/despawn.hpp) against the oracle (oracle/avo_world.hpp: World::despawn, itself
checked against an independent Python model of remove_collider in tests/test_despawn_cpu.py), tolerance 0.  After the despawn and after EVERY
step: colour lists with their order (the pops of a removed collider's pairs happen in the ContactGraph's edge-list order and swap_remove makes the
lists remember it), pipeline counters, bodies (renumbered), contact rows; the freed ContactIds must be handed out again lowest first.
Reference: collision/narrow_phase/mod.rs:399-457, contact_types/contact_graph.rs:641-700, data_structures/stable_graph.rs:251-315,
data_structures/id_pool.rs:31-40, dynamics/solver/islands/mod.rs:1336-1400."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
from pipeline_scenes import dropped_boxes, stack_and_projectile
from test_despawn_cpu import subset
from test_gpu_graph import compare_step
from test_gpu_sleeping import compare_sleeping

pytestmark = pytest.mark.gpu


def make_pair(bodies, colliders, bits=32, sleeping=None, friction=0.5):
    out = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(bits, substeps=4))
        w.bodies_upload(**bodies); w.colliders_upload(**colliders)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction)
        w.pipeline_enable()
        if sleeping is not None:
            w.sleeping_enable(**sleeping)
        out.append(w)
    return out


def despawn_both(worlds, bodies, colliders, gone, friction=0.5, single_colliders=()):
    """avn_despawn on both worlds + the uploads the header asks for; returns the compacted (bodies, colliders)."""
    n = len(bodies["inv_mass"])
    mask = np.ones(n, bool); mask[np.asarray(gone, int)] = False
    cmask = mask[np.asarray(colliders["body"])] & ~np.isin(colliders["entity_index"], np.asarray(single_colliders, np.uint32))
    new_index = np.cumsum(mask) - 1
    nb = subset(bodies, mask)
    nc = {k: (np.asarray(v)[cmask] if isinstance(v, np.ndarray) and len(v) == len(cmask) else v) for k, v in colliders.items()}
    nc["body"] = new_index[np.asarray(colliders["body"])[cmask]].astype(np.int32)
    for w in worlds:
        state = w.bodies_download()
        w.despawn(bodies=gone, collider_entities=single_colliders)
        kw = dict(nb)
        for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
            kw[k] = state[k][mask].astype(np.float64)
        w.bodies_upload(**kw); w.colliders_upload(**nc); w.collider_materials_upload(friction=friction)
    return nb, nc


@pytest.mark.parametrize("bits", [32, 64])
def test_despawn_ten_percent_of_a_settled_pile_then_forty_steps(bits):
    """80 tumbled boxes / balls settle for 45 steps; 10 % of them are despawned in a scrambled order (pops in edge-list order: a box in the
    middle of the pile has a dozen touching pairs in several colours); 40 steps; again 10 %; 40 more steps -- ids are reused on the way."""
    bodies, colliders = dropped_boxes(seed=41, n=80)
    wo, wh = make_pair(bodies, colliders, bits=bits)
    rng = np.random.default_rng(5)
    s = 0
    for _ in range(45):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    added_before = wh.pipeline_stats().pairs_added
    for rnd in range(2):
        n = len(bodies["inv_mass"])
        gone = rng.permutation(np.arange(1, n))[:max(3, n // 10)].astype(np.uint32)
        pops_before = wh.pipeline_stats().manifolds_popped
        bodies, colliders = despawn_both((wo, wh), bodies, colliders, gone)
        compare_step(f"after despawn {rnd}", wo, wh, check_rows=True)
        assert wh.pipeline_stats().manifolds_popped > pops_before, "the despawned boxes were touching their neighbours"
        for _ in range(40):
            wo.step(); wh.step(); compare_step(s, wo, wh, check_rows=(s % 10 == 0)); s += 1
    st = wh.pipeline_stats()
    assert st.pairs_added > added_before, "pairs formed after the despawn"
    ids = np.unique(wh.pipeline_handles()[1])
    assert len(ids) and ids.max() < st.pairs_added - st.pairs_removed + 64, "freed ContactIds are reused (lowest first), the table does not just grow"


def test_despawn_a_single_collider_and_the_ground():
    """A collider despawned on its own (its body stays: it falls through the pile from then on), then the static ground with every contact it
    carries -- the boxes resting on it fall."""
    bodies, colliders = dropped_boxes(seed=42, n=30)
    wo, wh = make_pair(bodies, colliders)
    s = 0
    for _ in range(40):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    ent = int(colliders["entity_index"][7])
    bodies, colliders = despawn_both((wo, wh), bodies, colliders, [], single_colliders=[ent])
    compare_step("single collider", wo, wh, check_rows=True)
    for _ in range(15):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    bodies, colliders = despawn_both((wo, wh), bodies, colliders, [0])
    compare_step("ground", wo, wh, check_rows=True)
    y0 = wh.bodies_download()["position"][:, 1].copy()
    for _ in range(30):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    assert (wh.bodies_download()["position"][:, 1] < y0 - 0.5).all(), "without the ground everything falls"


def test_despawn_in_a_100k_stack_mid_collapse():
    """cfg2's bodies, 12 steps into the collapse (the overflow colour is 1e5 strong): 5 % of the boxes leave at once -- tens of thousands of pops
    in one op batch, in edge-list order across several colours -- then 6 more steps."""
    import os
    os.environ["AVO_THREADS"] = str(max(1, min(64, os.cpu_count() or 1)))
    sc = scenes.box_stack(50, 40, 50)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    wo, wh = make_pair(bodies, colliders)
    s = 0
    for _ in range(12):
        wo.step(); wh.step(); s += 1
    compare_step(s, wo, wh)
    rng = np.random.default_rng(9)
    gone = rng.permutation(np.arange(1, sc.n))[:5000].astype(np.uint32)
    bodies, colliders = despawn_both((wo, wh), bodies, colliders, gone)
    compare_step("after the despawn", wo, wh)
    for _ in range(6):
        wo.step(); wh.step(); compare_step(s, wo, wh); s += 1
    assert wh.pipeline_stats().pairs_removed > 50_000


@pytest.mark.parametrize("bits", [32, 64])
def test_despawn_with_sleeping_on(bits):
    """With avn_sleeping_enable: a despawned body leaves its island (BodyIslandNode::on_remove), its touching pairs are unlinked
    (constraints_removed: the island may split later), and the queued WakeIslands wakes a sleeping island one of whose boxes was taken away."""
    sc = stack_and_projectile(3, 3, 3, height=60.0)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    slp = dict(time_to_sleep=0.3, linear_threshold=0.3, angular_threshold=0.6)
    wo, wh = make_pair(bodies, colliders, bits=bits, sleeping=slp)
    s = 0
    asleep = False
    for _ in range(120):
        wo.step(); wh.step(); compare_step(s, wo, wh); compare_sleeping(s, wo, wh); s += 1
        if wh.sleeping_state()["sleeping"].sum() >= 27:   # (the stack falls asleep for the first time after ~19 steps)
            asleep = True
            break
    assert asleep, "the stack must be asleep before the despawn"
    # a box from the bottom layer's corner and one from the middle of the stack
    bodies, colliders = despawn_both((wo, wh), bodies, colliders, np.array([14, 1], np.uint32))
    compare_step("after the despawn", wo, wh, check_rows=True); compare_sleeping("after the despawn", wo, wh)
    assert wh.sleeping_stats().islands.n_sleeping_islands == 0, "WakeIslands: the island a body was taken from wakes"
    for _ in range(80):
        wo.step(); wh.step(); compare_step(s, wo, wh); compare_sleeping(s, wo, wh); s += 1
    st = wh.sleeping_stats()
    assert st.islands.n_bodies == sc.n - 1 - 2
