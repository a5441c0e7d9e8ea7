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
from pipeline_scenes import dropped_boxes, stack_and_projectile, stack_chain_and_projectile
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


def despawn_link(worlds, sc_bodies, sc_colliders, joints, link):
    """Despawn body `link` of a chain together with the (one or two) joints that name it: avn_despawn(joints + body), then the three uploads."""
    gone_j = np.flatnonzero((joints["body1"] == link) | (joints["body2"] == link)).astype(np.uint32)
    n = len(sc_bodies["inv_mass"])
    mask = np.ones(n, bool); mask[link] = False
    new_index = np.cumsum(mask) - 1
    cmask = mask[np.asarray(sc_colliders["body"])]
    nb = subset(sc_bodies, mask)
    nc = {k: (np.asarray(v)[cmask] if isinstance(v, np.ndarray) and len(v) == len(cmask) else v) for k, v in sc_colliders.items()}
    nc["body"] = new_index[np.asarray(sc_colliders["body"])[cmask]].astype(np.int32)
    keep = np.ones(len(joints["body1"]), bool); keep[gone_j] = False
    nj = {k: (np.asarray(v)[keep] if isinstance(v, np.ndarray) and len(v) == len(keep) else v) for k, v in joints.items()}
    nj["body1"] = new_index[joints["body1"][keep]].astype(np.int32); nj["body2"] = new_index[joints["body2"][keep]].astype(np.int32)
    for w in worlds:
        state = w.bodies_download()
        w.despawn(bodies=[link], joints=gone_j)
        kw = dict(nb)
        for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
            kw[k] = state[k][mask].astype(np.float64)
        w.bodies_upload(**kw); w.colliders_upload(**nc); w.collider_materials_upload(friction=0.5)
        w.distance_joints_upload(**nj)
    return nb, nc, nj, len(gone_j)


@pytest.mark.parametrize("bits", [32, 64])
def test_despawn_a_jointed_body_with_sleeping_on(bits):
    """Round 5 (missing 6: avn_despawn refused a body that a joint names).  A chain of DistanceJoints draped over a stack, ONE island through
    add_joint; it falls asleep; a link in the middle of the chain is despawned with its two joints: remove_joint_from_graph per joint (constraints_removed,
    the sleeping island woken and its manifolds pushed back in the reference's order), then the body; the chain is two pieces from then on and the island
    splits when it next rests.  Every step against the oracle: colour lists with order, bodies, joints' solver data, island ids, body-list order, timers."""
    from helpers import compare_dicts
    sc, joints = stack_chain_and_projectile(height=400.0)   # (the projectile stays out of the way for the length of the test)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(bits, substeps=4))
        w.bodies_upload(**bodies); w.colliders_upload(**colliders); w.distance_joints_upload(**joints)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable(); w.sleeping_enable(time_to_sleep=0.3, linear_threshold=0.3, angular_threshold=0.6)
        worlds.append(w)
    wo, wh = worlds
    s = 0
    slept = False
    for _ in range(160):
        wo.step(); wh.step(); compare_step(s, wo, wh); compare_sleeping(s, wo, wh); s += 1
        if int(wh.sleeping_state()["sleeping"].sum()) >= sc.n - 2:   # stack + chain asleep (first at step 18; the f64 run flip-flops: asleep for a step at a time)
            slept = True
            break
    assert slept, "stack and chain must be asleep before the despawn"
    n_links = len(joints["body1"]) + 1
    link = int(joints["body1"][0]) + n_links // 2          # a link in the middle of the chain
    removed_before = wh.sleeping_stats().islands.n_sleeping_islands
    bodies, colliders, joints, n_gone = despawn_link((wo, wh), bodies, colliders, joints, link)
    assert n_gone == 2
    compare_step("after the despawn", wo, wh, check_rows=True); compare_sleeping("after the despawn", wo, wh)
    assert removed_before >= 1 and wh.sleeping_stats().islands.n_sleeping_islands == 0, "the island that lost two joints and a body wakes"
    for _ in range(120):
        wo.step(); wh.step(); compare_step(s, wo, wh); compare_sleeping(s, wo, wh)
        compare_dicts(wo.joints_download(), wh.joints_download(), f"step {s}: joints"); s += 1
    st = wh.sleeping_stats()
    assert st.islands.n_bodies == sc.n - 1 - 1 and st.islands.splits >= 1
