"""CPU (oracle backend): collision hooks -- CollisionHooks::filter_pairs / modify_contacts (reference collision/hooks.rs:137-231, called from
broad_phase.rs:431-439 and narrow_phase/system_param.rs:770-778) answered by host callbacks (avn_collision_hooks_set) inside the closed loop.  What the reference's
semantics imply is checked on small scenes: hooks that change nothing leave the world bit-identical; a rejected pair never enters the ContactGraph and is asked about
again every step; `false` from modify_contacts keeps a pair from touching; tangent_velocity set in the hook drives a body like a conveyor belt."""
import numpy as np
import pytest

from helpers import F, oracle_lib
from hook_helpers import Hooks, assert_same_hooked_step, hooked_world
from pipeline_scenes import dropped_boxes


def _two_boxes_on_ground(gap=0.02):
    """static slab (entity 10), box A resting on it (11), box B resting on A (12)."""
    pos = np.array([[0, -0.5, 0], [0, 0.5 + gap, 0], [0, 1.5 + 2 * gap, 0]], float)
    he = np.array([[20, 0.5, 20], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]], float)
    m = 3
    rot = np.tile([0.0, 0, 0, 1], (m, 1))
    rb = np.array([F.RB_STATIC, 0, 0], np.uint8)
    inv_mass = np.array([0.0, 1.0, 1.0])
    ii = np.zeros((m, 6)); ii[1:] = [6, 0, 0, 6, 0, 6]
    bodies = dict(position=pos, rotation=rot, linear_velocity=np.zeros((m, 3)), angular_velocity=np.zeros((m, 3)), inv_mass=inv_mass, inv_inertia_local=ii, rb_type=rb)
    colliders = dict(entity_index=np.array([10, 11, 12], np.uint32), body=np.arange(m, dtype=np.int32), shape=np.zeros(m, np.uint8), half_extents=he)
    return bodies, colliders


@pytest.mark.parametrize("bits,seed", [(32, 1), (64, 2)])
def test_hooks_that_change_nothing_leave_the_world_bit_identical(bits, seed):
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=seed, n=48)
    rng = np.random.default_rng(seed)
    flagged = rng.random(len(colliders["shape"])) < 0.4
    flagged[0] = seed % 2 == 0
    hooks = Hooks(identity=True)
    plain = hooked_world(lib, bits, bodies, colliders, np.zeros_like(flagged), None)
    hooked = hooked_world(lib, bits, bodies, colliders, flagged, hooks)
    plain.pipeline_enable(); hooked.pipeline_enable()
    for step in range(30):
        plain.step(); hooked.step()
        assert_same_hooked_step(plain, hooked, step, compare_flags=False)
        st = hooked.collision_hook_stats()
        assert st.last_filter_rejected == 0 and st.last_modify_rejected == 0
    assert len(hooks.filter_log) > 20 and len(hooks.modify_log) > 100, "the hooks must actually have been asked"
    # every pair the filter was asked about names a flagged collider; every record shown to modify_contacts belongs to a pair created with the flag
    fl = set(int(e) for e in np.asarray(colliders["entity_index"])[flagged])
    assert all(c1 in fl or c2 in fl for _, c1, c2 in hooks.filter_log)


def test_a_rejected_pair_never_enters_the_graph_and_is_asked_about_every_step():
    lib = oracle_lib()
    bodies, colliders = _two_boxes_on_ground()
    asked = []

    def flt(pairs, keep):
        for i in range(len(pairs)):
            asked.append((int(pairs["collider1"][i]), int(pairs["collider2"][i])))
            if {int(pairs["collider1"][i]), int(pairs["collider2"][i])} == {11, 12}:
                keep[i] = 0
    w = hooked_world(lib, 32, bodies, colliders, np.array([False, False, True]), None, which=F.COLLIDER_FILTER_PAIRS)
    w.collision_hooks_set(flt, None)
    w.pipeline_enable()
    seen = set()
    for step in range(150):
        n0 = len(asked)
        w.step()
        assert not w.host_shape_errors()
        for p in w.pairs_get():
            seen.add((int(p["collider1"]), int(p["collider2"])))
        if 4 <= step:   # (while B falls through A their AABBs overlap: the pair is not in the graph, so the sweep proposes it again -- from the step the gap closes)
            assert any({a, b} == {11, 12} for a, b in asked[n0:]), f"step {step}: the rejected pair is proposed again"
        assert w.collision_hook_stats().last_filter_rejected == sum(1 for a, b in asked[n0:] if {a, b} == {11, 12})
    assert not any({a, b} == {11, 12} for a, b in seen), "a rejected pair gets no ContactId"
    assert any({a, b} == {10, 12} for a, b in seen), "B against the ground (asked: B carries the flag; accepted)"
    pos = w.bodies_download()["position"]
    # B fell THROUGH A: both rest on the slab
    assert abs(pos[1][1] - 0.5) < 0.03 and abs(pos[2][1] - 0.5) < 0.03, pos


def test_modify_contacts_false_keeps_a_pair_from_touching():
    lib = oracle_lib()
    bodies, colliders = _two_boxes_on_ground()
    shown = []

    def mod(recs):
        for i in range(len(recs)):
            shown.append((int(recs["collider1"][i]), int(recs["collider2"][i]), int(recs["point_count"][i])))
            if {int(recs["collider1"][i]), int(recs["collider2"][i])} == {11, 12}:
                recs["touching"][i] = 0
    w = hooked_world(lib, 32, bodies, colliders, np.array([False, False, True]), None, which=F.COLLIDER_MODIFY_CONTACTS)
    w.collision_hooks_set(None, mod)
    w.pipeline_enable()
    for step in range(150):
        w.step()
        assert not w.host_shape_errors()
        offs, handles = w.pipeline_handles()
        rows = w.contacts_download(np.sort(handles))
        # the A-B pair exists (it has a ContactId) but never generates a constraint: every row with a constraint touches
        assert np.all(rows["flags"] & F.CP_TOUCHING)
    assert any({a, b} == {11, 12} for a, b, _ in shown) and any({a, b} == {10, 12} for a, b, _ in shown)
    pos = w.bodies_download()["position"]
    assert abs(pos[1][1] - 0.5) < 0.03 and abs(pos[2][1] - 0.5) < 0.03, pos


@pytest.mark.parametrize("bits", [32, 64])
def test_tangent_velocity_from_the_hook_is_a_conveyor_belt(bits):
    lib = oracle_lib()
    bodies, colliders = _two_boxes_on_ground()
    for k in ("position", "rotation", "linear_velocity", "angular_velocity", "inv_mass", "inv_inertia_local", "rb_type"):
        bodies[k] = bodies[k][:2]
    for k in ("entity_index", "body", "shape", "half_extents"):
        colliders[k] = colliders[k][:2]

    def mod(recs):
        recs["tangent_velocity"][:] = np.array([1.5, 0.0, 0.0], recs["friction"].dtype)
    w = hooked_world(lib, bits, bodies, colliders, np.array([True, False]), None, which=F.COLLIDER_MODIFY_CONTACTS, friction=0.9)
    w.collision_hooks_set(None, mod)
    w.pipeline_enable()
    for _ in range(240):
        w.step()
    assert not w.host_shape_errors()
    b = w.bodies_download()
    # the box is dragged along +x or -x by the belt (the sign follows the manifold's normal orientation: relative surface velocity of body 2 against body 1)
    assert abs(abs(b["linear_velocity"][1][0]) - 1.5) < 0.05, b["linear_velocity"][1]
    assert abs(b["position"][1][0]) > 2.0 and abs(b["position"][1][1] - 0.5) < 0.03
    assert w.collision_hook_stats().last_modify_queries == 1


def test_hooks_in_the_host_bookkeeping_mode_and_the_systems():
    """AVN_SYS_BROAD_PHASE / AVN_SYS_NARROW_PHASE with the hooks registered (an Avian integration's ContactPipeline): same world as the closed loop."""
    from avian_amd.pipeline import ContactPipeline
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=6, n=40)
    rng = np.random.default_rng(6)
    flagged = rng.random(len(colliders["shape"])) < 0.5
    ha, hb = Hooks(), Hooks()
    loop = hooked_world(lib, 32, bodies, colliders, flagged, ha)
    sysw = hooked_world(lib, 32, bodies, colliders, flagged, hb)
    loop.pipeline_enable()
    pipe = ContactPipeline(sysw, lib)
    for step in range(25):
        loop.step(); pipe.step()
        assert not loop.host_shape_errors() and not sysw.host_shape_errors()
        a, b = loop.bodies_download(), sysw.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
    assert ha.filter_log == hb.filter_log and ha.modify_log == hb.modify_log
    assert ha.rejected > 0 and ha.untouched > 0
