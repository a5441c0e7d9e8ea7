"""GPU: collision hooks on the HIP backend (include/avian_mi355x.h "collision hooks", avian_amd/csrc/world/hooks.hpp).  CollisionHooks::filter_pairs /
modify_contacts (reference collision/hooks.rs:137-231; call sites broad_phase.rs:431-439, narrow_phase/system_param.rs:770-778) are host callbacks; the pairs before
and after them stay on the device (k_hook_filter_collect / _compact in front of k_pg_add_pairs; three phases of the narrow phase around modify_contacts).  The hooks
of tests/hook_helpers.py are pure functions of what they are shown, so the HIP world must equal the oracle world bit for bit every step -- and the hooks must have
been SHOWN the same pairs and records (the logs are compared too)."""
import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from hook_helpers import Hooks, assert_same_hooked_step, hooked_world
from host_shape_helpers import HostShapes
from pipeline_scenes import dropped_boxes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,seed,n,steps", [(32, 1, 48, 40), (64, 2, 48, 40), (32, 5, 400, 30), (64, 8, 300, 20)])
def test_hooked_world_equals_the_oracle(bits, seed, n, steps):
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=seed, n=n)
    rng = np.random.default_rng(seed)
    flagged = rng.random(len(colliders["shape"])) < 0.4
    flagged[0] = seed % 2 == 0   # (the ground too, in half of the cases: every resting contact goes through the hook)
    hd, ho = Hooks(), Hooks()
    dev = hooked_world(hip, bits, bodies, colliders, flagged, hd)
    ref = hooked_world(orc, bits, bodies, colliders, flagged, ho)
    dev.pipeline_enable(); ref.pipeline_enable()
    for step in range(steps):
        dev.step(); ref.step()
        assert_same_hooked_step(dev, ref, step)
        sd, so = dev.collision_hook_stats(), ref.collision_hook_stats()
        for k in ("last_filter_queries", "last_filter_rejected", "last_modify_queries", "last_modify_rejected"):
            assert getattr(sd, k) == getattr(so, k), f"step {step}: {k}"
    assert hd.filter_log == ho.filter_log, "filter_pairs was asked about the same pairs in the same order"
    assert hd.modify_log == ho.modify_log, "modify_contacts was shown the same ContactPairs, bit for bit"
    assert hd.rejected > 0 and hd.untouched > 0 and len(hd.modify_log) > 50
    # the bus carried the hooks' pairs and nothing else
    st = dev.collision_hook_stats()
    rec = F.hook_contact_dtype(bits).itemsize
    assert st.bytes_to_host == len(hd.filter_log) * F.HOOK_PAIR_DTYPE.itemsize + len(hd.modify_log) * rec
    assert st.bytes_from_host == len(hd.filter_log) + len(hd.modify_log) * rec


def test_identity_hooks_leave_the_hip_world_bit_identical():
    hip = hip_lib()
    bodies, colliders = dropped_boxes(seed=3, n=200)
    rng = np.random.default_rng(3)
    flagged = rng.random(len(colliders["shape"])) < 0.5
    hooks = Hooks(identity=True)
    plain = hooked_world(hip, 32, bodies, colliders, np.zeros_like(flagged), None)
    hooked = hooked_world(hip, 32, bodies, colliders, flagged, hooks)
    plain.pipeline_enable(); hooked.pipeline_enable()
    for step in range(40):
        plain.step(); hooked.step()
        assert_same_hooked_step(plain, hooked, step, compare_flags=False)
    assert len(hooks.filter_log) > 100 and len(hooks.modify_log) > 1000


@pytest.mark.parametrize("register", [(True, False), (False, True)])
def test_one_hook_registered_the_other_at_its_default(register):
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=9, n=80)
    flagged = np.random.default_rng(9).random(len(colliders["shape"])) < 0.5
    hd, ho = Hooks(), Hooks()
    dev = hooked_world(hip, 32, bodies, colliders, flagged, hd, register=register)
    ref = hooked_world(orc, 32, bodies, colliders, flagged, ho, register=register)
    dev.pipeline_enable(); ref.pipeline_enable()
    for step in range(30):
        dev.step(); ref.step()
        assert_same_hooked_step(dev, ref, step)
    assert hd.filter_log == ho.filter_log and hd.modify_log == ho.modify_log
    assert (len(hd.filter_log) > 0) == register[0] and (len(hd.modify_log) > 0) == register[1]


def test_hooks_on_pairs_whose_manifold_is_the_hosts():
    """Host shapes and hooks together: a pair's manifold comes from contact_manifolds_with_context on the host, goes through the device's speculative filter and
    prune_points, is shown to modify_contacts, and is matched against the previous step's points on the device."""
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=12, n=60)
    rng = np.random.default_rng(12)
    host = rng.random(len(colliders["shape"])) < 0.4
    flagged = rng.random(len(colliders["shape"])) < 0.5
    worlds, hooks = [], []
    for lib in (hip, orc):
        h = Hooks()
        cols = dict(colliders, shape=np.where(host, F.SHAPE_HOST, colliders["shape"]).astype(np.uint8))
        w = hooked_world(lib, 32, bodies, cols, flagged, h)
        hs = HostShapes(F.World(orc, F.default_config(32)), colliders["entity_index"], colliders["shape"], colliders["half_extents"])
        w.host_shapes_set(hs.aabb, hs.manifolds)
        w._hs_obj = hs
        w.pipeline_enable()
        worlds.append(w); hooks.append(h)
    dev, ref = worlds
    both = 0
    for step in range(40):
        dev.step(); ref.step()
        assert_same_hooked_step(dev, ref, step)
    assert hooks[0].filter_log == hooks[1].filter_log and hooks[0].modify_log == hooks[1].modify_log
    ent = np.asarray(colliders["entity_index"])
    host_e = set(int(e) for e in ent[host])
    dt = F.hook_contact_dtype(32)
    for raw in hooks[0].modify_log:
        r = np.frombuffer(raw, dt)[0]
        both += int(r["collider1"]) in host_e or int(r["collider2"]) in host_e
    assert both > 20, "pairs with a host-shaped collider must have reached the hook"


def test_hooks_with_sleeping_enabled():
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=14, n=40)
    flagged = np.random.default_rng(14).random(len(colliders["shape"])) < 0.5
    hd, ho = Hooks(reject_mod=7), Hooks(reject_mod=7)
    dev = hooked_world(hip, 32, bodies, colliders, flagged, hd)
    ref = hooked_world(orc, 32, bodies, colliders, flagged, ho)
    for w in (dev, ref):
        w.pipeline_enable(); w.sleeping_enable()
    slept = 0
    for step in range(200):
        dev.step(); ref.step()
        a, b = dev.bodies_download(), ref.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        sa, sb = dev.sleeping_state(), ref.sleeping_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), f"step {step}: sleeping state.{k}"
        slept = max(slept, int(sa["sleeping"].sum()))
    assert hd.filter_log == ho.filter_log and hd.modify_log == ho.modify_log
    assert slept > 0


def test_hooks_in_the_host_bookkeeping_mode():
    from avian_amd.pipeline import ContactPipeline
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=6, n=40)
    flagged = np.random.default_rng(6).random(len(colliders["shape"])) < 0.5
    hd, ho = Hooks(), Hooks()
    dev = hooked_world(hip, 32, bodies, colliders, flagged, hd)
    ref = hooked_world(orc, 32, bodies, colliders, flagged, ho)
    pd, pr = ContactPipeline(dev, hip), ContactPipeline(ref, orc)
    for step in range(25):
        pd.step(); pr.step()
        assert not dev.host_shape_errors()
        a, b = dev.bodies_download(), ref.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        assert sorted(pd.pairs) == sorted(pr.pairs) and list(pd.active) == list(pr.active)
    assert hd.filter_log == ho.filter_log and hd.modify_log == ho.modify_log and hd.rejected > 0 and hd.untouched > 0


def test_a_pile_with_a_twentieth_of_its_boxes_hooked_stays_in_the_closed_loop():
    """Scale: 20 000 boxes, 5 % of them with ActiveCollisionHooks -- per step the bus carries those pairs only (the HostNarrowPhase mode re-sent every manifold)."""
    import time
    from avian_amd import scenes
    hip = hip_lib()
    sc = scenes.box_stack(20, 50, 20)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    rng = np.random.default_rng(0)
    flagged = rng.random(len(colliders["shape"])) < 0.05
    flagged[0] = False

    class Fast:
        shown = asked = 0
        def filter(self, pairs, keep): Fast.asked += len(pairs)
        def modify(self, recs):
            Fast.shown += len(recs)
            recs["friction"][:] *= recs["friction"].dtype.type(0.75)
    plain = hooked_world(hip, 32, bodies, colliders, np.zeros_like(flagged), None)
    hooked = hooked_world(hip, 32, bodies, colliders, flagged, Fast())
    times = {}
    for name, w in (("plain", plain), ("hooked", hooked)):
        w.pipeline_enable()
        for _ in range(30):
            w.step()
        w.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            w.step()
        w.synchronize()
        times[name] = (time.perf_counter() - t0) / 20 * 1e3
    st = hooked.collision_hook_stats()
    assert not hooked.host_shape_errors() and Fast.shown > 1000
    rec = F.hook_contact_dtype(32).itemsize
    print(f"\n[collision hooks] 20 000 boxes: {times['plain']:.2f} ms/step without hooks, {times['hooked']:.2f} ms/step with 5 % of the boxes hooked "
          f"({st.last_modify_queries} records = {st.last_modify_queries * rec / 1e3:.0f} kB each way per step, callback {st.last_callback_ms:.2f} ms)")
    assert st.last_modify_queries * rec < 5e6
