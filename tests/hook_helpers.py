"""Collision hooks (include/avian_mi355x.h "collision hooks"): test doubles of CollisionHooks::filter_pairs / modify_contacts (reference collision/hooks.rs:137-231).
The hooks are PURE functions of what they are shown, so two backends that show them the same pairs and records must end in the same world; everything shown is logged
so that a test can also compare WHAT the hooks saw (the oracle calls back one record at a time, the HIP library in batches: the concatenations must agree)."""
from __future__ import annotations

import numpy as np

from helpers import F


class Hooks:
    """filter: rejects the pair when (collider1 * 7 + collider2 * 3) % reject_mod == 0.  modify, by contact id: id % 4 == 0 -> not touching (the hook returns false);
    == 1 -> one point fewer, half the friction, a tangent (conveyor) velocity; == 2 -> restitution 0.25 and the manifold's first two points swapped; == 3 -> untouched."""

    def __init__(self, reject_mod=5, modify=True, identity=False):
        self.reject_mod, self.do_modify, self.identity = reject_mod, modify, identity
        self.filter_log, self.modify_log = [], []
        self.filter_calls = self.modify_calls = 0
        self.rejected = self.untouched = 0

    def filter(self, pairs, keep):
        self.filter_calls += 1
        assert np.all(keep == 1), "should_collide is preset to 1"
        assert np.all(np.diff(pairs["index"].astype(np.int64)) > 0), "pairs arrive in emission order"
        for i in range(len(pairs)):
            c1, c2 = int(pairs["collider1"][i]), int(pairs["collider2"][i])
            self.filter_log.append((int(pairs["index"][i]), c1, c2))
            if not self.identity and (c1 * 7 + c2 * 3) % self.reject_mod == 0:
                keep[i] = 0; self.rejected += 1

    def modify(self, recs):
        self.modify_calls += 1
        assert np.all(np.diff(recs["contact_id"].astype(np.int64)) > 0), "records arrive in ascending contact id"
        for i in range(len(recs)):
            r = recs[i]
            assert r["touching"] == 1 and r["manifold_count"] == 1 and 1 <= r["point_count"] <= 4 and r["flags"] & F.CP_MODIFY_CONTACTS
            assert np.all(r["tangent_velocity"] == 0)
            self.modify_log.append(r.tobytes())
            if self.identity or not self.do_modify:
                continue
            k = int(r["contact_id"]) % 4
            S = recs["friction"].dtype.type
            if k == 0:
                recs["touching"][i] = 0; self.untouched += 1
            elif k == 1:
                recs["point_count"][i] = max(1, int(r["point_count"]) - 1)
                recs["friction"][i] = r["friction"] * S(0.5)
                recs["tangent_velocity"][i] = np.array([0.25, 0.0, -0.125], S)
            elif k == 2:
                recs["restitution"][i] = S(0.25)
                if r["point_count"] >= 2:
                    for f in ("anchor1", "anchor2", "penetration", "normal_speed", "feature_id1", "feature_id2"):
                        a = recs[f][i][0].copy(); recs[f][i][0] = recs[f][i][1]; recs[f][i][1] = a


def hooked_world(lib, bits, bodies, colliders, flag_mask, hooks: Hooks | None, substeps=4, friction=0.6, which=F.COLLIDER_FILTER_PAIRS | F.COLLIDER_MODIFY_CONTACTS,
                 register=(True, True)):
    cols = dict(colliders)
    base = np.asarray(colliders.get("collider_flags", np.zeros(len(colliders["shape"]), np.uint8)), np.uint8)
    cols["collider_flags"] = (base | np.where(flag_mask, which, 0)).astype(np.uint8)
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    w.bodies_upload(**bodies); w.colliders_upload(**cols)
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction, restitution=0.0)
    if hooks is not None:
        w.collision_hooks_set(hooks.filter if register[0] else None, hooks.modify if register[1] else None)
    return w


def assert_same_hooked_step(a: F.World, b: F.World, step, compare_flags=True):
    """Two closed-loop worlds after the same step: bodies, the new pairs with their order and ids, colour lists, every live row."""
    assert not a.host_shape_errors() and not b.host_shape_errors()
    x, y = a.bodies_download(), b.bodies_download()
    for k in x:
        assert np.array_equal(x[k], y[k]), f"step {step}: bodies.{k}"
    pa, pb = a.pairs_get(), b.pairs_get()
    if compare_flags:
        assert np.array_equal(pa, pb), f"step {step}: the new pairs (order included)"
    else:
        assert np.array_equal(pa["collider1"], pb["collider1"]) and np.array_equal(pa["collider2"], pb["collider2"]), f"step {step}: the new pairs (order included)"
    assert np.array_equal(a.pipeline_new_pair_ids(), b.pipeline_new_pair_ids())
    (oa, ha), (ob, hb) = a.pipeline_handles(), b.pipeline_handles()
    assert np.array_equal(oa, ob) and np.array_equal(ha, hb), f"step {step}: colour lists"
    ids = np.sort(ha)
    ra, rb = a.contacts_download(ids), b.contacts_download(ids)
    for k in ra:
        if k == "flags" and not compare_flags:
            assert np.array_equal(ra[k] & ~np.uint32(F.CP_MODIFY_CONTACTS), rb[k] & ~np.uint32(F.CP_MODIFY_CONTACTS)), f"step {step}: contact rows.flags"
        else:
            assert np.array_equal(ra[k], rb[k]), f"step {step}: contact rows.{k}"
