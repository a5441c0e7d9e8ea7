"""Host shapes (include/avian_mi355x.h "host shapes"): test doubles of the two AnyCollider methods.  The colliders a test flags AVN_SHAPE_HOST really are parry Balls /
Cuboids whose geometry only the HOST knows: `aabb` restates parry's Cuboid::aabb / Ball::aabb in numpy scalars of the world's type (operation order of
k_broadphase.hip shape_aabb / oracle shape_aabb), `manifolds` answers through the batch query avn_contact_manifolds of a helper world -- so a world with host-flagged
colliders must reproduce the all-native world bit for bit: same pairs in the same order, same rows, same bodies."""
from __future__ import annotations

import numpy as np

from helpers import F


def shape_aabb(shape, he, pos, q, S):
    he = np.asarray(he, S); pos = np.asarray(pos, S)
    if shape == F.SHAPE_BALL:
        h = np.array([he[0], he[0], he[0]], S)
    else:
        i, j, k, w = (S(x) for x in q)
        two = S(2)
        ww, ii, jj, kk = w * w, i * i, j * j, k * k
        ij, wk, wj, ik, jk, wi = i * j * two, w * k * two, w * j * two, i * k * two, j * k * two, w * i * two
        m = [[abs(ww + ii - jj - kk), abs(ij - wk), abs(wj + ik)], [abs(wk + ij), abs(ww - ii + jj - kk), abs(jk - wi)], [abs(ik - wj), abs(wi + jk), abs(ww - ii - jj + kk)]]
        h = np.array([(m[r][0] * he[0] + m[r][1] * he[1]) + m[r][2] * he[2] for r in range(3)], S)
    return pos - h, pos + h


class HostShapes:
    """Callbacks for a world whose colliders `real_shape` / `half_extents` (by entity index) live on the host."""

    def __init__(self, query_world: F.World, entity_index, real_shape, half_extents):
        self.qw = query_world
        self.shape = {int(e): int(s) for e, s in zip(entity_index, real_shape)}
        self.he = {int(e): np.asarray(h, np.float64) for e, h in zip(entity_index, half_extents)}
        self.aabb_calls = 0; self.manifold_calls = 0; self.aabb_queries = 0; self.manifold_queries = 0

    def aabb(self, q, out):
        self.aabb_calls += 1; self.aabb_queries += len(q)
        S = q["start_position"].dtype.type
        for i in range(len(q)):
            e = int(q["collider"][i])
            mn, mx = shape_aabb(self.shape[e], self.he[e], q["start_position"][i], q["start_rotation"][i], S)
            if q["swept"][i]:   # AnyCollider::swept_aabb_with_context's default: aabb(start).merged(aabb(end))
                mn1, mx1 = shape_aabb(self.shape[e], self.he[e], q["end_position"][i], q["end_rotation"][i], S)
                mn, mx = np.minimum(mn, mn1), np.maximum(mx, mx1)
            out["min"][i] = mn; out["max"][i] = mx

    def manifolds(self, q, out):
        self.manifold_calls += 1; self.manifold_queries += len(q)
        n = len(q)
        assert np.all(np.diff(q["contact_id"].astype(np.int64)) > 0), "queries arrive in ascending contact id"
        e1, e2 = q["collider1"], q["collider2"]
        r = self.qw.contact_manifolds([self.shape[int(e)] for e in e1], np.array([self.he[int(e)] for e in e1]), q["position1"], q["rotation1"],
                                      [self.shape[int(e)] for e in e2], np.array([self.he[int(e)] for e in e2]), q["position2"], q["rotation2"], q["max_contact_distance"])
        out["point_count"][:] = r["point_count"]
        out["normal"][:] = r["normal"]
        out["anchor1"][:] = r["anchor1"].reshape(n, F.MAX_QUERY_POINTS, 3)
        out["penetration"][:] = r["penetration"].reshape(n, F.MAX_QUERY_POINTS)
        out["feature_id1"][:] = r["feature_id1"].reshape(n, F.MAX_QUERY_POINTS)
        out["feature_id2"][:] = r["feature_id2"].reshape(n, F.MAX_QUERY_POINTS)


def make_pair(lib, query_lib, bits, bodies, colliders, host_mask, substeps=4, friction=0.6):
    """(native world, world whose masked colliders are host shapes, its HostShapes)."""
    def mk(cols):
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**bodies); w.colliders_upload(**cols)
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction, restitution=0.0)
        return w
    native = mk(colliders)
    hc = dict(colliders)
    hc["shape"] = np.where(host_mask, F.SHAPE_HOST, colliders["shape"]).astype(np.uint8)
    hosted = mk(hc)
    qw = F.World(query_lib, F.default_config(bits))
    hs = HostShapes(qw, colliders["entity_index"], colliders["shape"], colliders["half_extents"])
    hosted.host_shapes_set(hs.aabb, hs.manifolds)
    return native, hosted, hs


def assert_same_closed_loop_step(native: F.World, hosted: F.World, step):
    assert not hosted.host_shape_errors()
    a, b = native.bodies_download(), hosted.bodies_download()
    for k in a:
        assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
    pa, pb = native.pairs_get(), hosted.pairs_get()
    assert np.array_equal(pa, pb), f"step {step}: the broad phase's new pairs (order included)"
    assert np.array_equal(native.pipeline_new_pair_ids(), hosted.pipeline_new_pair_ids())
    (oa, ha), (ob, hb) = native.pipeline_handles(), hosted.pipeline_handles()
    assert np.array_equal(oa, ob) and np.array_equal(ha, hb), f"step {step}: colour lists"
    ids = np.sort(ha)
    ra, rb = native.contacts_download(ids), hosted.contacts_download(ids)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), f"step {step}: contact rows.{k}"
    mna, mxa, _ = native.aabbs_download(); mnb, mxb, _ = hosted.aabbs_download()
    assert np.array_equal(mna, mnb) and np.array_equal(mxa, mxb), f"step {step}: ColliderAabb"


# ---- a shape the device does NOT hold: capsules (segment +-half_height along local y, radius), entirely on the host --------------------------------------------
def _qrot(q, v):
    x, y, z, w = q
    u = np.array([x, y, z]); v = np.asarray(v, np.float64)
    return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)


def _closest_segment_points(p1, q1, p2, q2):
    """Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9)."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    c = d1 @ r; b = d1 @ d2; den = a * e - b * b
    s = np.clip((b * f - c * e) / den, 0.0, 1.0) if den > 1e-12 else 0.0
    t = (b * s + f) / e
    if t < 0.0:
        t, s = 0.0, np.clip(-c / a, 0.0, 1.0)
    elif t > 1.0:
        t, s = 1.0, np.clip((b - c) / a, 0.0, 1.0)
    return p1 + d1 * s, p2 + d2 * t


class CapsuleShapes:
    """AnyCollider for capsules against an axis-aligned ground cuboid and against each other -- a host implementation of a shape the library has no kernel for.
    Written in float64 and rounded to the world's type on the way out: any backend that gets these answers must produce the same world."""

    def __init__(self, capsules, ground_entity, ground_top):
        self.cap = {int(e): (float(h), float(r)) for e, (h, r) in capsules.items()}
        self.ground, self.top = int(ground_entity), float(ground_top)
        self.queries = 0

    def _ends(self, e, pos, rot):
        h, r = self.cap[e]
        pos = np.asarray(pos, np.float64); rot = np.asarray(rot, np.float64)
        return pos + _qrot(rot, [0, -h, 0]), pos + _qrot(rot, [0, h, 0]), r

    def aabb(self, q, out):
        for i in range(len(q)):
            e = int(q["collider"][i])
            pts = list(self._ends(e, q["start_position"][i], q["start_rotation"][i])[:2])
            if q["swept"][i]:
                pts += list(self._ends(e, q["end_position"][i], q["end_rotation"][i])[:2])
            r = self.cap[e][1]
            out["min"][i] = np.min(pts, axis=0) - r; out["max"][i] = np.max(pts, axis=0) + r

    def manifolds(self, q, out):
        self.queries += len(q)
        for i in range(len(q)):
            e1, e2 = int(q["collider1"][i]), int(q["collider2"][i])
            p1, p2 = np.asarray(q["position1"][i], np.float64), np.asarray(q["position2"][i], np.float64)
            mcd = float(q["max_contact_distance"][i])
            pts = []   # (point on shape 1 relative to position1, signed distance), normal from 1 to 2
            if e1 == self.ground or e2 == self.ground:
                cap_is_2 = e1 == self.ground
                a, b, r = self._ends(e2 if cap_is_2 else e1, q["position2"][i] if cap_is_2 else q["position1"][i], q["rotation2"][i] if cap_is_2 else q["rotation1"][i])
                normal = np.array([0.0, 1.0, 0.0]) if cap_is_2 else np.array([0.0, -1.0, 0.0])
                for k, end in enumerate((a, b)):
                    dist = end[1] - r - self.top
                    if dist < mcd:
                        on1 = np.array([end[0], self.top, end[2]]) if cap_is_2 else end - np.array([0.0, r, 0.0])
                        pts.append((on1 - p1, dist, k + 1))
            else:
                a1, b1, r1 = self._ends(e1, q["position1"][i], q["rotation1"][i]); a2, b2, r2 = self._ends(e2, q["position2"][i], q["rotation2"][i])
                c1, c2 = _closest_segment_points(a1, b1, a2, b2)
                d = c2 - c1; L = float(np.linalg.norm(d))
                if L > 1e-9 and L - r1 - r2 < mcd:
                    normal = d / L
                    pts.append((c1 + normal * r1 - p1, L - r1 - r2, 1))
            out["point_count"][i] = len(pts)
            if pts:
                out["normal"][i] = normal
                for k, (on1, dist, fid) in enumerate(pts):
                    out["anchor1"][i, k] = on1 + normal * dist * 0.5      # contact_query.rs:243-246
                    out["penetration"][i, k] = -dist
                    out["feature_id1"][i, k] = fid; out["feature_id2"][i, k] = fid


def capsule_scene(n=24, seed=0):
    """n capsules (half height 0.4, radius 0.25) dropped tumbling over a static ground slab; all of them AVN_SHAPE_HOST, the ground a native cuboid."""
    from avian_amd import scenes
    rng = np.random.default_rng(seed)
    base = scenes.box_stack(1, 1, 1)
    m = n + 1
    pos = np.zeros((m, 3)); pos[0] = base.position[0]
    side = int(np.ceil(np.sqrt(n)))
    for k in range(n):
        pos[1 + k] = [1.6 * (k % side) + rng.uniform(-0.2, 0.2), 1.2 + 0.9 * (k // side % 3) + rng.uniform(0, 0.3), 1.6 * (k // side) * 0.6 + rng.uniform(-0.2, 0.2)]
    q = rng.normal(size=(m, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[0] = [0, 0, 0, 1]
    h, r = 0.4, 0.25
    vol = np.pi * r * r * (2 * h) + 4 / 3 * np.pi * r ** 3
    inv_mass = np.full(m, 1.0 / vol); inv_mass[0] = 0.0
    iy = 0.5 * vol * r * r; ix = vol * (3 * r * r + (2 * h) ** 2) / 12.0 + vol * h * h * 0.3
    ii = np.tile([1 / ix, 0, 0, 1 / iy, 0, 1 / ix], (m, 1)); ii[0] = 0
    rb = np.zeros(m, np.uint8); rb[0] = F.RB_STATIC
    he = np.tile([r, h + r, r], (m, 1)).astype(np.float64); he[0] = base.half_extents[0]
    shape = np.full(m, F.SHAPE_HOST, np.uint8); shape[0] = F.SHAPE_CUBOID
    bodies = dict(position=pos, rotation=q, linear_velocity=rng.normal(scale=0.3, size=(m, 3)) * (rb == 0)[:, None], angular_velocity=rng.normal(scale=1.0, size=(m, 3)) * (rb == 0)[:, None],
                  inv_mass=inv_mass, inv_inertia_local=ii, rb_type=rb)
    colliders = dict(entity_index=np.arange(m, dtype=np.uint32) + 10, body=np.arange(m, dtype=np.int32), shape=shape, half_extents=he)
    caps = {int(e): (h, r) for e in colliders["entity_index"][1:]}
    return bodies, colliders, caps, float(base.position[0][1] + base.half_extents[0][1])


def capsule_world(lib, bits, n=24, seed=0):
    bodies, colliders, caps, top = capsule_scene(n, seed)
    w = F.World(lib, F.default_config(bits, substeps=4))
    w.bodies_upload(**bodies); w.colliders_upload(**colliders)
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.6, restitution=0.0)
    cs = CapsuleShapes(caps, int(colliders["entity_index"][0]), top)
    w.host_shapes_set(cs.aabb, cs.manifolds)
    w.pipeline_enable()
    return w, cs, top
