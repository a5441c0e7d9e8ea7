"""CPU: contact manifolds of the oracle (and, through tests/test_gpu_narrow.py's bit-identity, of the HIP product) against an
INDEPENDENT float64 computation (tests/narrow_bruteforce.py) and against hand-derived configurations.  This is what pins the parry3d part of
the narrow phase -- written from the published algorithm, crate source unavailable -- to geometry rather than to its sibling implementation."""
import numpy as np
import pytest

from helpers import F, oracle_lib, random_unit_quats
from narrow_bruteforce import predict
from narrow_scenes import quat_axis_angle

I = [0.0, 0.0, 0.0, 1.0]


@pytest.fixture(scope="module", params=[32, 64])
def world(request):
    return F.World(oracle_lib(), F.default_config(request.param))


def manifold(world, he1, p1, r1, he2, p2, r2, pred=0.0, s1=0, s2=0):
    o = world.contact_manifolds([s1], [he1], [p1], [r1], [s2], [he2], [p2], [r2], [pred])
    k = int(o["point_count"][0])
    return k, o["normal"][0].astype(float), (np.asarray(p1, float) + o["anchor1"][0, :k]).astype(float), o["penetration"][0, :k].astype(float), \
        o["feature_id1"][0, :k], o["feature_id2"][0, :k]


def same_point_set(a, b, tol):
    if len(a) != len(b):
        return False
    used = set()
    for q in a:
        j = [i for i in range(len(b)) if i not in used and np.linalg.norm(q - b[i]) < tol]
        if not j:
            return False
        used.add(j[0])
    return True


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_face_contacts_equal_the_projected_polygon_intersection(world, seed):
    check_projected_polygon_intersection(world, seed)


def check_projected_polygon_intersection(world, seed):
    """Random tilted cuboid pairs pressed together: the point set must be the intersection of the two support-face polygons projected along
    the manifold normal (vertices lifted back onto both faces, midpoints reported, depth = gap, points beyond the prediction distance
    dropped) -- for face axes and for edge x edge axes alike -- and the normal must be a least-penetration axis of the 15-axis SAT."""
    rng = np.random.default_rng(seed)
    tol = 3e-5 if world.dtype == np.float32 else 1e-9
    checked = edge = skipped = 0
    for _ in range(400):
        he1, he2 = rng.uniform(0.3, 1.2, 3), rng.uniform(0.3, 1.2, 3)
        q1 = random_unit_quats(rng, 1)[0]
        tilt = quat_axis_angle(rng.normal(size=3), rng.uniform(0.0, 0.25))   # box 2 = box 1's frame tilted by up to ~14 degrees: face contacts
        x, y, z, w = q1; a, b, c, d = tilt
        q2 = np.array([d * x + a * w + b * z - c * y, d * y - a * z + b * w + c * x, d * z + a * y - b * x + c * w, d * w - a * x - b * y - c * z])
        k_ax = rng.integers(0, 3)
        from narrow_bruteforce import rot_matrix
        R1 = rot_matrix(q1)
        p1 = rng.uniform(-2, 2, 3)
        lateral = R1[:, (k_ax + 1) % 3] * rng.uniform(-0.3, 0.3) + R1[:, (k_ax + 2) % 3] * rng.uniform(-0.3, 0.3)
        p2 = p1 + R1[:, k_ax] * (he1[k_ax] + he2[k_ax]) * rng.uniform(0.9, 0.99) + lateral
        pred = float(rng.choice([0.0, 0.05]))
        k, n, pts, pen, _, _ = manifold(world, he1, p1, q1, he2, p2, q2, pred)
        if k == 0:
            continue
        kind, best, sep_n, want_pts, want_pen = predict(he1, p1, q1, he2, p2, q2, n, pred, tol)
        assert sep_n >= best - 50 * tol, f"the manifold normal is not a least-penetration axis: {sep_n} vs {best}"
        assert abs(pen.max() + sep_n) < 50 * tol, "the deepest point realises the separation along the normal"
        edge += kind == "edge"
        if len(want_pts) != k:   # (a vertex within rounding of a clipping edge may be kept by one computation and merged by the other)
            skipped += 1
            continue
        assert same_point_set(pts, want_pts, 200 * tol), (pts, want_pts)
        order = [int(np.argmin(np.linalg.norm(want_pts - q, axis=1))) for q in pts]
        assert np.allclose(pen, want_pen[order], atol=200 * tol)
        checked += 1
    assert checked > 250 and edge > 30 and skipped <= 0.05 * checked, (checked, edge, skipped)


def test_vertex_on_face_is_one_point_under_the_vertex(world):
    # cube 2 balanced on a corner (body diagonal vertical) over the top face of a big slab
    qd = quat_axis_angle(np.cross([1, 1, 1], [0, 1, 0]), np.arccos(1 / np.sqrt(3)))   # rotates (1,1,1)/sqrt3 onto +y
    h = 0.5 * np.sqrt(3)
    k, n, pts, pen, f1, f2 = manifold(world, [3, 0.5, 3], [0, 0, 0], I, [.5, .5, .5], [0.4, 0.5 + h - 0.02, -0.7], qd)
    # parry reports the whole clipped support face (the prediction distance gates manifolds, not points): the lowest vertex penetrates by
    # 0.02, the other three corners of that face are clear of the slab; Avian's own pruning (system_param.rs:731-757) drops those
    assert k == 4 and np.allclose(n, [0, 1, 0], atol=1e-6) and np.isclose(pen.max(), 0.02, atol=1e-5) and (np.sort(pen)[:3] < -0.5).all()
    assert np.allclose(pts[int(np.argmax(pen))], [0.4, 0.5 - 0.01, -0.7], atol=1e-5), "the contact sits under the lowest vertex, midway between vertex and face"


def test_edge_on_face_is_the_clipped_segment(world):
    # cube 2 rolled 45 degrees about z: its lowest EDGE (along z, length 1) rests on the slab, hanging 0.3 over the slab's +z border
    k, n, pts, pen, f1, f2 = manifold(world, [2, 0.5, 1], [0, 0, 0], I, [.5, .5, .5], [0.2, 0.5 + np.sqrt(0.5) - 0.01, 0.8], quat_axis_angle([0, 0, 1], np.pi / 4))
    assert k == 4 and np.allclose(n, [0, 1, 0], atol=1e-6)
    low = pen > 0          # the two ends of the resting edge; the other two corners of the support face are clear of the slab
    assert low.sum() == 2 and np.allclose(pen[low], 0.01, atol=1e-5) and (pen[~low] < -0.5).all()
    assert np.allclose(sorted(pts[low, 2]), [0.3, 1.0], atol=1e-5) and np.allclose(pts[low, 0], 0.2, atol=1e-5)
    assert len(set(zip(f1.tolist(), f2.tolist()))) == 4


def test_parallel_edges_degenerate_cross_product(world):
    """Two cubes rolled 45 degrees about the SAME axis, touching edge to edge.  The geometric minimum-translation direction (x) is not
    among the 15 SAT axes -- the edge x edge products of parallel edges vanish and are skipped -- so the least-penetration axis is one of
    the 45-degree face normals, with depth 0.02 cos 45; the patch is the clipped support faces along that normal (second opinion)."""
    r = quat_axis_angle([0, 0, 1], np.pi / 4)
    he, p1, p2 = [.5, .5, .5], [0, 0, 0], [2 * np.sqrt(0.5) - 0.02, 0, 0.4]
    k, n, pts, pen, _, _ = manifold(world, he, p1, r, he, p2, r)
    tol = 3e-5 if world.dtype == np.float32 else 1e-9
    assert k >= 2 and np.isclose(abs(n[0]), np.sqrt(0.5), atol=1e-6) and np.isclose(abs(n[1]), np.sqrt(0.5), atol=1e-6) and abs(n[2]) < 1e-6
    assert np.isclose(pen.max(), 0.02 * np.sqrt(0.5), atol=1e-5)
    kind, best, sep_n, want_pts, want_pen = predict(he, p1, r, he, p2, r, n, 0.0, tol)
    assert kind == "face" and abs(sep_n - best) < 50 * tol and len(want_pts) == k and same_point_set(pts, want_pts, 200 * tol)
    assert np.allclose(sorted(pts[:, 2]), sorted(want_pts[:, 2]), atol=1e-5) and np.isclose(pts[:, 2].min(), -0.1, atol=1e-5) and np.isclose(pts[:, 2].max(), 0.5, atol=1e-5)


def test_feature_ids_are_stable_under_small_motion_and_swap_with_the_shapes(world):
    """match_contacts (contact_types/mod.rs:426-472) carries warm-start impulses over by feature id: ids must not change while the same
    features stay in contact, must differ between the points of one manifold, and must swap sides when the colliders are swapped."""
    # (box 2's bottom face lies well inside box 1's top face: the patch is that face, before and after the small motion)
    base = dict(he1=[.5, .5, .5], p1=[0, 0, 0], r1=I, he2=[.2, .5, .25], r2=quat_axis_angle([0, 1, 0], 0.3))
    k0, _, pts0, _, f1a, f2a = manifold(world, p2=[0.05, 0.99, -0.02], **base)
    k1, _, pts1, _, f1b, f2b = manifold(world, p2=[0.06, 0.985, -0.025], **base)
    assert k0 == k1 == 4
    ids0, ids1 = list(zip(f1a.tolist(), f2a.tolist())), list(zip(f1b.tolist(), f2b.tolist()))
    assert len(set(ids0)) == k0, "feature id pairs are unique inside a manifold"
    for i, q in enumerate(pts0):   # the point that stays (nearly) in place keeps its ids
        j = int(np.argmin(np.linalg.norm(pts1 - q, axis=1)))
        assert ids0[i] == ids1[j]
    o = world.contact_manifolds([0], [base["he2"]], [[0.05, 0.99, -0.02]], [base["r2"]], [0], [base["he1"]], [[0, 0, 0]], [I], [0.0])
    ks = int(o["point_count"][0])
    swapped = set(zip(o["feature_id2"][0, :ks].tolist(), o["feature_id1"][0, :ks].tolist()))
    assert ks == k0 and swapped == set(ids0)


def test_ball_against_cuboid_regions(world):
    he = [1.0, 0.5, 2.0]
    cases = [([0.3, 0.9, -0.5], [0, 1, 0], 0.1),                                   # over a face
             ([1.3, 0.8, 0.0], np.array([0.3, 0.3, 0]) / np.hypot(0.3, 0.3), 0.5 - np.hypot(0.3, 0.3)),   # nearest feature: an edge
             ([1.2, 0.7, 2.2], np.array([0.2, 0.2, 0.2]) / np.sqrt(0.12), 0.5 - np.sqrt(0.12))]            # a corner
    for centre, nrm, pen in cases:
        k, n, pts, p, _, _ = manifold(world, he, [0, 0, 0], I, [.5, 0, 0], centre, I, s2=1)
        assert k == 1 and np.allclose(n, nrm, atol=1e-5) and np.allclose(p[0], pen, atol=1e-5)
        surf = np.clip(centre, -np.asarray(he), he)   # closest point of the box
        assert np.allclose(pts[0], surf - np.asarray(nrm) * pen / 2, atol=1e-5), "contact = midpoint between the two surface points"
    assert manifold(world, he, [0, 0, 0], I, [.5, 0, 0], [0.2, 0.1, 0.3], I, s2=1)[0] == 0, "centre inside the solid cuboid: no contact (parry's projection)"
    assert manifold(world, he, [0, 0, 0], I, [.5, 0, 0], [0.3, 1.2, -0.5], I, s2=1)[0] == 0, "0.2 apart, prediction 0"
    assert manifold(world, he, [0, 0, 0], I, [.5, 0, 0], [0.3, 1.2, -0.5], I, pred=0.25, s2=1)[0] == 1
