"""CPU: the oracle's restatement of contact_query::contact_manifolds (Ball / Cuboid pairs) — analytic cases and
geometric invariants.  The parry3d part of the algorithm is parity-unpinned (see oracle/avo_narrow.hpp)."""
import numpy as np
import pytest

from helpers import F, oracle_lib
from narrow_checks import check_invariants, check_swap_symmetry
from narrow_scenes import quat_axis_angle, random_pairs

I = [0.0, 0.0, 0.0, 1.0]


@pytest.fixture(scope="module", params=[32, 64])
def world(request):
    return F.World(oracle_lib(), F.default_config(request.param))


def one(world, s1, he1, p1, r1, s2, he2, p2, r2, pred=0.0):
    o = world.contact_manifolds([s1], [he1], [p1], [r1], [s2], [he2], [p2], [r2], [pred])
    k = int(o["point_count"][0])
    return k, o["normal"][0], o["anchor1"][0, :k], o["anchor2"][0, :k], o["penetration"][0, :k], o["feature_id1"][0, :k], o["feature_id2"][0, :k]


def test_stacked_cubes_face_contact(world):
    k, n, a1, a2, pen, f1, f2 = one(world, 0, [.5, .5, .5], [0, 0, 0], I, 0, [.5, .5, .5], [0.3, 0.99, 0.2], I)
    assert k == 4 and np.allclose(n, [0, 1, 0]) and np.allclose(pen, 0.01, atol=1e-6)
    # the contact patch is the overlap rectangle [-0.2, 0.5] x [-0.3, 0.5] of the two faces, midway between them
    assert np.allclose(sorted(a1[:, 0]), [-0.2, -0.2, 0.5, 0.5], atol=1e-6) and np.allclose(sorted(a1[:, 2]), [-0.3, -0.3, 0.5, 0.5], atol=1e-6)
    assert np.allclose(a1[:, 1], 0.495, atol=1e-6) and np.allclose(a2[:, 1], -0.495, atol=1e-6)
    assert len(set(zip(f1.tolist(), f2.tolist()))) == 4, "feature id pairs identify the four contacts"


def test_rotated_cube_on_cube_is_an_octagon(world):
    k, n, a1, _, pen, _, _ = one(world, 0, [.5, .5, .5], [0, 0, 0], I, 0, [.5, .5, .5], [0, 0.99, 0], quat_axis_angle([0, 1, 0], np.pi / 4))
    assert k == 8 and np.allclose(n, [0, 1, 0], atol=1e-6) and np.allclose(pen, 0.01, atol=1e-6)
    r = np.hypot(a1[:, 0], a1[:, 2])
    assert np.allclose(r, r[0], atol=1e-5), "two squares at 45 degrees intersect in a regular octagon"


def test_ball_ball_and_ball_cuboid(world):
    k, n, a1, a2, pen, f1, f2 = one(world, 1, [.5, 0, 0], [0, 0, 0], I, 1, [.25, 0, 0], [0.7, 0, 0], I)
    assert k == 1 and np.allclose(n, [1, 0, 0]) and np.allclose(pen, 0.05, atol=1e-6) and np.allclose(a1[0], [0.475, 0, 0], atol=1e-6)
    k, n, a1, a2, pen, f1, f2 = one(world, 0, [1, .5, 2], [0, 0, 0], I, 1, [.5, 0, 0], [0.2, 0.9, -0.3], I)
    assert k == 1 and np.allclose(n, [0, 1, 0]) and np.allclose(pen, 0.1, atol=1e-6) and np.allclose(a1[0], [0.2, 0.45, -0.3], atol=1e-6)
    assert f1[0] == 0 and f2[0] == (3 << 30), "cuboid side UNKNOWN, ball side face(0)"
    # ball centre inside the (solid) cuboid: parry's projection returns the point itself -> no contact
    assert one(world, 0, [1, 1, 1], [0, 0, 0], I, 1, [.5, 0, 0], [0.1, 0.2, 0.0], I)[0] == 0


def test_prediction_distance_gates_the_manifold(world):
    args = (0, [.5, .5, .5], [0, 0, 0], I, 0, [.5, .5, .5], [0, 1.05, 0], I)
    assert one(world, *args, pred=0.0)[0] == 0
    k, n, _, _, pen, _, _ = one(world, *args, pred=0.1)
    assert k == 8 and np.allclose(pen, -0.05, atol=1e-6)   # coincident faces: both vertex loops report all four corners


def test_edge_edge_best_axis(world):
    # cube 1 rolled 45 deg about z (an edge points up), cube 2 rolled 45 deg about x (an edge points down), crossing edges
    k, n, a1, a2, pen, _, _ = one(world, 0, [.5, .5, .5], [0, 0, 0], quat_axis_angle([0, 0, 1], np.pi / 4), 0, [.5, .5, .5], [0, 1.38, 0],
                                  quat_axis_angle([1, 0, 0], np.pi / 4))
    assert k >= 1 and np.allclose(n, [0, 1, 0], atol=1e-5)
    assert np.isclose(pen.max(), 2 * 0.5 * np.sqrt(2) - 1.38, atol=1e-5), "deepest point = overlap of the two crossing edges"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_pairs_invariants(world, seed):
    pairs = random_pairs(seed, 1500)
    out = world.contact_manifolds(**pairs)
    tol = 2e-5 if world.dtype == np.float32 else 1e-10
    n_manifolds = check_invariants(pairs, out, tol)
    assert n_manifolds > 300
    check_swap_symmetry(world, pairs, out, tol)
    # every shape combination produced manifolds
    for a in (0, 1):
        for b in (0, 1):
            assert np.any((out["point_count"] > 0) & (pairs["shape1"] == a) & (pairs["shape2"] == b))
