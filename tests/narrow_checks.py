"""Checks shared by the CPU (oracle) and GPU (product) narrow-phase tests: geometric invariants of
contact_query::contact_manifolds results, valid for any correct implementation."""
from __future__ import annotations

import numpy as np

from helpers import F
from narrow_scenes import support


def check_invariants(pairs, out, tol):
    n = len(pairs["shape1"])
    cnt = out["point_count"].astype(int)
    has = cnt > 0
    nrm = out["normal"].astype(np.float64)
    assert np.all(np.abs(np.linalg.norm(nrm[has], axis=1) - 1.0) < 1e-3), "normals must be unit"
    p1, p2 = pairs["position1"].astype(np.float64), pairs["position2"].astype(np.float64)
    r1, r2 = pairs["rotation1"].astype(np.float64), pairs["rotation2"].astype(np.float64)
    s1, s2 = pairs["shape1"], pairs["shape2"]
    he1, he2 = pairs["half_extents1"].astype(np.float64), pairs["half_extents2"].astype(np.float64)
    pred = np.asarray(pairs["prediction_distance"], np.float64)
    # separation along the manifold normal (a lower bound of the true distance): >= -max penetration - tol
    sep_n = ((p2 - p1) * nrm).sum(1) - support(s1, he1, r1, nrm) - support(s2, he2, r2, -nrm)
    deep = []
    for i in np.flatnonzero(has):
        k = cnt[i]
        a1 = out["anchor1"][i, :k].astype(np.float64); a2 = out["anchor2"][i, :k].astype(np.float64)
        pen = out["penetration"][i, :k].astype(np.float64)
        # anchor2 = anchor1 + (position1 - position2); point = position1 + anchor1
        assert np.allclose(a2, a1 + (p1[i] - p2[i]), atol=tol, rtol=0)
        assert np.allclose(out["point"][i, :k], p1[i] + a1, atol=tol, rtol=0)
        # the deepest point realises the separation along the normal; nothing is deeper
        assert pen.max() <= -sep_n[i] + 50 * tol, (i, pen, sep_n[i])
        if s1[i] == F.SHAPE_BALL or s2[i] == F.SHAPE_BALL:
            assert k == 1 and abs(pen[0] + sep_n[i]) < 50 * tol
        # the point on shape 1 (midpoint + n pen/2) is on or inside shape 1's support plane along the normal, and the
        # deepest contact lies ON it and realises the separation along the normal
        h1 = (a1 * nrm[i]).sum(1) - support(s1[i:i + 1], he1[i:i + 1], r1[i:i + 1], nrm[i:i + 1])[0]
        assert np.all(h1 + pen / 2 < 50 * tol), (i, h1, pen)
        deep.append(abs(pen.max() + sep_n[i]))
        # manifolds only exist within the prediction distance
        assert sep_n[i] <= pred[i] + 50 * tol
    # a pair whose bounding spheres are further apart than the prediction distance has no manifold
    rad = lambda s, he: np.where(s == F.SHAPE_BALL, he[:, 0], np.linalg.norm(he, axis=1))
    far = np.linalg.norm(p2 - p1, axis=1) - rad(s1, he1) - rad(s2, he2) > pred + 1e-6
    assert not np.any(has & far)
    # (statistical: clipping may drop the deepest vertex in degenerate configurations, never in the bulk)
    assert np.mean(np.asarray(deep) < 100 * tol) > 0.9, np.mean(np.asarray(deep) < 100 * tol)
    return int(has.sum())


def check_swap_symmetry(world, pairs, out, tol):
    """Swapping the two colliders mirrors a ball manifold exactly (normal negated, anchors exchanged)."""
    sw = dict(shape1=pairs["shape2"], half_extents1=pairs["half_extents2"], position1=pairs["position2"], rotation1=pairs["rotation2"],
              shape2=pairs["shape1"], half_extents2=pairs["half_extents1"], position2=pairs["position1"], rotation2=pairs["rotation1"],
              prediction_distance=pairs["prediction_distance"])
    o2 = world.contact_manifolds(**sw)
    ball = (pairs["shape1"] == F.SHAPE_BALL) | (pairs["shape2"] == F.SHAPE_BALL)
    both = (out["point_count"] > 0) & (o2["point_count"] > 0) & ball
    assert np.allclose(out["normal"][both], -o2["normal"][both], atol=20 * tol)
    assert np.allclose(out["penetration"][both, 0], o2["penetration"][both, 0], atol=20 * tol)
    assert np.allclose(out["anchor1"][both, 0], o2["anchor2"][both, 0], atol=50 * tol)
    return o2
