"""Seeded shape-pair sets for the narrow-phase tests (Ball / Cuboid pairs in every configuration class)."""
from __future__ import annotations

import numpy as np

from helpers import F, random_unit_quats


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(angle / 2), [np.cos(angle / 2)]])


def quat_rotate(q, v):
    """Rotate vectors v [n,3] by unit quaternions q [n,4] (xyzw), float64."""
    b = q[:, :3]
    w = q[:, 3:4]
    return v * (w * w - (b * b).sum(1, keepdims=True)) + b * (2 * (v * b).sum(1, keepdims=True)) + np.cross(b, v) * (2 * w)


def random_pairs(seed: int, n: int):
    """n random pairs: shapes drawn uniformly from {cuboid, ball}^2, random sizes and orientations, the second shape
    placed so that the pair is touching, slightly separated (within the prediction distance) or clearly apart."""
    rng = np.random.default_rng(seed)
    s1 = rng.integers(0, 2, n).astype(np.uint8)
    s2 = rng.integers(0, 2, n).astype(np.uint8)
    he1 = rng.uniform(0.2, 1.5, (n, 3)); he2 = rng.uniform(0.2, 1.5, (n, 3))
    he1[s1 == F.SHAPE_BALL, 1:] = 0.0; he2[s2 == F.SHAPE_BALL, 1:] = 0.0
    p1 = rng.uniform(-5, 5, (n, 3))
    r1 = random_unit_quats(rng, n); r2 = random_unit_quats(rng, n)
    # a fraction axis-aligned (exact face/face stacks with coincident edges: the degenerate clipping cases)
    aligned = rng.random(n) < 0.25
    r1[aligned] = [0, 0, 0, 1]; r2[aligned] = [0, 0, 0, 1]
    # bounding radii -> centre distance between "deep", "touching" and "apart"
    rad = lambda s, he: np.where(s == F.SHAPE_BALL, he[:, 0], np.linalg.norm(he, axis=1))
    inner = lambda s, he: np.where(s == F.SHAPE_BALL, he[:, 0], he.min(axis=1))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis_dir = rng.random(n) < 0.4   # along a coordinate axis (face contacts for the aligned ones)
    ax = rng.integers(0, 3, n); sg = rng.choice([-1.0, 1.0], n)
    d[axis_dir] = 0.0; d[axis_dir, ax[axis_dir]] = sg[axis_dir]
    lo, hi = inner(s1, he1) + inner(s2, he2), rad(s1, he1) + rad(s2, he2)
    t = rng.random(n)
    dist = lo * 0.7 + (hi * 1.15 - lo * 0.7) * t
    # aligned axis-direction cuboid pairs: put them exactly face to face with a small penetration / gap
    both_cub = (s1 == F.SHAPE_CUBOID) & (s2 == F.SHAPE_CUBOID) & aligned & axis_dir
    ext = he1[np.arange(n), ax] + he2[np.arange(n), ax]
    dist[both_cub] = ext[both_cub] + rng.uniform(-0.05, 0.05, both_cub.sum())
    p2 = p1 + d * dist[:, None]
    # lateral offsets for the face-to-face ones
    lat = rng.uniform(-0.4, 0.4, (n, 3)); lat[np.arange(n), ax] = 0.0
    p2[both_cub] += lat[both_cub]
    pred = rng.choice([0.0, 0.02, 0.1, 0.5], n)
    return dict(shape1=s1, half_extents1=he1, position1=p1, rotation1=r1, shape2=s2, half_extents2=he2, position2=p2,
                rotation2=r2, prediction_distance=pred)


def support(shape, he, rot, direction):
    """Support distance of a shape (at the origin, rotated) along `direction` [n,3] (unit), float64."""
    out = np.empty(len(shape))
    ball = shape == F.SHAPE_BALL
    out[ball] = he[ball, 0]
    c = ~ball
    if c.any():
        qc = rot[c].copy(); qc[:, :3] *= -1
        local = quat_rotate(qc, direction[c])
        out[c] = (np.abs(local) * he[c]).sum(1)
    return out
