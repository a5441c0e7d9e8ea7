"""CPU: the arithmetic argument behind k_body_warm_start_quad (avian_amd/csrc/k_contacts.hip), checked in numpy float32 / float64.

The lane-per-body warm start walks a body's manifolds in colour order and does, per point, `v = v - dv` (the body is body1) or `v = v + dv` (body2).
The quad form lets lane q of a body's four lanes own the body's q-th, (q + 4)-th ... populated colour, turns every point into a SIGNED ADDEND
(-dv | +dv, and -0.0 for a point that does not exist) and has all four lanes add the round's addends in colour order.  Same bits, because
  * x - y == x + (-y) for every pair of floats (negation is exact),
  * x + (-0.0) == x for every x, including x = -0.0, x = +0.0, infinities and NaN payload-preserving quiet NaNs.
The test replays both orders on random bodies with the special values mixed in.  (The kernel itself is compared with the oracle bit for bit in the
closed-loop GPU suites; this file pins the identities the restructuring rests on, on the CPU.)"""
import numpy as np
import pytest

SPECIAL32 = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38, 1.17549435e-38], np.float32)


def bits(a):
    a = np.asarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_subtraction_is_addition_of_the_negation_and_minus_zero_is_the_additive_identity(dtype):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000).astype(dtype) * dtype(10.0) ** rng.integers(-30, 30, 4000).astype(dtype), SPECIAL32.astype(dtype)])
    y = np.concatenate([rng.standard_normal(4000).astype(dtype) * dtype(10.0) ** rng.integers(-30, 30, 4000).astype(dtype), SPECIAL32[::-1].astype(dtype)])
    X, Y = np.meshgrid(x[:600], y[:600])
    X = np.concatenate([X.ravel(), np.repeat(SPECIAL32.astype(dtype), len(SPECIAL32))]); Y = np.concatenate([Y.ravel(), np.tile(SPECIAL32.astype(dtype), len(SPECIAL32))])
    with np.errstate(invalid="ignore", over="ignore"):
        a, b = X - Y, X + (-Y)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    assert same.all()
    with np.errstate(invalid="ignore"):
        c = x + dtype(-0.0)
    assert (bits(c) == bits(x)).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quad_rounds_add_the_same_sequence_as_the_colour_walk(dtype):
    rng = np.random.default_rng(11)
    n_colors, max_points = 23, 4
    for trial in range(300):
        mask = rng.random(n_colors) < rng.choice([0.1, 0.2, 0.5, 0.9])
        side = rng.integers(0, 2, n_colors)
        npts = rng.integers(0, max_points + 1, n_colors)            # 0: a manifold whose constraint is absent (np == 0)
        dv = (rng.standard_normal((n_colors, max_points, 6)) * 10.0 ** rng.integers(-6, 3, (n_colors, max_points, 1))).astype(dtype)
        dv[rng.random(dv.shape) < 0.05] = dtype(0.0); dv[rng.random(dv.shape) < 0.02] = dtype(-0.0)
        v0 = (rng.standard_normal(6) * 10.0 ** rng.integers(-3, 3)).astype(dtype)
        if trial % 7 == 0:
            v0[rng.integers(0, 6)] = dtype(-0.0)
        # the lane-per-body form: colours in order, points in order, subtract (body1) or add (body2)
        v = v0.copy()
        for c in range(n_colors):
            if not mask[c]:
                continue
            for k in range(npts[c]):
                v = (v + dv[c, k]) if side[c] else (v - dv[c, k])
        # the quad form: the j-th populated colour belongs to lane j % 4 in round j // 4; every lane adds the round's four x four addends in lane, point order
        populated = np.flatnonzero(mask)
        u = v0.copy()
        for r0 in range(0, len(populated), 4):
            addends = np.full((4, max_points, 6), dtype(-0.0), dtype)
            for lane, c in enumerate(populated[r0:r0 + 4]):
                for k in range(max_points):
                    if k < npts[c]:
                        addends[lane, k] = dv[c, k] if side[c] else -dv[c, k]
            for lane in range(4):
                for k in range(max_points):
                    u = u + addends[lane, k]
        assert (bits(u) == bits(v)).all(), (trial, u, v)
