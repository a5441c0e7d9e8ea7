"""AccumulatedLocalAcceleration (avn_local_accelerations_upload; reference dynamics/rigid_body/forces/plugin.rs:207-241): inputs shared by the CPU test
(oracle against an independent float64 derivation) and the GPU test (HIP against the oracle, bit for bit)."""
from __future__ import annotations

import numpy as np

from helpers import F


def random_local_accelerations(seed, n, fraction=0.6):
    rng = np.random.default_rng(seed)
    lin = np.where(rng.random((n, 1)) < fraction, rng.normal(scale=4.0, size=(n, 3)), 0.0)
    ang = np.where(rng.random((n, 1)) < fraction, rng.normal(scale=2.0, size=(n, 3)), 0.0)
    return lin, ang


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qrot(q, v):
    u = q[:3]; w = q[3]
    return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)


def free_flight_float64(pos, rot, lin, ang, acc_lin, acc_ang, locked, dt, substeps, steps):
    """One free body with isotropic inertia, no gravity, no damping, nothing to touch: the substep loop written from the reference's text in float64 --
    apply_local_acceleration (rotation = delta_rotation * Rotation; v += locked(rotation * a) * h), integrate_positions (delta_position += v h,
    delta_rotation = from_scaled_axis(omega h) * delta_rotation), writeback (Position += delta_position about the centre of mass = the origin here,
    Rotation = normalize(delta_rotation * Rotation))."""
    h = dt / substeps
    pos = np.array(pos, float); rot = np.array(rot, float); v = np.array(lin, float); om = np.array(ang, float)

    def lock(x):
        x = x.copy()
        if locked & 0x20: x[0] = 0.0
        if locked & 0x10: x[1] = 0.0
        if locked & 0x08: x[2] = 0.0
        return x
    for _ in range(steps):
        dp = np.zeros(3); dq = np.array([0.0, 0, 0, 1])
        for _ in range(substeps):
            r = qmul(dq, rot)
            v = v + lock(qrot(r, np.array(acc_lin, float))) * h
            om = om + lock(qrot(r, np.array(acc_ang, float))) * h
            dp = dp + v * h
            th = om * h; a = np.linalg.norm(th)
            q = np.array([0.0, 0, 0, 1]) if a == 0 else np.concatenate([th / a * np.sin(a / 2), [np.cos(a / 2)]])
            dq = qmul(q, dq)
        pos = pos + dp
        rot = qmul(dq, rot); rot = rot / np.linalg.norm(rot)
    return pos, rot, v, om


def single_body_world(lib, bits, pos, rot, lin, ang, locked=0, rb_type=None, substeps=5, flags=0):
    w = F.World(lib, F.default_config(bits, substeps=substeps, dt=1.0 / 60.0, gravity=(0.0, 0.0, 0.0)))
    w.bodies_upload(position=np.array([pos]), rotation=np.array([rot]), linear_velocity=np.array([lin]), angular_velocity=np.array([ang]), inv_mass=np.array([0.5]),
                    inv_inertia_local=np.array([[2.0, 0, 0, 2.0, 0, 2.0]]), rb_type=np.array([F.RB_DYNAMIC if rb_type is None else rb_type], np.uint8),
                    locked_axes=np.array([locked], np.uint8), body_flags=np.array([flags], np.uint8))
    return w
