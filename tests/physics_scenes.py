"""Tiny scenes with answers that come from mechanics, not from the oracle: used by tests/test_physics_sanity.py (CPU, oracle backend) and
tests/test_gpu_physics_sanity.py (HIP backend).  Every scene runs the closed loop of the library (avn_pipeline_enable: broad phase ->
narrow phase -> contact bookkeeping -> solver), i.e. the whole path."""
from __future__ import annotations

import math

import numpy as np

from helpers import F

G = 9.81
DT = 1.0 / 60.0


def cuboid_props(hx, hy, hz, density=1.0):
    m = density * 8 * hx * hy * hz
    i = (m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy))
    return 1.0 / m, [1.0 / i[0], 0, 0, 1.0 / i[1], 0, 1.0 / i[2]]


def ball_props(r, density=1.0):
    m = density * 4.0 / 3.0 * math.pi * r ** 3
    i = 0.4 * m * r * r
    return 1.0 / m, [1.0 / i, 0, 0, 1.0 / i, 0, 1.0 / i]


class Scene:
    """Body 0 is always a big static slab whose top face is y = 0."""
    def __init__(self):
        self.pos = [[0.0, -10.0, 0.0]]; self.rot = [[0, 0, 0, 1.0]]; self.lin = [[0.0, 0, 0]]; self.ang = [[0.0, 0, 0]]
        self.inv_m = [0.0]; self.inv_i = [[0.0] * 6]; self.rb = [F.RB_STATIC]; self.he = [[200.0, 10.0, 200.0]]; self.shape = [0]
        self.friction = [0.5]; self.restitution = [0.0]; self.gscale = [1.0]

    def add_box(self, pos, half=(0.5, 0.5, 0.5), vel=(0, 0, 0), friction=0.5, restitution=0.0, gravity_scale=1.0):
        im, ii = cuboid_props(*half)
        self._add(pos, vel, im, ii, list(half), 0, friction, restitution, gravity_scale)
        return len(self.pos) - 1

    def add_ball(self, pos, r=0.5, vel=(0, 0, 0), friction=0.5, restitution=0.0, gravity_scale=1.0):
        im, ii = ball_props(r)
        self._add(pos, vel, im, ii, [r, 0.0, 0.0], 1, friction, restitution, gravity_scale)
        return len(self.pos) - 1

    def _add(self, pos, vel, im, ii, he, shape, fr, re, gs):
        self.pos.append(list(map(float, pos))); self.rot.append([0, 0, 0, 1.0]); self.lin.append(list(map(float, vel))); self.ang.append([0.0, 0, 0])
        self.inv_m.append(im); self.inv_i.append(ii); self.rb.append(F.RB_DYNAMIC); self.he.append(he); self.shape.append(shape)
        self.friction.append(fr); self.restitution.append(re); self.gscale.append(gs)

    def world(self, lib, bits=32, substeps=6, gravity=(0.0, -G, 0.0), joints=None):
        n = len(self.pos)
        w = F.World(lib, F.default_config(bits, substeps=substeps, dt=DT, gravity=gravity))
        w.bodies_upload(position=np.array(self.pos), rotation=np.array(self.rot), linear_velocity=np.array(self.lin), angular_velocity=np.array(self.ang),
                        inv_mass=np.array(self.inv_m), inv_inertia_local=np.array(self.inv_i), rb_type=np.array(self.rb, np.uint8),
                        gravity_scale=np.array(self.gscale))
        w.colliders_upload(entity_index=np.arange(n, dtype=np.uint32), body=np.arange(n, dtype=np.int32), shape=np.array(self.shape, np.uint8), half_extents=np.array(self.he))
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.collider_materials_upload(friction=np.array(self.friction), restitution=np.array(self.restitution))
        if joints is not None:
            w.distance_joints_upload(**joints)
        w.pipeline_enable()
        return w


def run(w, steps, every=None):
    out = []
    for s in range(steps):
        w.step()
        if every is not None and (s + 1) % every == 0:
            out.append(w.bodies_download())
    w.synchronize()
    return out if every is not None else w.bodies_download()


# ---- the checks (each takes the backend library and the scalar width) -----------------------------------------------------------------

def check_free_fall(lib, bits):
    """No contact: semi-implicit Euler with S substeps per step is exact arithmetic: v_N = -g h N, y_N = y0 - g h^2 N (N + 1) / 2."""
    sc = Scene(); b = sc.add_box((0, 50.0, 0))
    S, steps = 4, 60
    w = sc.world(lib, bits, substeps=S)
    out = run(w, steps)
    h = DT / S; N = steps * S
    assert abs(out["linear_velocity"][b, 1] - (-G * h * N)) < 2e-3
    assert abs(out["position"][b, 1] - (50.0 - G * h * h * N * (N + 1) / 2)) < 2e-3


def check_box_comes_to_rest(lib, bits):
    sc = Scene(); b = sc.add_box((0, 0.6, 0))
    w = sc.world(lib, bits)
    out = run(w, 180)
    assert abs(out["position"][b, 1] - 0.5) < 0.01, "rests on the slab: centre half a box above it (soft contact: within 1 cm)"
    assert np.abs(out["linear_velocity"][b]).max() < 0.02 and np.abs(out["angular_velocity"][b]).max() < 0.02
    assert abs(out["position"][b, 0]) < 1e-3 and abs(out["position"][b, 2]) < 1e-3


def check_friction_cone(lib, bits):
    """Tilted gravity instead of a tilted slab: mu = 0.5 holds at 20 degrees (tan = 0.36), lets go at 35 (tan = 0.70) with
    a = g (sin - mu cos) along the slope."""
    for deg, slides in ((20.0, False), (35.0, True)):
        th = math.radians(deg)
        sc = Scene(); b = sc.add_box((0, 0.5, 0), friction=0.5)
        sc.friction[0] = 0.5
        w = sc.world(lib, bits, gravity=(G * math.sin(th), -G * math.cos(th), 0.0))
        t = 1.5
        out = run(w, int(round(t / DT)))
        x = float(out["position"][b, 0])
        if slides:
            a = G * (math.sin(th) - 0.5 * math.cos(th))
            assert abs(x - 0.5 * a * t * t) < 0.08 * 0.5 * a * t * t, f"slides {x} m, expected {0.5 * a * t * t}"
        else:
            assert abs(x) < 0.01 and abs(out["linear_velocity"][b, 0]) < 0.02, f"held by friction, moved {x} m"


def check_restitution(lib, bits):
    """A ball dropped from 2 m on a slab, both with restitution 0.8 (Average -> 0.8): the rebound apex is e^2 times the drop height."""
    sc = Scene(); r = 0.25
    b = sc.add_ball((0, 2.0 + r, 0), r=r, restitution=0.8)
    sc.restitution[0] = 0.8
    w = sc.world(lib, bits, substeps=8)
    ys = np.array([o["position"][b, 1] for o in run(w, 150, every=1)]) - r
    hit = int(np.argmin(ys[:80]))
    apex = float(ys[hit:hit + 70].max())
    assert ys[hit] < 0.05 and abs(apex - 0.64 * 2.0) < 0.15, f"rebound apex {apex} m, expected {0.64 * 2.0}"
    sc2 = Scene(); b2 = sc2.add_ball((0, 2.0 + r, 0), r=r, restitution=0.0)
    ys2 = np.array([o["position"][b2, 1] for o in run(sc2.world(lib, bits, substeps=8), 150, every=1)]) - r
    assert float(ys2[int(np.argmin(ys2[:80])):].max()) < 0.05, "no restitution: the ball stays down"


def check_momentum_is_conserved(lib, bits):
    """Two balls, no gravity, oblique collision: contact impulses are equal and opposite, so total linear momentum does not change."""
    sc = Scene()
    a = sc.add_ball((-2.0, 5.0, 0.1), r=0.5, vel=(3.0, 0, 0), gravity_scale=0.0, restitution=0.5)
    b = sc.add_ball((2.0, 5.0, -0.1), r=0.3, vel=(-1.0, 0.2, 0), gravity_scale=0.0, restitution=0.5)
    ma, mb = 1.0 / sc.inv_m[a], 1.0 / sc.inv_m[b]
    p0 = ma * np.array(sc.lin[a]) + mb * np.array(sc.lin[b])
    w = sc.world(lib, bits)
    out = run(w, 120)
    p1 = ma * out["linear_velocity"][a].astype(np.float64) + mb * out["linear_velocity"][b].astype(np.float64)
    assert np.abs(out["linear_velocity"][a] - np.array(sc.lin[a])).max() > 0.1, "they must have collided"
    assert np.abs(p1 - p0).max() < 2e-4 * max(1.0, np.abs(p0).max())


def check_stack_stays_put(lib, bits):
    sc = Scene()
    ids = [sc.add_box((0, 0.5 + 1.0 * k, 0)) for k in range(5)]
    w = sc.world(lib, bits)
    out = run(w, 300)
    p = out["position"][ids]
    assert np.abs(p[:, 1] - (0.5 + np.arange(5))).max() < 0.03, "a resting stack of five sinks less than 3 cm in 5 s"
    assert np.abs(p[:, [0, 2]]).max() < 0.01, "and does not creep sideways"


def check_pendulum_keeps_its_length(lib, bits):
    """A ball on a rigid distance joint (compliance 0) released horizontally: the joint length holds to a millimetre, and it swings through."""
    sc = Scene()
    pivot = sc.add_ball((0, 5.0, 0), r=0.1); sc.rb[pivot] = F.RB_STATIC; sc.inv_m[pivot] = 0.0; sc.inv_i[pivot] = [0.0] * 6
    bob = sc.add_ball((2.0, 5.0, 0), r=0.1)
    J = dict(body1=np.array([pivot], np.int32), body2=np.array([bob], np.int32), local_anchor1=np.zeros((1, 3)), local_anchor2=np.zeros((1, 3)),
             limit_min=np.array([2.0]), limit_max=np.array([2.0]), compliance=np.array([0.0]))
    w = sc.world(lib, bits, substeps=8, joints=J)
    outs = run(w, 90, every=1)
    d = np.array([np.linalg.norm(o["position"][bob].astype(np.float64) - np.array([0, 5.0, 0])) for o in outs])
    assert np.abs(d - 2.0).max() < 2e-3, f"joint length drifts by {np.abs(d - 2.0).max()}"
    ymin = min(o["position"][bob, 1] for o in outs)
    assert ymin < 3.1, "the bob swings down to the bottom of its arc"


def _qrot(q, v):
    q = np.asarray(q, np.float64); v = np.asarray(v, np.float64)
    u, w = q[:3], q[3]
    return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)


def _joint_world(lib, bits, jtype, vel=(0, 0, 0), ang=(0, 0, 0), substeps=8):
    """A static anchor body at (0, 10, 0) and a dynamic 0.5 m box whose centre starts 1 m along +x, joined at the anchor body's centre."""
    sc = Scene()
    a = sc.add_ball((0, 10.0, 0), r=0.1); sc.rb[a] = F.RB_STATIC; sc.inv_m[a] = 0.0; sc.inv_i[a] = [0.0] * 6
    b = sc.add_box((1.0, 10.0, 0), half=(0.25, 0.25, 0.25), vel=vel)
    sc.ang[b] = list(map(float, ang))
    w = sc.world(lib, bits, substeps=substeps)
    w.joints_upload(joint_type=np.array([jtype], np.uint8), body1=np.array([a], np.int32), body2=np.array([b], np.int32),
                    local_anchor1=np.zeros((1, 3)), local_anchor2=np.array([[-1.0, 0, 0]]), compliance=np.zeros((1, 3)))
    return w, a, b


def _anchor_gap(o, a, b):
    pa = o["position"][a].astype(np.float64)
    pb = o["position"][b].astype(np.float64) + _qrot(o["rotation"][b], [-1.0, 0, 0])
    return float(np.linalg.norm(pa - pb))


def check_fixed_joint_holds_the_pose(lib, bits):
    """FixedJoint, compliance 0, to a static body: the box neither falls nor turns, whatever it was doing before."""
    w, a, b = _joint_world(lib, bits, F.JOINT_FIXED, vel=(0.5, -0.5, 0.3), ang=(1.0, 2.0, -1.0))
    outs = run(w, 120, every=1)
    assert max(_anchor_gap(o, a, b) for o in outs[10:]) < 5e-3
    q = outs[-1]["rotation"][b].astype(np.float64)
    assert 2.0 * math.acos(min(1.0, abs(q[3]))) < 2e-2, "relative rotation stays the initial one (identity)"
    assert np.abs(outs[-1]["position"][b] - np.array([1.0, 10.0, 0.0])).max() < 5e-3


def check_revolute_joint_is_a_hinge(lib, bits):
    """RevoluteJoint about Z: the arm swings down in the XY plane, its own Z axis never leaves the world's, the pivot stays in the anchor."""
    w, a, b = _joint_world(lib, bits, F.JOINT_REVOLUTE, vel=(0, 0, 0.5), ang=(0.5, 0.5, 0))   # an out-of-plane kick the hinge must absorb
    outs = run(w, 90, every=1)
    assert max(_anchor_gap(o, a, b) for o in outs[10:]) < 5e-3
    for o in outs[10:]:
        z = _qrot(o["rotation"][b], [0, 0, 1.0])
        assert abs(z[2]) > 1.0 - 1e-3, "the hinge axis of the arm stays parallel to the anchor body's"
        assert abs(o["position"][b][2]) < 5e-3, "and the arm stays in the hinge plane"
    assert min(o["position"][b][1] for o in outs) < 9.2, "it swings down"


def check_spherical_joint_keeps_the_pivot(lib, bits):
    w, a, b = _joint_world(lib, bits, F.JOINT_SPHERICAL, vel=(0, 0, 2.0), ang=(1.0, 0, 0))
    outs = run(w, 90, every=1)
    assert max(_anchor_gap(o, a, b) for o in outs[10:]) < 5e-3
    assert max(abs(o["position"][b][2]) for o in outs) > 0.3, "a ball joint lets the arm leave the plane"
    assert max(abs(np.linalg.norm(o["position"][b].astype(np.float64) - np.array([0, 10.0, 0])) - 1.0) for o in outs[10:]) < 5e-3


def check_prismatic_joint_slides_along_its_axis(lib, bits):
    """PrismaticJoint along X (no limits): gravity cannot pull the slider off its rail, the push along the rail is not resisted."""
    w, a, b = _joint_world(lib, bits, F.JOINT_PRISMATIC, vel=(1.0, 0, 0))
    out = run(w, 60)
    p = out["position"][b].astype(np.float64)
    assert abs(p[0] - 2.0) < 0.02, f"x = {p[0]} after 1 s at 1 m/s from x = 1"
    assert abs(p[1] - 10.0) < 5e-3 and abs(p[2]) < 5e-3
    q = out["rotation"][b].astype(np.float64)
    assert 2.0 * math.acos(min(1.0, abs(q[3]))) < 1e-2


CHECKS = [check_fixed_joint_holds_the_pose, check_revolute_joint_is_a_hinge, check_spherical_joint_keeps_the_pivot, check_prismatic_joint_slides_along_its_axis,
          check_free_fall, check_box_comes_to_rest, check_friction_cone, check_restitution, check_momentum_is_conserved, check_stack_stays_put,
          check_pendulum_keeps_its_length]
