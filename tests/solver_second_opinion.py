"""A THIRD restatement of the contact solver and the XPBD distance joint, written from the reference's text
only -- /root/reference/src/dynamics/solver/{plugin.rs, contact/mod.rs, contact/normal_part.rs, contact/tangent_part.rs,
softness_parameters/mod.rs, xpbd/mod.rs, xpbd/positional_constraint.rs, xpbd/joints/distance.rs}, dynamics/joints/mod.rs -- without
reading oracle/ or the kernels: plain Python loops over numpy SCALARS of the world's type, so that every operation rounds once, in the
order the Rust source writes it.  tests/test_solver_second_opinion.py runs it next to the oracle system by system (through the C ABI:
the oracle's state before a system goes in, its state after the system must come out, bit for bit).

What is taken from glam 0.30.8 (un-vendored) rather than from the reference: `Vec3::dot` = (x x' + y y') + z z',
`Vec3::cross`, `Quat * Vec3` = v (w w - b.b) + b (2 (v.b)) + (b x v) (2 w), `Vec2::clamp_length_max` = max * (v / |v|),
`SymmetricMat3 * Vec3` = (m.0 x + m.1 y) + m.2 z per row, `Vec3::try_normalize`, `any_orthonormal_vector`, `from_scaled_axis`, and the
quaternion product: f32 pairs its four products (a + b) + (c + d) (the SSE2 form), f64 sums left to right.
"""
import ctypes
import math

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
for _n in ("sinf", "cosf"):
    getattr(_libm, _n).restype = ctypes.c_float; getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("sin", "cos"):
    getattr(_libm, _n).restype = ctypes.c_double; getattr(_libm, _n).argtypes = [ctypes.c_double]


class Arith:
    """Scalar type + the vector helpers, every one of them written out operation by operation."""

    def __init__(self, bits):
        self.T = np.float32 if bits == 32 else np.float64
        self.bits = bits
        self.eps = self.T(np.finfo(self.T).eps)
        self.zero, self.one, self.two = self.T(0), self.T(1), self.T(2)

    def v(self, a):
        return (self.T(a[0]), self.T(a[1]), self.T(a[2]))

    def sin(self, x):
        return self.T(_libm.sinf(float(x))) if self.bits == 32 else self.T(_libm.sin(float(x)))

    def cos(self, x):
        return self.T(_libm.cosf(float(x))) if self.bits == 32 else self.T(_libm.cos(float(x)))

    # ---- Vec3 -----------------------------------------------------------------------------------------
    @staticmethod
    def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
    @staticmethod
    def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
    @staticmethod
    def neg(a): return (-a[0], -a[1], -a[2])
    @staticmethod
    def scale(a, s): return (a[0] * s, a[1] * s, a[2] * s)
    @staticmethod
    def cmul(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
    @staticmethod
    def dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
    @staticmethod
    def cross(a, b): return (a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])

    def sym_mul(self, m, v):
        """SymmetricMat3 {m00, m01, m02, m11, m12, m22} times a vector."""
        m00, m01, m02, m11, m12, m22 = m
        return ((m00 * v[0] + m01 * v[1]) + m02 * v[2], (m01 * v[0] + m11 * v[1]) + m12 * v[2], (m02 * v[0] + m12 * v[1]) + m22 * v[2])

    def recip_or_zero(self, x):   # math/mod.rs:248-257
        return self.one / x if (x != 0 and np.isfinite(x)) else self.zero

    def try_normalize(self, a):
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            r = self.one / np.sqrt(self.dot(a, a))
        return self.scale(a, r) if (np.isfinite(r) and r > 0) else None

    def any_orthonormal_vector(self, a):
        sign = self.T(-1) if np.signbit(a[2]) else self.one
        p = self.T(-1) / (sign + a[2])
        q = a[0] * a[1] * p
        return (q, sign + a[1] * a[1] * p, -a[1])

    # ---- quaternions (x, y, z, w) ---------------------------------------------------------------------
    def qrot(self, q, v):
        b = (q[0], q[1], q[2]); w = q[3]
        b2 = self.dot(b, b)
        t1 = self.scale(v, w * w - b2)
        t2 = self.scale(b, self.dot(v, b) * self.two)
        t3 = self.scale(self.cross(b, v), w * self.two)
        return self.add(self.add(t1, t2), t3)

    def qmul(self, l, r):
        lx, ly, lz, lw = l; rx, ry, rz, rw = r
        if self.bits == 32:   # SSE2: (a + b) + (c + d)
            return ((lx * rw + lw * rx) + (ly * rz - lz * ry), (ly * rw + lz * rx) + (lw * ry - lx * rz),
                    (lz * rw - ly * rx) + (lx * ry + lw * rz), (lw * rw - lx * rx) + (-(ly * ry) - lz * rz))
        return (lw * rx + lx * rw + ly * rz - lz * ry, lw * ry - lx * rz + ly * rw + lz * rx,
                lw * rz + lx * ry - ly * rx + lz * rw, lw * rw - lx * rx - ly * ry - lz * rz)

    def from_scaled_axis(self, v):
        length = np.sqrt(self.dot(v, v))
        if length == 0:
            return (self.zero, self.zero, self.zero, self.one)
        axis = (v[0] / length, v[1] / length, v[2] / length)   # from_axis_angle(v / length, length)
        half = length * self.T(0.5)
        s, c = self.sin(half), self.cos(half)
        return (axis[0] * s, axis[1] * s, axis[2] * s, c)


# ---- time and softness (solver/schedule.rs:194-200, plugin.rs:326-350, softness_parameters/mod.rs:36-79) -----------------------------
def time_scalars(A, dt_ns, substeps):
    """Duration arithmetic: the substep Duration is dt.div_f64(substeps); scalars are taken from the Durations both ways."""
    secs64 = lambda ns: float(ns // 1_000_000_000) + float(ns % 1_000_000_000) / 1e9
    h_ns = int(round(secs64(dt_ns) / float(substeps) * 1e9))
    T = A.T
    def adjusted(ns):   # delta_seconds_adjusted = Duration::as_secs_f32 in f32 builds, as_secs_f64 in f64 builds
        if A.bits == 64: return T(secs64(ns))
        return np.float32(np.float32(ns // 1_000_000_000) + np.float32(ns % 1_000_000_000) / np.float32(1e9))
    return dict(dt=T(secs64(dt_ns)), h=T(secs64(h_ns)), h_adj=adjusted(h_ns), dt_adj=adjusted(dt_ns))


def softness_coefficients(A, damping_ratio, hz, delta_secs):
    T = A.T
    TAU = T(6.283185307179586476925286766559)
    ddr = A.two * T(damping_ratio)
    omega = TAU * hz
    a1 = ddr + omega * delta_secs
    a2 = omega * delta_secs * a1
    a3 = A.one / (A.one + a2)
    return dict(bias=omega / a1, impulse_scale=a3, mass_scale=a2 * a3)


def contact_softness(A, cfg, ts):
    T = A.T
    dt, h = ts["dt"], ts["h"]
    max_hz = A.one / (dt * A.two)
    hz = T(cfg.contact_frequency_factor) * min(max_hz, T(0.25) / h)
    return dict(dynamic=softness_coefficients(A, cfg.contact_damping_ratio, hz, h),
                non_dynamic=softness_coefficients(A, cfg.contact_damping_ratio, A.two * hz, h))


# ---- bodies as the solver sees them ---------------------------------------------------------------------------------------------------
NO_SOLVER_BODY = 1 << 31


class Bodies:
    """SolverBody + SolverBodyInertia of every body, from avn_solver_bodies_download of the state BEFORE the system under test."""

    def __init__(self, A, sb):
        self.A = A
        n = len(sb["inv_mass"])
        self.lin = [A.v(sb["linear_velocity"][i]) for i in range(n)]
        self.ang = [A.v(sb["angular_velocity"][i]) for i in range(n)]
        self.dp = [A.v(sb["delta_position"][i]) for i in range(n)]
        self.dq = [tuple(A.T(x) for x in sb["delta_rotation"][i]) for i in range(n)]
        self.flags = [int(f) for f in sb["flags"]]
        self.inv_mass = [A.T(x) for x in sb["inv_mass"]]
        self.inv_I = [tuple(A.T(x) for x in sb["inv_inertia_world"][i]) for i in range(n)]
        self.dominance = [int(d) for d in sb["dominance"]]

    def has_solver_body(self, i):
        return not (self.flags[i] & NO_SOLVER_BODY)

    def effective_inv_mass(self, i):   # solver_body/mod.rs:437-451
        A = self.A
        m = self.inv_mass[i]; f = self.flags[i]
        return (A.zero if f & 0b100000 else m, A.zero if f & 0b010000 else m, A.zero if f & 0b001000 else m)

    def inertia(self, i, dummy):
        """(effective_inv_mass, effective_inv_angular_inertia), or SolverBodyInertia::DUMMY's zeros."""
        A = self.A
        if dummy or not self.has_solver_body(i):
            return (A.zero,) * 3, (A.zero,) * 6
        return self.effective_inv_mass(i), self.inv_I[i]

    def dominance_of(self, i):   # DUMMY: dominance i8::MAX as i16 + 1 = 128 (solver_body/mod.rs:265-276)
        return self.dominance[i] if self.has_solver_body(i) else 128


def velocity_at_point(A, lin, ang, p):
    return A.add(lin, A.cross(ang, p))


# ---- ContactConstraint::generate (contact/mod.rs:110-220, normal_part.rs:39-112, tangent_part.rs:35-151) -----------------------------------
def generate(A, bodies, body_lin_vel, mf, m, friction, restitution, warm_n, warm_t, warm_start_enabled, rb_type, body_flags):
    """One manifold -> a constraint dict, or None when prepare_contact_constraints skips it (plugin.rs:400-438)."""
    T = A.T
    b1, b2 = int(mf["body1"][m]), int(mf["body2"][m])
    np_ = int(mf["point_count"][m])
    generates = bool(mf["manifold_flags"][m] & 1) if "manifold_flags" in mf else True
    SLEEPING, DISABLED, DYNAMIC = 1, 2, 0
    if not generates: return None
    if (body_flags[b1] & (DISABLED | SLEEPING)) or (body_flags[b2] & (DISABLED | SLEEPING)): return None   # Query<BodyQuery, RigidBodyActiveFilter>::get fails
    if rb_type[b1] != DYNAMIC and rb_type[b2] != DYNAMIC: return None                   # neither body is dynamic
    if np_ == 0: return None                                                            # no points: the constraint is dropped
    rel_dom = bodies.dominance_of(b1) - bodies.dominance_of(b2)
    im1, i1 = bodies.inertia(b1, rel_dom > 0)
    im2, i2 = bodies.inertia(b2, rel_dom < 0)
    w_sum = A.add(im1, im2)
    normal = A.v(mf["normal"][m])
    fric = T(friction)
    # compute_tangent_directions from the LinearVelocity components
    force_direction = A.neg(normal)
    relative_velocity = A.sub(A.v(body_lin_vel[b1]), A.v(body_lin_vel[b2]))
    tangent_velocity = A.sub(relative_velocity, A.scale(force_direction, A.dot(force_direction, relative_velocity)))
    tangent = A.try_normalize(tangent_velocity)
    if tangent is None: tangent = A.any_orthonormal_vector(force_direction)
    bitangent = A.cross(force_direction, tangent)
    points = []
    for k in range(np_):
        r1, r2 = A.v(mf["anchor1"][m, k]), A.v(mf["anchor2"][m, k])
        r1n, r2n = A.cross(r1, normal), A.cross(r2, normal)
        k_linear = A.dot(normal, A.cmul(w_sum, normal))
        kk = k_linear + A.dot(r1n, A.sym_mul(i1, r1n)) + A.dot(r2n, A.sym_mul(i2, r2n))
        p = dict(anchor1=r1, anchor2=r2, impulse=T(warm_n[m, k]) if warm_start_enabled else A.zero, total_impulse=A.zero,
                 effective_mass=A.recip_or_zero(kk), normal_speed=T(mf["normal_speed"][m, k]), tangent=None,
                 initial_separation=-T(mf["penetration"][m, k]) - A.dot(A.sub(r2, r1), normal))
        if fric > 0:
            rt11, rt12, rt21, rt22 = A.cross(r1, tangent), A.cross(r2, tangent), A.cross(r1, bitangent), A.cross(r2, bitangent)
            i1_rt11, i2_rt12, i1_rt21, i2_rt22 = A.sym_mul(i1, rt11), A.sym_mul(i2, rt12), A.sym_mul(i1, rt21), A.sym_mul(i2, rt22)
            k_linear1 = A.dot(tangent, A.cmul(w_sum, tangent))
            k_linear2 = A.dot(bitangent, A.cmul(w_sum, bitangent))
            k1 = k_linear1 + A.dot(rt11, i1_rt11) + A.dot(rt12, i2_rt12)
            k2 = k_linear2 + A.dot(rt21, i1_rt21) + A.dot(rt22, i2_rt22)
            k3 = A.two * (A.dot(rt11, i1_rt21) + A.dot(rt12, i2_rt22))
            p["tangent"] = dict(impulse=(T(warm_t[m, k, 0]), T(warm_t[m, k, 1])) if warm_start_enabled else (A.zero, A.zero), k=(k1, k2, k3))
        points.append(p)
    tv = A.v(mf["tangent_velocity"][m]) if "tangent_velocity" in mf else (A.zero,) * 3
    return dict(body1=b1, body2=b2, relative_dominance=rel_dom, friction=fric, restitution=T(restitution), tangent_velocity=tv,
                normal=normal, tangent1=tangent, points=points, non_dynamic_softness=rel_dom != 0)


def _bodies_of(A, bodies, c):
    """warm_start_internal / solve_contacts_internal: the two SolverBodies (a fresh DUMMY for a missing one) and their inertias."""
    out = []
    for i, dummy_inertia in ((c["body1"], c["relative_dominance"] > 0), (c["body2"], c["relative_dominance"] < 0)):
        has = bodies.has_solver_body(i)
        lin = bodies.lin[i] if has else (A.zero,) * 3
        ang = bodies.ang[i] if has else (A.zero,) * 3
        dp = bodies.dp[i] if has else (A.zero,) * 3
        dq = bodies.dq[i] if has else (A.zero, A.zero, A.zero, A.one)
        im, I = bodies.inertia(i, dummy_inertia)
        out.append(dict(i=i, has=has, lin=lin, ang=ang, dp=dp, dq=dq, im=im, I=I))
    return out


def _write_back(bodies, bs):
    for b in bs:
        if b["has"]:
            bodies.lin[b["i"]] = b["lin"]; bodies.ang[b["i"]] = b["ang"]


def _apply(A, b1, b2, impulse, r1, r2):
    b1["lin"] = A.sub(b1["lin"], A.cmul(impulse, b1["im"]))
    b1["ang"] = A.sub(b1["ang"], A.sym_mul(b1["I"], A.cross(r1, impulse)))
    b2["lin"] = A.add(b2["lin"], A.cmul(impulse, b2["im"]))
    b2["ang"] = A.add(b2["ang"], A.sym_mul(b2["I"], A.cross(r2, impulse)))


def tangent_directions(A, c):
    return c["tangent1"], A.cross(c["tangent1"], c["normal"])


# ---- ContactConstraint::warm_start (contact/mod.rs:223-264) ---------------------------------------------------------------------------------
def warm_start(A, bodies, c, coefficient):
    b1, b2 = _bodies_of(A, bodies, c)
    t0, t1 = tangent_directions(A, c)
    for p in c["points"]:
        ti = p["tangent"]["impulse"] if p["tangent"] else (A.zero, A.zero)
        imp = A.scale(A.add(A.add(A.scale(c["normal"], p["impulse"]), A.scale(t0, ti[0])), A.scale(t1, ti[1])), coefficient)
        _apply(A, b1, b2, imp, p["anchor1"], p["anchor2"])
    _write_back(bodies, (b1, b2))


# ---- ContactConstraint::solve (contact/mod.rs:267-354, normal_part.rs:116-166, tangent_part.rs:155-244) ------------------------------------------
def solve(A, bodies, c, delta_secs, use_bias, max_overlap_solve_speed, softness):
    T = A.T
    b1, b2 = _bodies_of(A, bodies, c)
    soft = softness["non_dynamic"] if c["non_dynamic_softness"] else softness["dynamic"]
    n = c["normal"]
    delta_translation = A.sub(b2["dp"], b1["dp"])
    for p in c["points"]:
        r1 = A.qrot(b1["dq"], p["anchor1"]); r2 = A.qrot(b2["dq"], p["anchor2"])
        delta_separation = A.add(delta_translation, A.sub(r2, r1))
        separation = A.dot(delta_separation, n) + p["initial_separation"]
        r1, r2 = p["anchor1"], p["anchor2"]
        rel = A.sub(velocity_at_point(A, b2["lin"], b2["ang"], r2), velocity_at_point(A, b1["lin"], b1["ang"], r1))
        normal_speed = A.dot(rel, n)
        if separation > 0:
            impulse = -p["effective_mass"] * (normal_speed + separation / delta_secs)
        elif use_bias:
            bias = max(soft["bias"] * separation, -max_overlap_solve_speed)
            scaled_mass = soft["mass_scale"] * p["effective_mass"]
            scaled_impulse = soft["impulse_scale"] * p["impulse"]
            impulse = -scaled_mass * (normal_speed + bias) - scaled_impulse
        else:
            impulse = -p["effective_mass"] * normal_speed
        new_impulse = max(p["impulse"] + impulse, A.zero)
        impulse = new_impulse - p["impulse"]
        p["impulse"] = new_impulse
        p["total_impulse"] = p["total_impulse"] + new_impulse
        _apply(A, b1, b2, A.scale(n, impulse), r1, r2)
    t0, t1 = tangent_directions(A, c)
    for p in c["points"]:
        tp = p["tangent"]
        if tp is None: continue
        r1, r2 = p["anchor1"], p["anchor2"]
        rel = A.sub(velocity_at_point(A, b2["lin"], b2["ang"], r2), velocity_at_point(A, b1["lin"], b1["ang"], r1))
        impulse_limit = c["friction"] * p["impulse"]
        rel = A.add(rel, c["tangent_velocity"])
        ts1, ts2 = A.dot(rel, t0), A.dot(rel, t1)
        t11, t22, t12 = ts1 * ts1, ts2 * ts2, ts1 * ts2
        inv = t11 * tp["k"][0] + t22 * tp["k"][1] + t12 * tp["k"][2]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            effective_mass = (t11 + t22) * (A.one / inv)
        if not np.isfinite(effective_mass):
            out = (A.zero,) * 3
        else:
            d = (effective_mass * ts1, effective_mass * ts2)
            cand = (tp["impulse"][0] - d[0], tp["impulse"][1] - d[1])
            len_sq = cand[0] * cand[0] + cand[1] * cand[1]
            if len_sq > impulse_limit * impulse_limit:
                l = np.sqrt(len_sq)
                cand = (impulse_limit * (cand[0] / l), impulse_limit * (cand[1] / l))
            di = (cand[0] - tp["impulse"][0], cand[1] - tp["impulse"][1])
            tp["impulse"] = cand
            out = A.add(A.scale(t0, di[0]), A.scale(t1, di[1]))
        _apply(A, b1, b2, out, r1, r2)
    _write_back(bodies, (b1, b2))


# ---- ContactConstraint::apply_restitution (contact/mod.rs:358-407) + solve_restitution_internal (plugin.rs:680-718) ----------------------------
def restitution(A, bodies, c, threshold, iterations):
    if c["restitution"] == 0: return
    b1, b2 = _bodies_of(A, bodies, c)
    n = c["normal"]
    for _ in range(iterations if len(c["points"]) > 1 else 1):
        for p in c["points"]:
            if p["normal_speed"] > -threshold or p["total_impulse"] == 0: continue
            r1, r2 = p["anchor1"], p["anchor2"]
            rel = A.sub(velocity_at_point(A, b2["lin"], b2["ang"], r2), velocity_at_point(A, b1["lin"], b1["ang"], r1))
            normal_speed = A.dot(rel, n)
            impulse = -p["effective_mass"] * (normal_speed + c["restitution"] * p["normal_speed"])
            new_impulse = max(p["impulse"] + impulse, A.zero)
            impulse = new_impulse - p["impulse"]
            p["impulse"] = new_impulse
            p["total_impulse"] = p["total_impulse"] + impulse
            _apply(A, b1, b2, A.scale(n, impulse), r1, r2)
    _write_back(bodies, (b1, b2))


def in_solve_order(offsets):
    """Manifold indices in the order the passes visit them: the overflow colour first, then colours 0..22 (plugin.rs:461-479)."""
    order = list(range(int(offsets[23]), int(offsets[24])))
    for c in range(23):
        order += list(range(int(offsets[c]), int(offsets[c + 1])))
    return order


# ---- XPBD: DistanceLimit::compute_correction (dynamics/joints/mod.rs:321-340), compute_lagrange_update (xpbd/mod.rs:393-413),
#      DistanceJoint::solve (xpbd/joints/distance.rs:61-117), apply_positional_impulse (positional_constraint.rs:10-56) ------------------------------
def vdiv(a, s): return (a[0] / s, a[1] / s, a[2] / s)


def distance_limit_correction(A, lo, hi, separation):
    """(direction, magnitude) of the correction that brings |separation| back into [min, max]."""
    distance_squared = A.dot(separation, separation)
    if distance_squared <= A.eps:
        return (A.zero,) * 3, A.zero
    distance = np.sqrt(distance_squared)
    if distance < lo:
        return vdiv(separation, distance), lo - distance
    if distance > hi:
        return vdiv(A.neg(separation), distance), distance - hi
    return (A.zero,) * 3, A.zero


def compute_lagrange_update(A, lagrange, c, ws, compliance, dt):
    w_sum = A.zero
    for w in ws: w_sum = w_sum + w
    if w_sum <= A.eps: return A.zero
    tilde = compliance / (dt * dt)
    return (-c - tilde * lagrange) / (w_sum + tilde)


def distance_joint_prepare(A, pos, rot, com, j):
    """DistanceJoint::prepare (distance.rs:36-59) from the Position / Rotation / ComputedCenterOfMass components."""
    b1, b2 = j["body1"], j["body2"]
    r1 = A.qrot(rot[b1], A.sub(j["local_anchor1"], com[b1]))
    r2 = A.qrot(rot[b2], A.sub(j["local_anchor2"], com[b2]))
    cd = A.add(A.sub(pos[b2], pos[b1]), A.sub(A.qrot(rot[b2], com[b2]), A.qrot(rot[b1], com[b1])))
    return dict(world_r1=r1, world_r2=r2, center_difference=cd, total_lagrange=(A.zero,) * 3)


def distance_joint_solve(A, bodies, j, data, dt):
    """solve_xpbd_joint (xpbd/plugin.rs:145-189: DUMMY inertia for the dominant body, DUMMY body for a missing one) + DistanceJoint::solve."""
    i1, i2 = j["body1"], j["body2"]
    rel_dom = bodies.dominance_of(i1) - bodies.dominance_of(i2)
    bs = []
    for i, dummy in ((i1, rel_dom > 0), (i2, rel_dom < 0)):
        has = bodies.has_solver_body(i)
        im, I = bodies.inertia(i, dummy)
        bs.append(dict(i=i, has=has, dp=bodies.dp[i] if has else (A.zero,) * 3, dq=bodies.dq[i] if has else (A.zero, A.zero, A.zero, A.one), im=im, I=I))
    b1, b2 = bs
    world_r1 = A.qrot(b1["dq"], data["world_r1"]); world_r2 = A.qrot(b2["dq"], data["world_r2"])
    separation = A.add(A.add(A.sub(b2["dp"], b1["dp"]), A.sub(world_r2, world_r1)), data["center_difference"])
    direction, distance = distance_limit_correction(A, j["limit_min"], j["limit_max"], separation)
    if distance <= A.eps: return
    def gen_inv_mass(im, I, r):
        rn = A.cross(r, direction)
        return max(im[0], max(im[1], im[2])) + A.dot(rn, A.sym_mul(I, rn))
    w = [gen_inv_mass(b1["im"], b1["I"], world_r1), gen_inv_mass(b2["im"], b2["I"], world_r2)]
    delta_lagrange = compute_lagrange_update(A, A.zero, distance, w, j["compliance"], dt)
    impulse = A.scale(direction, delta_lagrange)
    data["total_lagrange"] = A.add(data["total_lagrange"], impulse)
    # apply_positional_impulse
    b1["dp"] = A.add(b1["dp"], A.cmul(impulse, b1["im"]))
    b1["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b1["I"], A.cross(world_r1, impulse))), b1["dq"])
    b2["dp"] = A.sub(b2["dp"], A.cmul(impulse, b2["im"]))
    b2["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b2["I"], A.cross(world_r2, A.neg(impulse)))), b2["dq"])
    for b in bs:
        if b["has"]: bodies.dp[b["i"]] = b["dp"]; bodies.dq[b["i"]] = b["dq"]


# ---- FixedJoint = FixedAngleConstraintShared, then PointConstraintShared (xpbd/joints/fixed.rs:36-91, shared/fixed_angle_constraint.rs:36-97,
#      shared/point_constraint.rs:36-110, angular_constraint.rs:144-190, 233-280) ------------------------------------------------------------------
def qconj(q): return (-q[0], -q[1], -q[2], q[3])   # Quat::inverse of a unit quaternion


def fixed_joint_prepare(A, pos, rot, com, j):
    d = distance_joint_prepare(A, pos, rot, com, j)   # PointConstraintShared::prepare is DistanceJoint::prepare's arithmetic
    b1, b2 = j["body1"], j["body2"]
    d["rotation_difference"] = A.qmul(A.qmul(rot[b1], j["local_basis1"]), qconj(A.qmul(rot[b2], j["local_basis2"])))
    d["total_rotation_lagrange"] = (A.zero,) * 3
    return d


def _joint_bodies(A, bodies, j):
    i1, i2 = j["body1"], j["body2"]
    rel_dom = bodies.dominance_of(i1) - bodies.dominance_of(i2)
    bs = []
    for i, dummy in ((i1, rel_dom > 0), (i2, rel_dom < 0)):
        has = bodies.has_solver_body(i)
        im, I = bodies.inertia(i, dummy)
        bs.append(dict(i=i, has=has, dp=bodies.dp[i] if has else (A.zero,) * 3, dq=bodies.dq[i] if has else (A.zero, A.zero, A.zero, A.one), im=im, I=I))
    return bs


def fixed_joint_solve(A, bodies, j, data, dt):
    b1, b2 = bs = _joint_bodies(A, bodies, j)
    # -- the angular constraint
    q = A.qmul(A.qmul(data["rotation_difference"], b1["dq"]), qconj(b2["dq"]))
    m2 = A.T(-2.0)
    difference = (m2 * q[0], m2 * q[1], m2 * q[2])
    angle = np.sqrt(A.dot(difference, difference))
    if not angle <= A.eps:
        axis = vdiv(difference, angle)
        w = [A.dot(axis, A.sym_mul(b1["I"], axis)), A.dot(axis, A.sym_mul(b2["I"], axis))]
        dl = compute_lagrange_update(A, A.zero, angle, w, j["compliance"][1], dt)
        if not abs(dl) <= A.eps:
            impulse = A.scale(axis, -dl)
            b1["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b1["I"], impulse)), b1["dq"])
            b2["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b2["I"], A.neg(impulse))), b2["dq"])
        data["total_rotation_lagrange"] = A.add(data["total_rotation_lagrange"], A.scale(axis, dl))
    # -- the point constraint
    world_r1 = A.qrot(b1["dq"], data["world_r1"]); world_r2 = A.qrot(b2["dq"], data["world_r2"])
    separation = A.add(A.add(A.sub(b2["dp"], b1["dp"]), A.sub(world_r2, world_r1)), data["center_difference"])
    magnitude_squared = A.dot(separation, separation)
    if magnitude_squared != 0:
        magnitude = np.sqrt(magnitude_squared)
        direction = vdiv(A.neg(separation), magnitude)
        def gen_inv_mass(im, I, r):
            rn = A.cross(r, direction)
            return max(im[0], max(im[1], im[2])) + A.dot(rn, A.sym_mul(I, rn))
        w = [gen_inv_mass(b1["im"], b1["I"], world_r1), gen_inv_mass(b2["im"], b2["I"], world_r2)]
        dl = compute_lagrange_update(A, A.zero, magnitude, w, j["compliance"][0], dt)
        impulse = A.scale(direction, dl)
        data["total_lagrange"] = A.add(data["total_lagrange"], impulse)
        b1["dp"] = A.add(b1["dp"], A.cmul(impulse, b1["im"]))
        b1["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b1["I"], A.cross(world_r1, impulse))), b1["dq"])
        b2["dp"] = A.sub(b2["dp"], A.cmul(impulse, b2["im"]))
        b2["dq"] = A.qmul(A.from_scaled_axis(A.sym_mul(b2["I"], A.cross(world_r2, A.neg(impulse)))), b2["dq"])
    for b in bs:
        if b["has"]: bodies.dp[b["i"]] = b["dp"]; bodies.dq[b["i"]] = b["dq"]


def unprepared(A):
    """Solver data of a joint whose `prepare` never ran: prepare_xpbd_joint only prepares joints whose two bodies pass `Without<RigidBodyDisabled>`
    (xpbd/plugin.rs:128, :138-140); solve_xpbd_joint still solves them, the disabled body as SolverBody::default() with the DUMMY inertia (:160-176),
    on the components' Default: zero vectors, the identity rotation difference."""
    z = (A.zero,) * 3
    return dict(world_r1=z, world_r2=z, center_difference=z, total_lagrange=z, total_rotation_lagrange=z, rotation_difference=(A.zero, A.zero, A.zero, A.one))


# ---- apply_local_acceleration (dynamics/rigid_body/forces/plugin.rs:207-241), written from its text --------------------------------------------------------
def apply_local_acceleration(A, lin_vel, ang_vel, delta_rotation, rotation, local_linear, local_angular, locked_bits, delta_secs):
    """let rotation = solver_body.delta_rotation * *rotation;  world = locked_axes.apply_to_vec(rotation * local)  for BOTH vectors (apply_to_vec = the translation
    locks: bits 0b100_000 x, 0b010_000 y, 0b001_000 z, locked_axes.rs:230-243);  velocity += world * delta_secs."""
    rot = A.qmul(delta_rotation, rotation)

    def apply_to_vec(v):
        return (A.zero if locked_bits & 0b100_000 else v[0], A.zero if locked_bits & 0b010_000 else v[1], A.zero if locked_bits & 0b001_000 else v[2])
    wl = apply_to_vec(A.qrot(rot, local_linear))
    wa = apply_to_vec(A.qrot(rot, local_angular))
    return A.add(lin_vel, A.scale(wl, delta_secs)), A.add(ang_vel, A.scale(wa, delta_secs))
