// k_xpbd.hip — XPBD joint projection (fixed, revolute, spherical, prismatic, distance), joint damping, joint forces.
//
// The reference solves joints in ONE serial loop in query order (xpbd/plugin.rs:145-189) — a Gauss-Seidel
// sweep whose result depends on that order.  To stay bit-identical AND parallel, joints are scheduled on
// the host at upload time (avn_world.cpp: JointSchedule):
//   * joints are split into connected components of the joint graph (independent "joint islands");
//   * inside a component, joint j gets level(j) = 1 + max(level(i) : i < j, i shares a body with j).
//     Two joints on the same level never share a body, and running levels in ascending order executes
//     every pair of body-sharing joints in their original relative order => identical to the serial loop.
// One workgroup owns one component and walks its levels with a workgroup barrier between levels; body state
// written by one wave is visible to the others after __syncthreads() because all waves of a workgroup share
// the CU's vector L1 (workgroup-scope coherence), so no agent-scope fences are needed.
//
// Reference functions replaced (paths relative to /root/reference/src/dynamics):
//   k_prepare_joints            solver/xpbd/plugin.rs:125-142 + prepare() of solver/xpbd/joints/{fixed,revolute,spherical,
//                               prismatic,distance}.rs and joints/shared/{point_constraint,fixed_angle_constraint}.rs
//   k_joint_schedule<0>         solver/xpbd/plugin.rs:145-189 + solve() of the same files, joints/mod.rs:321-357,427-472,
//                               xpbd/mod.rs:393-413, xpbd/positional_constraint.rs:10-93, xpbd/angular_constraint.rs:50-184
//   k_joint_damping             solver/plugin.rs:759-806
//   k_writeback_joint_forces    solver/xpbd/plugin.rs:242-260
#include "avn_kernels.h"

namespace avn {

#define JOINT_THREADS 64

// joint meta word (w lane of j_par): bit 0 has JointDamping, bits 8-15 AVN_JOINT_* type, bits 16-23 AVN_JOINT_HAS_LIMIT*
__device__ __forceinline__ uint32_t jm_type(uint32_t m) { return (m >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t jm_limits(uint32_t m) { return (m >> 16) & 0xFFu; }

// prepare_xpbd_joint<T> (xpbd/plugin.rs:125-142) + the per-type prepare():
//   point constraint  xpbd/joints/shared/point_constraint.rs:38-53     fixed angle  shared/fixed_angle_constraint.rs:38-57
//   fixed fixed.rs:39-72   revolute revolute.rs:48-90   spherical spherical.rs:44-83   prismatic prismatic.rs:43-81
//   distance distance.rs:36-59
template <class T>
__global__ __launch_bounds__(256) void k_prepare_joints(DW<T> w) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    Vec4<T> z = make4<T>(0, 0, 0, 0);
    w.j_lag[j] = z; w.j_rl0[j] = z; w.j_rl1[j] = z;  // clear_lagrange_multipliers
    int2 b = w.j_bodies[j];
    uint32_t m1 = w.bmeta[b.x], m2 = w.bmeta[b.y];
    if ((meta_flags(m1) | meta_flags(m2)) & AVN_BODY_DISABLED) return;  // bodies.get_many fails: solver data untouched
    uint32_t type = jm_type(scalar_to_bits(w.j_par[j].w));
    Q4<T> q1 = quat<T>(w.rot[b.x]), q2 = quat<T>(w.rot[b.y]);
    V3<T> com1 = xyz<T>(w.com[b.x]), com2 = xyz<T>(w.com[b.y]);
    V3<T> p1 = xyz<T>(w.pos[b.x]), p2 = xyz<T>(w.pos[b.y]);
    V3<T> anchor1 = xyz<T>(w.j_a1[j]), anchor2 = xyz<T>(w.j_a2[j]);
    V3<T> cd = (p2 - p1) + (qrot(q2, com2) - qrot(q1, com1));
    Q4<T> basis1 = quat<T>(w.j_b1[j]), basis2 = quat<T>(w.j_b2[j]);
    V3<T> axis = xyz<T>(w.j_ax[j]);
    if (type == AVN_JOINT_SPHERICAL) {  // rotation MATRICES here, like the reference (spherical.rs:66-82)
        M3<T> r1m = mat3_from_quat(q1), r2m = mat3_from_quat(q2);
        w.j_r1[j] = make4<T>(mmul(r1m, anchor1 - com1), 0);
        w.j_r2[j] = make4<T>(mmul(r2m, anchor2 - com2), 0);
        w.j_cd[j] = make4<T>(cd, 0);
        V3<T> swing_axis = any_orthonormal_vector(axis);
        w.j_s0[j] = make4<T>(mmul(r1m, qrot(basis1, swing_axis)), 0);
        w.j_s1[j] = make4<T>(mmul(r2m, qrot(basis2, swing_axis)), 0);
        w.j_s2[j] = make4<T>(mmul(r1m, qrot(basis1, axis)), 0);
        w.j_s3[j] = make4<T>(mmul(r2m, qrot(basis2, axis)), 0);
        return;
    }
    w.j_r1[j] = make4<T>(qrot(q1, anchor1 - com1), 0);
    w.j_r2[j] = make4<T>(qrot(q2, anchor2 - com2), 0);
    w.j_cd[j] = make4<T>(cd, 0);
    if (type == AVN_JOINT_FIXED || type == AVN_JOINT_PRISMATIC)
        w.j_s0[j] = make4<T>(qmul(qmul(q1, basis1), qinverse(qmul(q2, basis2))));
    if (type == AVN_JOINT_PRISMATIC) w.j_s1[j] = make4<T>(qrot(qmul(q1, basis1), axis), 0);  // free_axis1
    if (type == AVN_JOINT_REVOLUTE) {
        Q4<T> f1 = qmul(q1, basis1), f2 = qmul(q2, basis2);
        V3<T> ortho = any_orthonormal_vector(axis);
        w.j_s0[j] = make4<T>(qrot(f1, axis), 0); w.j_s1[j] = make4<T>(qrot(f2, axis), 0);    // a1, a2
        w.j_s2[j] = make4<T>(qrot(f1, ortho), 0); w.j_s3[j] = make4<T>(qrot(f2, ortho), 0);  // b1, b2
    }
}

template <class T> struct JBody {
    V3<T> dp; Q4<T> dq; V3<T> inv_mass; Sym3<T> I; T dp_w;
};
// (all four records are loaded unconditionally -- a body without a SolverBody still has valid slots -- and the DUMMY substitutions are
//  selects on the loaded values: the loads do not wait for the flag words, one memory round trip less per schedule level)
template <class T> __device__ __forceinline__ void jselect(Vec4<T> d, Vec4<T> q, Vec4<T> sa, Vec4<T> sb, bool nobody, bool dummy_inertia, JBody<T>& b) {
    if (nobody) { b.dp = vzero<T>(); b.dq = qidentity<T>(); b.inv_mass = vzero<T>(); b.I = sym_zero<T>(); b.dp_w = 0; return; }
    b.dp = xyz<T>(d); b.dp_w = d.w; b.dq = quat<T>(q);
    if (dummy_inertia) { b.inv_mass = vzero<T>(); b.I = sym_zero<T>(); }
    else {
        b.inv_mass = effective_inv_mass<T>(sa.x, scalar_to_bits(sb.w));
        b.I = Sym3<T>{sa.y, sa.z, sa.w, sb.x, sb.y, sb.z};
    }
}
template <class T> __device__ __forceinline__ void jload(const DW<T>& w, int idx, bool nobody, bool dummy_inertia, JBody<T>& b) {
    Vec4<T> d = w.sb_dp[idx], q = w.sb_dq[idx], sa = w.si_a[idx], sb = w.si_b[idx];
    jselect<T>(d, q, sa, sb, nobody, dummy_inertia, b);
}
// xpbd/mod.rs:393-413 compute_lagrange_update (w = [w1, w2]; `iter().sum()` starts from 0.0)
template <class T> __device__ __forceinline__ T compute_lagrange_update(T lagrange, T c, T w1, T w2, T compliance, T dt) {
    T w_sum = T(0) + w1 + w2;
    if (w_sum <= Limits<T>::eps) return T(0);
    T tilde_compliance = compliance / (dt * dt);
    return (-c - tilde_compliance * lagrange) / (w_sum + tilde_compliance);
}
// positional_constraint.rs:10-51 / :68-82
template <class T> __device__ __forceinline__ void apply_positional_impulse(JBody<T>& b1, JBody<T>& b2, V3<T> impulse, V3<T> r1, V3<T> r2) {
    b1.dp = b1.dp + cmul(impulse, b1.inv_mass);
    b1.dq = qmul(from_scaled_axis(smul(b1.I, cross(r1, impulse))), b1.dq);
    b2.dp = b2.dp - cmul(impulse, b2.inv_mass);
    b2.dq = qmul(from_scaled_axis(smul(b2.I, cross(r2, -impulse))), b2.dq);
}
template <class T> __device__ __forceinline__ T positional_w(T inv_mass_max, const Sym3<T>& I, V3<T> r, V3<T> dir) {
    V3<T> rc = cross(r, dir);
    return inv_mass_max + dot(rc, smul(I, rc));
}
// the per-joint solver data a solve touches, held in registers for the duration of the joint
template <class T> struct JData { V3<T> r1, r2, cd, lag, rl0, rl1; };

// shared/point_constraint.rs:56-108
template <class T> __device__ __forceinline__ void point_constraint_solve(JData<T>& d, JBody<T>& b1, JBody<T>& b2, T compliance, T dt) {
    V3<T> world_r1 = qrot(b1.dq, d.r1), world_r2 = qrot(b2.dq, d.r2);
    V3<T> separation = ((b2.dp - b1.dp) + (world_r2 - world_r1)) + d.cd;
    T magnitude_squared = length_squared(separation);
    if (magnitude_squared == T(0)) return;
    T magnitude = sqrt_t(magnitude_squared);
    V3<T> dir = (-separation) / magnitude;
    T w1 = positional_w(max_element(b1.inv_mass), b1.I, world_r1, dir);
    T w2 = positional_w(max_element(b2.inv_mass), b2.I, world_r2, dir);
    T delta_lagrange = compute_lagrange_update<T>(T(0), magnitude, w1, w2, compliance, dt);
    V3<T> impulse = delta_lagrange * dir;
    d.lag = d.lag + impulse;
    apply_positional_impulse(b1, b2, impulse, world_r1, world_r2);
}
// angular_constraint.rs:146-184 align_orientation + :50-93 apply_angular_lagrange_update / apply_angular_impulse (3D)
template <class T> __device__ __forceinline__ V3<T> align_orientation(JBody<T>& b1, JBody<T>& b2, V3<T> rotation_difference, T lagrange, T compliance, T dt) {
    T angle = length(rotation_difference);
    if (angle <= Limits<T>::eps) return vzero<T>();
    V3<T> axis = rotation_difference / angle;
    T w1 = dot(axis, smul(b1.I, axis)), w2 = dot(axis, smul(b2.I, axis));
    T delta_lagrange = compute_lagrange_update<T>(lagrange, angle, w1, w2, compliance, dt);
    if (!(fabs_t(delta_lagrange) <= Limits<T>::eps)) {
        V3<T> impulse = (-delta_lagrange) * axis;
        b1.dq = qmul(from_scaled_axis(smul(b1.I, impulse)), b1.dq);
        b2.dq = qmul(from_scaled_axis(smul(b2.I, -impulse)), b2.dq);
    }
    return delta_lagrange * axis;
}
// shared/fixed_angle_constraint.rs:60-95
template <class T> __device__ __forceinline__ void fixed_angle_solve(JData<T>& d, Q4<T> rotation_difference, JBody<T>& b1, JBody<T>& b2, T compliance, T dt) {
    Q4<T> q = qmul(qmul(rotation_difference, b1.dq), qinverse(b2.dq));
    V3<T> difference = T(-2) * V3<T>{q.x, q.y, q.z};
    d.rl0 = d.rl0 + align_orientation(b1, b2, difference, T(0), compliance, dt);
}
// dynamics/joints/mod.rs:427-472 AngleLimit::compute_correction (3D)
template <class T> __device__ __forceinline__ bool angle_limit_correction(T lim_min, T lim_max, V3<T> limit_axis, V3<T> axis1, V3<T> axis2, T max_correction, V3<T>& out) {
    const T PI = T(3.14159265358979323846264338327950288), TAU = T(6.28318530717958647692528676655900577);
    T phi = asin_t(dot(cross(axis1, axis2), limit_axis));
    if (dot(axis1, axis2) < T(0)) phi = PI - phi;
    if (phi > PI) phi -= TAU;
    if (phi < lim_min || phi > lim_max) {
        phi = clamp_t(phi, lim_min, lim_max);
        Q4<T> rot = from_axis_angle(limit_axis, phi);
        out = clamp_length_max(cross(qrot(rot, axis1), axis2), max_correction);
        return true;
    }
    return false;
}
// dynamics/joints/mod.rs:345-357 DistanceLimit::compute_correction_along_axis
template <class T> __device__ __forceinline__ V3<T> correction_along_axis(T lim_min, T lim_max, V3<T> separation, V3<T> axis) {
    T a = dot(separation, axis);
    if (a < lim_min) return axis * (lim_min - a);
    if (a > lim_max) return (-axis) * (a - lim_max);
    return vzero<T>();
}

// solve_xpbd_joint<T> (xpbd/plugin.rs:145-189) for one joint of any type: fixed.rs:74-91, revolute.rs:92-183,
// spherical.rs:85-209, prismatic.rs:83-192, distance.rs:61-117
template <class T> __device__ __forceinline__ void joint_solve_one(const DW<T>& w, const StepParams<T>& p, uint32_t j, int2 b) {
    // one memory level: flags, both bodies' four records and the joint's own records are all addressed by (j, b) from the schedule record
    uint32_t f1 = w.sb_flags[b.x], f2 = w.sb_flags[b.y];
    Vec4<T> d1 = w.sb_dp[b.x], q1 = w.sb_dq[b.x], sa1 = w.si_a[b.x], sb1 = w.si_b[b.x];
    Vec4<T> d2 = w.sb_dp[b.y], q2 = w.sb_dq[b.y], sa2 = w.si_a[b.y], sb2 = w.si_b[b.y];
    Vec4<T> a1v = w.j_a1[j], a2v = w.j_a2[j], par = w.j_par[j];
    Vec4<T> jr1 = w.j_r1[j], jr2 = w.j_r2[j], jcd = w.j_cd[j], jlag = w.j_lag[j];
    asm volatile("" ::: "memory");   // (keeps the loads above in one batch: see k_contacts.hip solve_core)
    bool nobody1 = f1 & AVN_SBF_NO_SOLVER_BODY, nobody2 = f2 & AVN_SBF_NO_SOLVER_BODY;
    // dominance of the (possibly DUMMY) inertias; DUMMY rows carry dominance 128
    int dom1 = (int)(int16_t)(scalar_to_bits(sb1.w) >> 16), dom2 = (int)(int16_t)(scalar_to_bits(sb2.w) >> 16);
    int rel = dom1 - dom2;
    JBody<T> b1, b2;
    jselect<T>(d1, q1, sa1, sb1, nobody1, rel > 0, b1);
    jselect<T>(d2, q2, sa2, sb2, nobody2, rel < 0, b2);
    uint32_t meta = scalar_to_bits(par.w);
    uint32_t type = jm_type(meta), limits = jm_limits(meta);
    T limit_min = a1v.w, limit_max = a2v.w, c0 = par.x;
    T dt = p.h_adj;
    const T PI = T(3.14159265358979323846264338327950288), EPS = Limits<T>::eps;
    JData<T> d;
    d.r1 = xyz<T>(jr1); d.r2 = xyz<T>(jr2); d.cd = xyz<T>(jcd);
    d.lag = xyz<T>(jlag);
    d.rl0 = vzero<T>(); d.rl1 = vzero<T>();
    bool angular = type != AVN_JOINT_DISTANCE;
    T c1 = 0, c2 = 0;
    if (angular) { d.rl0 = xyz<T>(w.j_rl0[j]); d.rl1 = xyz<T>(w.j_rl1[j]); c1 = w.j_ax[j].w; c2 = w.j_l2[j].z; }

    if (type == AVN_JOINT_DISTANCE) {
        V3<T> world_r1 = qrot(b1.dq, d.r1), world_r2 = qrot(b2.dq, d.r2);
        V3<T> separation = ((b2.dp - b1.dp) + (world_r2 - world_r1)) + d.cd;
        // DistanceLimit::compute_correction (dynamics/joints/mod.rs:321-340)
        V3<T> dir = vzero<T>();
        T distance = 0;
        T dsq = length_squared(separation);
        if (!(dsq <= EPS)) {
            T dd = sqrt_t(dsq);
            if (dd < limit_min) { dir = separation / dd; distance = limit_min - dd; }
            else if (dd > limit_max) { dir = (-separation) / dd; distance = dd - limit_max; }
        }
        if (distance <= EPS) return;  // nothing was modified
        T w1 = positional_w(max_element(b1.inv_mass), b1.I, world_r1, dir);
        T w2 = positional_w(max_element(b2.inv_mass), b2.I, world_r2, dir);
        T delta_lagrange = compute_lagrange_update<T>(T(0), distance, w1, w2, c0, dt);
        V3<T> impulse = delta_lagrange * dir;
        d.lag = d.lag + impulse;
        apply_positional_impulse(b1, b2, impulse, world_r1, world_r2);
    } else if (type == AVN_JOINT_FIXED) {
        fixed_angle_solve(d, quat<T>(w.j_s0[j]), b1, b2, c1, dt);
        point_constraint_solve(d, b1, b2, c0, dt);
    } else if (type == AVN_JOINT_REVOLUTE) {
        V3<T> sa1 = xyz<T>(w.j_s0[j]), sa2 = xyz<T>(w.j_s1[j]);
        {
            V3<T> a1 = qrot(b1.dq, sa1), a2 = qrot(b2.dq, sa2);
            d.rl0 = d.rl0 + align_orientation(b1, b2, cross(a1, a2), T(0), c1, dt);
        }
        if (limits & AVN_JOINT_HAS_LIMIT1) {
            V3<T> a1 = qrot(b1.dq, sa1), bb1 = qrot(b1.dq, xyz<T>(w.j_s2[j])), bb2 = qrot(b2.dq, xyz<T>(w.j_s3[j]));
            V3<T> correction;
            if (angle_limit_correction(limit_min, limit_max, a1, bb1, bb2, PI, correction))
                d.rl1 = d.rl1 + align_orientation(b1, b2, correction, T(0), c2, dt);
        }
        point_constraint_solve(d, b1, b2, c0, dt);
    } else if (type == AVN_JOINT_SPHERICAL) {
        point_constraint_solve(d, b1, b2, c0, dt);
        V3<T> sw1 = xyz<T>(w.j_s0[j]), sw2 = xyz<T>(w.j_s1[j]);
        if (limits & AVN_JOINT_HAS_LIMIT1) {  // apply_swing_limits
            V3<T> a1 = qrot(b1.dq, sw1), a2 = qrot(b2.dq, sw2);
            V3<T> n = cross(a1, a2);
            T n_magnitude = length(n);
            if (!(n_magnitude <= EPS)) {
                n = n / n_magnitude;
                V3<T> correction;
                if (angle_limit_correction(limit_min, limit_max, n, a1, a2, PI, correction))
                    d.rl0 = d.rl0 + align_orientation(b1, b2, correction, T(0), c1, dt);
            }
        }
        if (limits & AVN_JOINT_HAS_LIMIT2) {  // apply_twist_limits (every early `return` there only skips the rest of this block)
            V3<T> a1 = qrot(b1.dq, sw1), a2 = qrot(b2.dq, sw2);
            V3<T> n = a1 + a2;
            T n_magnitude = length(n);
            if (!(n_magnitude <= EPS)) {
                V3<T> tb1 = qrot(b1.dq, xyz<T>(w.j_s2[j])), tb2 = qrot(b2.dq, xyz<T>(w.j_s3[j]));
                n = n / n_magnitude;
                V3<T> n1 = tb1 - dot(n, tb1) * n, n2 = tb2 - dot(n, tb2) * n;
                T n1_magnitude = length(n1), n2_magnitude = length(n2);
                if (!(n1_magnitude <= EPS || n2_magnitude <= EPS)) {
                    n1 = n1 / n1_magnitude; n2 = n2 / n2_magnitude;
                    T max_correction = dot(a1, a2) > T(-0.5) ? T(2) * PI : dt;
                    Vec4<T> l2 = w.j_l2[j];
                    V3<T> correction;
                    if (angle_limit_correction(l2.x, l2.y, n, n1, n2, max_correction, correction))
                        d.rl1 = d.rl1 + align_orientation(b1, b2, correction, T(0), c2, dt);
                }
            }
        }
    } else {  // AVN_JOINT_PRISMATIC
        fixed_angle_solve(d, quat<T>(w.j_s0[j]), b1, b2, c1, dt);
        V3<T> world_r1 = qrot(b1.dq, d.r1), world_r2 = qrot(b2.dq, d.r2);
        V3<T> delta_x = vzero<T>();
        V3<T> axis1 = qrot(b1.dq, xyz<T>(w.j_s1[j]));
        V3<T> separation = ((b2.dp - b1.dp) + (world_r2 - world_r1)) + d.cd;
        if (limits & AVN_JOINT_HAS_LIMIT1) delta_x = delta_x + correction_along_axis(limit_min, limit_max, separation, axis1);
        V3<T> axis2 = any_orthogonal_vector(axis1);
        V3<T> axis3 = cross(axis1, axis2);
        delta_x = delta_x + correction_along_axis(T(0), T(0), separation, axis2);  // DistanceLimit::ZERO
        delta_x = delta_x + correction_along_axis(T(0), T(0), separation, axis3);
        T magnitude = length(delta_x);
        if (!(magnitude <= EPS)) {
            V3<T> dir = delta_x / magnitude;
            T w1 = positional_w(max_element(b1.inv_mass), b1.I, world_r1, dir);
            T w2 = positional_w(max_element(b2.inv_mass), b2.I, world_r2, dir);
            T delta_lagrange = compute_lagrange_update<T>(T(0), magnitude, w1, w2, c0, dt);
            V3<T> impulse = delta_lagrange * dir;
            d.lag = d.lag + impulse;
            apply_positional_impulse(b1, b2, impulse, world_r1, world_r2);
        }
    }
    w.j_lag[j] = make4<T>(d.lag, 0);
    if (angular) { w.j_rl0[j] = make4<T>(d.rl0, 0); w.j_rl1[j] = make4<T>(d.rl1, 0); }
    if (!nobody1) { w.sb_dp[b.x] = make4<T>(b1.dp, b1.dp_w); w.sb_dq[b.x] = make4<T>(b1.dq); }
    if (!nobody2) { w.sb_dp[b.y] = make4<T>(b2.dp, b2.dp_w); w.sb_dq[b.y] = make4<T>(b2.dq); }
}

template <class T> __device__ __forceinline__ void joint_damping_one(const DW<T>& w, const StepParams<T>& p, uint32_t j, int2 b) {
    Vec4<T> par = w.j_par[j];
    if (!(scalar_to_bits(par.w) & 1u)) return;  // no JointDamping component
    uint32_t f1 = w.sb_flags[b.x], f2 = w.sb_flags[b.y];
    bool nobody1 = f1 & AVN_SBF_NO_SOLVER_BODY, nobody2 = f2 & AVN_SBF_NO_SOLVER_BODY;
    T delta_secs = p.h_adj;
    // Missing bodies use the two DUMMY SolverBodies that the reference declares OUTSIDE its joint loop
    // (solver/plugin.rs:766-767): they are shared by all joints and their angular velocity is mutated, so they
    // live in virtual body slots behind the real bodies that the host resets before every launch and that the damping
    // schedule treats as ordinary bodies (=> joints touching them are serialised, as in the reference).  joint_damping::<T>
    // is one system per joint type with FRESH dummies (plugin.rs:139-150): slots n_bodies + 2 * type, + 2 * type + 1.
    uint32_t jtype = jm_type(scalar_to_bits(par.w));
    int i1 = nobody1 ? (int)(w.n_bodies + 2u * jtype) : b.x, i2 = nobody2 ? (int)(w.n_bodies + 2u * jtype + 1u) : b.y;
    Vec4<T> l1 = w.sb_lin[i1], g1 = w.sb_ang[i1], l2 = w.sb_lin[i2], g2 = w.sb_ang[i2];
    V3<T> v1 = xyz<T>(l1), om1 = xyz<T>(g1), v2 = xyz<T>(l2), om2 = xyz<T>(g2);
    V3<T> delta_omega = (om2 - om1) * smin(par.z * delta_secs, T(1));
    if (nobody1 || !(f1 & AVN_SB_KINEMATIC)) om1 = om1 + delta_omega;
    if (nobody2 || !(f2 & AVN_SB_KINEMATIC)) om2 = om2 - delta_omega;
    V3<T> delta_v = (v2 - v1) * smin(par.y * delta_secs, T(1));
    V3<T> w1 = vzero<T>(), w2 = vzero<T>();  // DUMMY inertia for missing bodies; no dominance swap in joint_damping
    if (!nobody1) w1 = effective_inv_mass<T>(w.si_a[b.x].x, scalar_to_bits(w.si_b[b.x].w));
    if (!nobody2) w2 = effective_inv_mass<T>(w.si_a[b.y].x, scalar_to_bits(w.si_b[b.y].w));
    V3<T> pp = cmul(delta_v, recip_or_zero(w1 + w2));
    v1 = v1 + cmul(pp, w1);
    v2 = v2 - cmul(pp, w2);
    w.sb_lin[i1] = make4<T>(v1, l1.w); w.sb_ang[i1] = make4<T>(om1, g1.w);
    w.sb_lin[i2] = make4<T>(v2, l2.w); w.sb_ang[i2] = make4<T>(om2, g2.w);
}

// One workgroup (one wave) per joint component; levels separated by workgroup barriers.
//   comp_level_begin[c] .. comp_level_begin[c+1]  : level slots of component c
//   level_offsets[l] .. level_offsets[l+1]        : schedule slots of level slot l
//   rec[k] = (joint, body1, body2, 0)             : the slot's record, written by the host with the schedule
// A chain is the worst case: 99 levels of ONE joint, every level a dependent walk through memory.  The walk used to be
// level_offsets -> order -> j_bodies -> flags / dominance -> body records (five round trips per level); with the bodies in the slot record
// and the record of the lane's NEXT level fetched while the current one is solved, a level costs one round trip (its body and joint
// records, all addressed by the record) plus the solve.
template <class T, int OP>
__global__ __launch_bounds__(JOINT_THREADS) void k_joint_schedule(DW<T> w, StepParams<T> p, const uint32_t* __restrict__ comp_level_begin,
                                                                   const uint32_t* __restrict__ level_offsets,
                                                                   const int4* __restrict__ rec) {
    const uint32_t c = blockIdx.x;
    const uint32_t l0 = comp_level_begin[c], l1 = comp_level_begin[c + 1];
    if (l0 >= l1) return;
    uint32_t j0 = level_offsets[l0], j1 = level_offsets[l0 + 1];
    int4 cur = make_int4(0, 0, 0, 0);
    if (j0 + threadIdx.x < j1) cur = rec[j0 + threadIdx.x];
    for (uint32_t l = l0; l < l1; ++l) {
        // the next level's bounds and this lane's first record there: independent of anything this level computes
        uint32_t n0 = j1, n1 = j1;
        int4 nxt = make_int4(0, 0, 0, 0);
        if (l + 1 < l1) { n1 = level_offsets[l + 2]; if (n0 + threadIdx.x < n1) nxt = rec[n0 + threadIdx.x]; }
        for (uint32_t k = j0 + threadIdx.x; k < j1; k += JOINT_THREADS) {
            const int4 r = (k == j0 + threadIdx.x) ? cur : rec[k];
            if (OP == 0) joint_solve_one<T>(w, p, (uint32_t)r.x, make_int2(r.y, r.z));
            else joint_damping_one<T>(w, p, (uint32_t)r.x, make_int2(r.y, r.z));
        }
        __syncthreads();
        j0 = n0; j1 = n1; cur = nxt;
    }
}

// Round 6: the same level walk with the component's records in LDS.  A level of the global form is a dependent trip through memory -- the bodies a level writes are
// the bodies the next level reads, and on gfx9 a pending store turns every later load wait into vmcnt(0): ~2.2 us per level for ~1 us of arithmetic, 220 us for the
// 99 levels of cfg3's chains.  Here the workgroup stages everything joint_solve_one touches -- per body SolverBody delta position / rotation, SolverBodyInertia and flags,
// per joint its fifteen records -- once, walks the levels on LDS with workgroup barriers (the same functions through a DW whose pointers name the LDS copies and whose
// indices are the component's local ones: rec.w = local slot of body1 | body2 << 16, the joint's local index = its schedule slot), and writes back what the solve changes
// (delta position / rotation, the three lagrange accumulators) at the end.  Every body and every joint sees the global form's operation sequence: same bits.
// Components the host found too large for the workgroup's LDS (rec.w = 0xFFFFFFFF) walk the global form.
#define JL_BODY_RECS 4u     // dp dq si_a si_b (+ one flags word)
#define JL_JOINT_RECS 15u   // a1 a2 par r1 r2 cd lag rl0 rl1 ax l2 s0 s1 s2 s3
template <class T>
__global__ __launch_bounds__(JOINT_THREADS) void k_joint_schedule_lds(DW<T> w, StepParams<T> p, const uint32_t* __restrict__ comp_level_begin, const uint32_t* __restrict__ level_offsets,
                                                                        const int4* __restrict__ rec, const uint32_t* __restrict__ comp_bodies) {
    extern __shared__ __align__(16) unsigned char jl_smem[];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    const uint32_t l0 = comp_level_begin[c], l1 = comp_level_begin[c + 1];
    if (l0 >= l1) return;
    const uint32_t s0 = level_offsets[l0], s1 = level_offsets[l1];   // the component's schedule slots
    const uint32_t nj = s1 - s0, nb = comp_bodies[c];
    if (nb == 0xFFFFFFFFu) {   // too large for LDS: the global walk
        uint32_t j0 = s0, j1 = level_offsets[l0 + 1];
        for (uint32_t l = l0; l < l1; ++l) {
            for (uint32_t k = j0 + t; k < j1; k += JOINT_THREADS) { const int4 r = rec[k]; joint_solve_one<T>(w, p, (uint32_t)r.x, make_int2(r.y, r.z)); }
            __syncthreads();
            j0 = j1; if (l + 1 < l1) j1 = level_offsets[l + 2];
        }
        return;
    }
    Vec4<T>* const lb = reinterpret_cast<Vec4<T>*>(jl_smem);                    // [nb] (dp | dq) pairs, then [nb] (si_a | si_b) pairs
    Vec4<T>* const lj = lb + (size_t)JL_BODY_RECS * nb;                         // [15][nj]
    uint32_t* const lf = reinterpret_cast<uint32_t*>(lj + (size_t)JL_JOINT_RECS * nj);   // [nb]
    DW<T> lw = w;
    lw.sb_dp.p = lb; lw.sb_dq.p = lb + 1; lw.si_a.p = lb + 2u * nb; lw.si_b.p = lb + 2u * nb + 1; lw.sb_flags = lf;   // (Pair2: field[i] is p[2 i] -- the two records of a pair interleave, as in HBM)
    lw.j_a1 = lj; lw.j_a2 = lj + nj; lw.j_par = lj + 2u * nj; lw.j_r1 = lj + 3u * nj; lw.j_r2 = lj + 4u * nj; lw.j_cd = lj + 5u * nj; lw.j_lag = lj + 6u * nj;
    lw.j_rl0 = lj + 7u * nj; lw.j_rl1 = lj + 8u * nj; lw.j_ax = lj + 9u * nj; lw.j_l2 = lj + 10u * nj; lw.j_s0 = lj + 11u * nj; lw.j_s1 = lj + 12u * nj; lw.j_s2 = lj + 13u * nj; lw.j_s3 = lj + 14u * nj;
    for (uint32_t k = t; k < nj; k += JOINT_THREADS) {   // stage: one round trip for the whole component (a body named by several joints is written several times with the same bits)
        const int4 r = rec[s0 + k];
        const uint32_t j = (uint32_t)r.x, a = (uint32_t)r.w & 0xFFFFu, b = (uint32_t)r.w >> 16;
        lw.j_a1[k] = w.j_a1[j]; lw.j_a2[k] = w.j_a2[j]; lw.j_par[k] = w.j_par[j]; lw.j_r1[k] = w.j_r1[j]; lw.j_r2[k] = w.j_r2[j]; lw.j_cd[k] = w.j_cd[j]; lw.j_lag[k] = w.j_lag[j];
        lw.j_rl0[k] = w.j_rl0[j]; lw.j_rl1[k] = w.j_rl1[j]; lw.j_ax[k] = w.j_ax[j]; lw.j_l2[k] = w.j_l2[j]; lw.j_s0[k] = w.j_s0[j]; lw.j_s1[k] = w.j_s1[j]; lw.j_s2[k] = w.j_s2[j]; lw.j_s3[k] = w.j_s3[j];
        lw.sb_dp[a] = w.sb_dp[r.y]; lw.sb_dq[a] = w.sb_dq[r.y]; lw.si_a[a] = w.si_a[r.y]; lw.si_b[a] = w.si_b[r.y]; lf[a] = w.sb_flags[r.y];
        lw.sb_dp[b] = w.sb_dp[r.z]; lw.sb_dq[b] = w.sb_dq[r.z]; lw.si_a[b] = w.si_a[r.z]; lw.si_b[b] = w.si_b[r.z]; lf[b] = w.sb_flags[r.z];
    }
    __syncthreads();
    uint32_t j0 = s0, j1 = level_offsets[l0 + 1];
    for (uint32_t l = l0; l < l1; ++l) {
        for (uint32_t k = j0 + t; k < j1; k += JOINT_THREADS) {
            const uint32_t sl = (uint32_t)rec[k].w;
            joint_solve_one<T>(lw, p, k - s0, make_int2((int)(sl & 0xFFFFu), (int)(sl >> 16)));
        }
        __syncthreads();
        j0 = j1; if (l + 1 < l1) j1 = level_offsets[l + 2];
    }
    for (uint32_t k = t; k < nj; k += JOINT_THREADS) {   // write back what the solve writes (joint_solve_one's stores; a body without a SolverBody was never written)
        const int4 r = rec[s0 + k];
        const uint32_t j = (uint32_t)r.x, a = (uint32_t)r.w & 0xFFFFu, b = (uint32_t)r.w >> 16;
        w.j_lag[j] = lw.j_lag[k]; w.j_rl0[j] = lw.j_rl0[k]; w.j_rl1[j] = lw.j_rl1[k];
        if (!(lf[a] & AVN_SBF_NO_SOLVER_BODY)) { w.sb_dp[r.y] = lw.sb_dp[a]; w.sb_dq[r.y] = lw.sb_dq[a]; }
        if (!(lf[b] & AVN_SBF_NO_SOLVER_BODY)) { w.sb_dp[r.z] = lw.sb_dp[b]; w.sb_dq[r.z] = lw.sb_dq[b]; }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_writeback_joint_forces(DW<T> w, StepParams<T> p) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    T delta_secs = p.dt_adj;
    T rhs = recip_or_zero(delta_secs * delta_secs) * p.substeps_as_scalar;
    w.j_force[j] = make4<T>(xyz<T>(w.j_lag[j]) * rhs, 0);
    w.j_torque[j] = make4<T>((xyz<T>(w.j_rl0[j]) + xyz<T>(w.j_rl1[j])) * rhs, 0);  // total_rotation_lagrange() * rhs
}

template <class T> void launch_prepare_joints(const DW<T>& w, hipStream_t s) {
    if (w.n_joints) hipLaunchKernelGGL(k_prepare_joints<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w);
}
template <class T> void launch_joint_schedule(const DW<T>& w, const StepParams<T>& p, int op, uint32_t n_components,
                                              const uint32_t* comp_level_begin, const uint32_t* level_offsets, const int4* rec, hipStream_t s) {
    if (!w.n_joints || !n_components) return;
    if (op == 0) hipLaunchKernelGGL((k_joint_schedule<T, 0>), dim3(n_components), dim3(JOINT_THREADS), 0, s, w, p, comp_level_begin, level_offsets, rec);
    else hipLaunchKernelGGL((k_joint_schedule<T, 1>), dim3(n_components), dim3(JOINT_THREADS), 0, s, w, p, comp_level_begin, level_offsets, rec);
}
template <class T> void launch_joint_schedule_lds(const DW<T>& w, const StepParams<T>& p, uint32_t n_components, const uint32_t* comp_level_begin, const uint32_t* level_offsets, const int4* rec,
                                                  const uint32_t* comp_bodies, uint32_t lds_bytes, hipStream_t s) {
    if (!w.n_joints || !n_components) return;
    hipLaunchKernelGGL((k_joint_schedule_lds<T>), dim3(n_components), dim3(JOINT_THREADS), lds_bytes, s, w, p, comp_level_begin, level_offsets, rec, comp_bodies);
}
template <class T> void launch_writeback_joint_forces(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_joints) hipLaunchKernelGGL(k_writeback_joint_forces<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w, p);
}

#define INST(T)                                                                                    \
    template void launch_prepare_joints<T>(const DW<T>&, hipStream_t);                    \
    template void launch_joint_schedule<T>(const DW<T>&, const StepParams<T>&, int, uint32_t, const uint32_t*, const uint32_t*, const int4*, hipStream_t); \
    template void launch_joint_schedule_lds<T>(const DW<T>&, const StepParams<T>&, uint32_t, const uint32_t*, const uint32_t*, const int4*, const uint32_t*, uint32_t, hipStream_t); \
    template void launch_writeback_joint_forces<T>(const DW<T>&, const StepParams<T>&, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
