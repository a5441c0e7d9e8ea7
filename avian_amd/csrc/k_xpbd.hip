// k_xpbd.hip — XPBD DistanceJoint projection, joint damping, joint forces.
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
//   k_prepare_distance_joints   solver/xpbd/plugin.rs:125-142, solver/xpbd/joints/distance.rs:36-59
//   k_xpbd_distance_joints      solver/xpbd/plugin.rs:145-189, joints/distance.rs:61-117,
//                               joints/mod.rs:321-340, xpbd/mod.rs:393-413, xpbd/positional_constraint.rs:10-93
//   k_joint_damping             solver/plugin.rs:759-806
//   k_writeback_joint_forces    solver/xpbd/plugin.rs:242-260
#include "avn_kernels.h"

namespace avn {

#define JOINT_THREADS 64

template <class T>
__global__ __launch_bounds__(256) void k_prepare_distance_joints(DW<T> w) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    w.j_lag[j] = make4<T>(0, 0, 0, 0);  // clear_lagrange_multipliers
    int2 b = w.j_bodies[j];
    uint32_t m1 = w.bmeta[b.x], m2 = w.bmeta[b.y];
    if ((meta_flags(m1) | meta_flags(m2)) & AVN_BODY_DISABLED) return;  // bodies.get_many fails: solver data untouched
    Q4<T> q1 = quat<T>(w.rot[b.x]), q2 = quat<T>(w.rot[b.y]);
    V3<T> com1 = xyz<T>(w.com[b.x]), com2 = xyz<T>(w.com[b.y]);
    V3<T> p1 = xyz<T>(w.pos[b.x]), p2 = xyz<T>(w.pos[b.y]);
    V3<T> world_r1 = qrot(q1, xyz<T>(w.j_a1[j]) - com1);
    V3<T> world_r2 = qrot(q2, xyz<T>(w.j_a2[j]) - com2);
    V3<T> cd = (p2 - p1) + (qrot(q2, com2) - qrot(q1, com1));
    w.j_r1[j] = make4<T>(world_r1, 0);
    w.j_r2[j] = make4<T>(world_r2, 0);
    w.j_cd[j] = make4<T>(cd, 0);
}

template <class T> struct JBody {
    V3<T> dp; Q4<T> dq; V3<T> inv_mass; Sym3<T> I; T dp_w;
};
template <class T> __device__ __forceinline__ void jload(const DW<T>& w, int idx, bool nobody, bool dummy_inertia, JBody<T>& b) {
    if (nobody) { b.dp = vzero<T>(); b.dq = qidentity<T>(); b.inv_mass = vzero<T>(); b.I = sym_zero<T>(); b.dp_w = 0; return; }
    Vec4<T> d = w.sb_dp[idx];
    b.dp = xyz<T>(d); b.dp_w = d.w; b.dq = quat<T>(w.sb_dq[idx]);
    if (dummy_inertia) { b.inv_mass = vzero<T>(); b.I = sym_zero<T>(); }
    else {
        Vec4<T> sa = w.si_a[idx], sb = w.si_b[idx];
        b.inv_mass = effective_inv_mass<T>(sa.x, scalar_to_bits(sb.w));
        b.I = Sym3<T>{sa.y, sa.z, sa.w, sb.x, sb.y, sb.z};
    }
}

template <class T> __device__ __forceinline__ void distance_joint_solve(const DW<T>& w, const StepParams<T>& p, uint32_t j) {
    int2 b = w.j_bodies[j];
    uint32_t f1 = w.sb_flags[b.x], f2 = w.sb_flags[b.y];
    bool nobody1 = f1 & AVN_SBF_NO_SOLVER_BODY, nobody2 = f2 & AVN_SBF_NO_SOLVER_BODY;
    // dominance of the (possibly DUMMY) inertias; DUMMY rows carry dominance 128
    int dom1 = (int)(int16_t)(scalar_to_bits(w.si_b[b.x].w) >> 16), dom2 = (int)(int16_t)(scalar_to_bits(w.si_b[b.y].w) >> 16);
    int rel = dom1 - dom2;
    JBody<T> b1, b2;
    jload<T>(w, b.x, nobody1, rel > 0, b1);
    jload<T>(w, b.y, nobody2, rel < 0, b2);
    Vec4<T> a1 = w.j_a1[j], a2 = w.j_a2[j];
    T limit_min = a1.w, limit_max = a2.w, compliance = w.j_par[j].x;
    V3<T> world_r1 = qrot(b1.dq, xyz<T>(w.j_r1[j]));
    V3<T> world_r2 = qrot(b2.dq, xyz<T>(w.j_r2[j]));
    V3<T> separation = ((b2.dp - b1.dp) + (world_r2 - world_r1)) + xyz<T>(w.j_cd[j]);
    // DistanceLimit::compute_correction
    V3<T> dir = vzero<T>();
    T distance = 0;
    T dsq = length_squared(separation);
    if (!(dsq <= Limits<T>::eps)) {
        T d = sqrt_t(dsq);
        if (d < limit_min) { dir = separation / d; distance = limit_min - d; }
        else if (d > limit_max) { dir = (-separation) / d; distance = d - limit_max; }
    }
    if (distance <= Limits<T>::eps) return;
    V3<T> rc1 = cross(world_r1, dir);
    T w1 = max_element(b1.inv_mass) + dot(rc1, smul(b1.I, rc1));
    V3<T> rc2 = cross(world_r2, dir);
    T w2 = max_element(b2.inv_mass) + dot(rc2, smul(b2.I, rc2));
    // compute_lagrange_update(lagrange = 0, c = distance, [w1, w2], compliance, dt)
    T w_sum = T(0) + w1 + w2;
    T delta_lagrange = T(0);
    if (!(w_sum <= Limits<T>::eps)) {
        T dt = p.h_adj;
        T tilde_compliance = compliance / (dt * dt);
        delta_lagrange = (-distance - tilde_compliance * T(0)) / (w_sum + tilde_compliance);
    }
    V3<T> impulse = delta_lagrange * dir;
    Vec4<T> lag = w.j_lag[j];
    w.j_lag[j] = make4<T>(xyz<T>(lag) + impulse, lag.w);
    // apply_positional_impulse
    b1.dp = b1.dp + cmul(impulse, b1.inv_mass);
    b1.dq = qmul(from_scaled_axis(smul(b1.I, cross(world_r1, impulse))), b1.dq);
    b2.dp = b2.dp - cmul(impulse, b2.inv_mass);
    b2.dq = qmul(from_scaled_axis(smul(b2.I, cross(world_r2, -impulse))), b2.dq);
    if (!nobody1) { w.sb_dp[b.x] = make4<T>(b1.dp, b1.dp_w); w.sb_dq[b.x] = make4<T>(b1.dq); }
    if (!nobody2) { w.sb_dp[b.y] = make4<T>(b2.dp, b2.dp_w); w.sb_dq[b.y] = make4<T>(b2.dq); }
}

template <class T> __device__ __forceinline__ void joint_damping_one(const DW<T>& w, const StepParams<T>& p, uint32_t j) {
    Vec4<T> par = w.j_par[j];
    if (!(scalar_to_bits(par.w) & 1u)) return;  // no JointDamping component
    int2 b = w.j_bodies[j];
    uint32_t f1 = w.sb_flags[b.x], f2 = w.sb_flags[b.y];
    bool nobody1 = f1 & AVN_SBF_NO_SOLVER_BODY, nobody2 = f2 & AVN_SBF_NO_SOLVER_BODY;
    T delta_secs = p.h_adj;
    // Missing bodies use the two DUMMY SolverBodies that the reference declares OUTSIDE its joint loop
    // (solver/plugin.rs:766-767): they are shared by all joints and their angular velocity is mutated, so they
    // live in two virtual body slots (n_bodies, n_bodies + 1) that the host resets before every launch and that
    // the damping schedule treats as ordinary bodies (=> joints touching them are serialised, as in the reference).
    int i1 = nobody1 ? (int)w.n_bodies : b.x, i2 = nobody2 ? (int)w.n_bodies + 1 : b.y;
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

// One workgroup per joint component; levels separated by workgroup barriers.
//   comp_level_begin[c] .. comp_level_begin[c+1]  : level slots of component c
//   level_offsets[l] .. level_offsets[l+1]        : joints (via `order`) of level slot l
template <class T, int OP>
__global__ __launch_bounds__(JOINT_THREADS) void k_joint_schedule(DW<T> w, StepParams<T> p, const uint32_t* __restrict__ comp_level_begin,
                                                                   const uint32_t* __restrict__ level_offsets,
                                                                   const uint32_t* __restrict__ order) {
    uint32_t c = blockIdx.x;
    uint32_t l0 = comp_level_begin[c], l1 = comp_level_begin[c + 1];
    for (uint32_t l = l0; l < l1; ++l) {
        uint32_t j0 = level_offsets[l], j1 = level_offsets[l + 1];
        for (uint32_t k = j0 + threadIdx.x; k < j1; k += JOINT_THREADS) {
            uint32_t j = order[k];
            if (OP == 0) distance_joint_solve<T>(w, p, j);
            else joint_damping_one<T>(w, p, j);
        }
        __syncthreads();
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_writeback_joint_forces(DW<T> w, StepParams<T> p) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    T delta_secs = p.dt_adj;
    T rhs = recip_or_zero(delta_secs * delta_secs) * p.substeps_as_scalar;
    w.j_force[j] = make4<T>(xyz<T>(w.j_lag[j]) * rhs, 0);
}

template <class T> void launch_prepare_distance_joints(const DW<T>& w, hipStream_t s) {
    if (w.n_joints) hipLaunchKernelGGL(k_prepare_distance_joints<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w);
}
template <class T> void launch_joint_schedule(const DW<T>& w, const StepParams<T>& p, int op, uint32_t n_components,
                                              const uint32_t* comp_level_begin, const uint32_t* level_offsets, const uint32_t* order, hipStream_t s) {
    if (!w.n_joints || !n_components) return;
    if (op == 0) hipLaunchKernelGGL((k_joint_schedule<T, 0>), dim3(n_components), dim3(JOINT_THREADS), 0, s, w, p, comp_level_begin, level_offsets, order);
    else hipLaunchKernelGGL((k_joint_schedule<T, 1>), dim3(n_components), dim3(JOINT_THREADS), 0, s, w, p, comp_level_begin, level_offsets, order);
}
template <class T> void launch_writeback_joint_forces(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_joints) hipLaunchKernelGGL(k_writeback_joint_forces<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w, p);
}

#define INST(T)                                                                                    \
    template void launch_prepare_distance_joints<T>(const DW<T>&, hipStream_t);                    \
    template void launch_joint_schedule<T>(const DW<T>&, const StepParams<T>&, int, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*, hipStream_t); \
    template void launch_writeback_joint_forces<T>(const DW<T>&, const StepParams<T>&, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
