// k_bodies.hip — per-body streaming kernels (one thread per rigid body, coalesced Vec4 records).
// HBM-bound: each kernel reads/writes a handful of 16-byte records per body and does O(100) flops.
//
// Reference systems replaced (paths relative to /root/reference/src):
//   k_prepare_solver_bodies      dynamics/solver/solver_body/plugin.rs:173-251 (+ mod.rs:378-423)
//   k_pre_process_increments     dynamics/integrator/mod.rs:260-313
//   k_integrate_velocities       dynamics/integrator/mod.rs:343-391, 403-460, 467-500
//   k_integrate_positions        dynamics/integrator/mod.rs:503-535 + solver_body/plugin.rs:287-295
//   k_clear_increments           dynamics/integrator/mod.rs:316-328
//   k_writeback_solver_bodies    dynamics/solver/solver_body/plugin.rs:255-284
//   k_xpbd_snapshot / k_xpbd_velocity_projection   dynamics/solver/xpbd/plugin.rs:61-76, 192-240
#include "avn_kernels.h"
#include "avn_body_ops.h"

namespace avn {

#define BODY_THREADS 256

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_prepare_solver_bodies(DW<T> w) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    uint32_t meta = w.bmeta[i];
    if (!meta_has_solver_body(meta)) {
        // No SolverBody component: constraints see SolverBody::DUMMY / SolverBodyInertia::DUMMY.
        w.sb_flags[i] = AVN_SBF_NO_SOLVER_BODY;
        w.sb_lin[i] = make4<T>(0, 0, 0, 0);
        w.sb_ang[i] = make4<T>(0, 0, 0, 0);
        w.sb_dp[i] = make4<T>(0, 0, 0, 0);
        w.sb_dq[i] = make4<T>(0, 0, 0, 1);
        w.si_a[i] = make4<T>(0, 0, 0, 0);
        w.si_b[i] = make4<T>(0, 0, 0, bits_to_scalar(0xC0u | (128u << 16), T(0)));
        return;
    }
    Vec4<T> pos = w.pos[i], rot = w.rot[i], lv = w.lvel[i], av = w.avel[i], ia = w.iloc_a[i], ib = w.iloc_b[i];
    uint32_t locked = meta_locked(meta);
    Sym3<T> local{ia.x, ia.y, ia.z, ia.w, ib.x, ib.y};
    Sym3<T> inv_inertia = rotated_inverse_inertia(local, quat<T>(rot));
    uint32_t iflags = locked;
    T inv_mass = pos.w;
    if (inv_mass == T(0)) iflags |= 1u << 6;
    if (sym_is_zero(inv_inertia)) iflags |= 1u << 7;
    lock_rotation_axes(inv_inertia, locked);
    int dominance = meta_rb_type(meta) == AVN_RB_DYNAMIC ? meta_dominance(meta) : 128;
    uint32_t sbf = locked;
    if (meta_rb_type(meta) == AVN_RB_KINEMATIC) sbf |= AVN_SB_KINEMATIC;
    bool rotation_locked = (locked & 7u) == 7u;
    if (!rotation_locked && !sym_is_isotropic(local, T(1e-6))) sbf |= AVN_SB_GYROSCOPIC;
    w.sb_lin[i] = make4<T>(lv.x, lv.y, lv.z, 0);
    w.sb_ang[i] = make4<T>(av.x, av.y, av.z, 0);
    w.sb_dp[i] = make4<T>(0, 0, 0, 0);
    w.sb_dq[i] = make4<T>(0, 0, 0, 1);
    w.si_a[i] = make4<T>(inv_mass, inv_inertia.m00, inv_inertia.m01, inv_inertia.m02);
    w.si_b[i] = make4<T>(inv_inertia.m11, inv_inertia.m12, inv_inertia.m22,
                         bits_to_scalar(iflags | ((uint32_t)(uint16_t)(int16_t)dominance << 16), T(0)));
    w.sb_flags[i] = sbf;
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_pre_process_increments(DW<T> w, StepParams<T> p) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    uint32_t meta = w.bmeta[i];
    if (meta_rb_type(meta) != AVN_RB_DYNAMIC) return;
    T delta_secs = p.h_f64cast;
    uint32_t locked = meta_locked(meta);
    Vec4<T> lv = w.lvel[i], av = w.avel[i], cm = w.com[i];
    T gravity_scale = lv.w, lin_damping = av.w, ang_damping = cm.w;
    V3<T> lin = xyz<T>(w.acc_l[i]);
    V3<T> ang = xyz<T>(w.acc_a[i]);
    T lin_rhs = T(1) / (T(1) + delta_secs * lin_damping);
    T ang_rhs = T(1) / (T(1) + delta_secs * ang_damping);
    V3<T> gravity{p.gravity[0], p.gravity[1], p.gravity[2]};
    lin = lin + gravity * gravity_scale;
    if (locked & 0x20u) lin.x = 0;
    if (locked & 0x10u) lin.y = 0;
    if (locked & 0x08u) lin.z = 0;
    if (locked & 0x04u) ang.x = 0;
    if (locked & 0x02u) ang.y = 0;
    if (locked & 0x01u) ang.z = 0;
    lin = lin * delta_secs;
    ang = ang * delta_secs;
    w.vid_l[i] = make4<T>(lin, lin_rhs);
    w.vid_a[i] = make4<T>(ang, ang_rhs);
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_clear_increments(DW<T> w) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (w.sb_flags[i] & AVN_SBF_NO_SOLVER_BODY) return;
    Vec4<T> l = w.vid_l[i], a = w.vid_a[i];
    w.vid_l[i] = make4<T>(0, 0, 0, l.w);
    w.vid_a[i] = make4<T>(0, 0, 0, a.w);
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_integrate_velocities(DW<T> w, StepParams<T> p) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (!body_in_group(w, i)) return;
    uint32_t sbf = w.sb_flags[i];
    if (sbf & AVN_SBF_NO_SOLVER_BODY) return;
    Vec4<T> l4 = w.sb_lin[i], a4 = w.sb_ang[i];
    V3<T> v = xyz<T>(l4), om = xyz<T>(a4);
    if (integrate_velocities_one<T>(w, p, i, sbf, v, om, &w.sb_dq[i])) {
        w.sb_lin[i] = make4<T>(v, l4.w);
        w.sb_ang[i] = make4<T>(om, a4.w);
    }
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_integrate_positions(DW<T> w, StepParams<T> p) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (!body_in_group(w, i)) return;
    uint32_t sbf = w.sb_flags[i];
    if (sbf & AVN_SBF_NO_SOLVER_BODY) return;
    V3<T> v = xyz<T>(w.sb_lin[i]), om = xyz<T>(w.sb_ang[i]);
    Vec4<T> dp4 = w.sb_dp[i], dq4 = w.sb_dq[i], sa = w.si_a[i], sbv = w.si_b[i];
    integrate_positions_one<T>(w, p, i, v, om, dp4, dq4, sa, sbv);
    w.sb_dp[i] = dp4; w.sb_dq[i] = dq4;
    w.si_a[i] = sa; w.si_b[i] = sbv;
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_writeback_solver_bodies(DW<T> w) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (w.sb_flags[i] & AVN_SBF_NO_SOLVER_BODY) return;
    Vec4<T> pos = w.pos[i], lv = w.lvel[i], av = w.avel[i];
    Q4<T> rot = quat<T>(w.rot[i]);
    V3<T> com = xyz<T>(w.com[i]);
    V3<T> old_world_com = qrot(rot, com);
    Q4<T> new_rot = fast_renormalize(qmul(quat<T>(w.sb_dq[i]), rot));
    V3<T> new_world_com = qrot(new_rot, com);
    V3<T> np = xyz<T>(pos) + ((xyz<T>(w.sb_dp[i]) + old_world_com) - new_world_com);
    w.pos[i] = make4<T>(np, pos.w);
    w.rot[i] = make4<T>(new_rot);
    w.lvel[i] = make4<T>(xyz<T>(w.sb_lin[i]), lv.w);
    w.avel[i] = make4<T>(xyz<T>(w.sb_ang[i]), av.w);
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_xpbd_snapshot(DW<T> w) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (!body_in_group(w, i)) return;
    if (w.sb_flags[i] & AVN_SBF_NO_SOLVER_BODY) return;
    w.pre_dp[i] = w.sb_dp[i];
    w.pre_dq[i] = w.sb_dq[i];
}

template <class T>
__global__ __launch_bounds__(BODY_THREADS) void k_xpbd_velocity_projection(DW<T> w, StepParams<T> p) {
    uint32_t i = blockIdx.x * BODY_THREADS + threadIdx.x;
    if (i >= w.n_bodies) return;
    if (!body_in_group(w, i)) return;
    if (w.sb_flags[i] & AVN_SBF_NO_SOLVER_BODY) return;
    T delta_secs = p.h_adj;
    Vec4<T> l4 = w.sb_lin[i], a4 = w.sb_ang[i];
    V3<T> new_lin_vel = (xyz<T>(w.sb_dp[i]) - xyz<T>(w.pre_dp[i])) / delta_secs;
    V3<T> v = xyz<T>(l4) + new_lin_vel;
    Q4<T> delta_rot = qmul(quat<T>(w.sb_dq[i]), qinverse(quat<T>(w.pre_dq[i])));
    V3<T> new_ang_vel = (T(2) * V3<T>{delta_rot.x, delta_rot.y, delta_rot.z}) / delta_secs;
    if (delta_rot.w < T(0)) new_ang_vel = -new_ang_vel;
    V3<T> om = xyz<T>(a4) + new_ang_vel;
    w.sb_lin[i] = make4<T>(v, l4.w);
    w.sb_ang[i] = make4<T>(om, a4.w);
}

// ---- launchers ------------------------------------------------------------------------------------
static inline dim3 body_grid(uint32_t n) { return dim3((n + BODY_THREADS - 1) / BODY_THREADS); }

template <class T> void launch_prepare_solver_bodies(const DW<T>& w, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_prepare_solver_bodies<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w);
}
template <class T> void launch_pre_process_increments(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_pre_process_increments<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w, p);
}
template <class T> void launch_clear_increments(const DW<T>& w, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_clear_increments<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w);
}
template <class T> void launch_integrate_velocities(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_integrate_velocities<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w, p);
}
template <class T> void launch_integrate_positions(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_integrate_positions<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w, p);
}
template <class T> void launch_writeback_solver_bodies(const DW<T>& w, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_writeback_solver_bodies<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w);
}
template <class T> void launch_xpbd_snapshot(const DW<T>& w, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_xpbd_snapshot<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w);
}
template <class T> void launch_xpbd_velocity_projection(const DW<T>& w, const StepParams<T>& p, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_xpbd_velocity_projection<T>, body_grid(w.n_bodies), dim3(BODY_THREADS), 0, s, w, p);
}

#define INST(T)                                                                                   \
    template void launch_prepare_solver_bodies<T>(const DW<T>&, hipStream_t);                     \
    template void launch_pre_process_increments<T>(const DW<T>&, const StepParams<T>&, hipStream_t); \
    template void launch_clear_increments<T>(const DW<T>&, hipStream_t);                          \
    template void launch_integrate_velocities<T>(const DW<T>&, const StepParams<T>&, hipStream_t); \
    template void launch_integrate_positions<T>(const DW<T>&, const StepParams<T>&, hipStream_t);  \
    template void launch_writeback_solver_bodies<T>(const DW<T>&, hipStream_t);                   \
    template void launch_xpbd_snapshot<T>(const DW<T>&, hipStream_t);                             \
    template void launch_xpbd_velocity_projection<T>(const DW<T>&, const StepParams<T>&, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
