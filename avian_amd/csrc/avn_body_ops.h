// avn_body_ops.h — per-body device functions shared by the body kernels (k_bodies.hip) and the fused
// integrate-velocities + warm-start kernel (k_contacts.hip).
#pragma once
#include "avn_device.h"

namespace avn {

// reference dynamics/integrator/mod.rs:403-460
template <class T>
__device__ __forceinline__ V3<T> solve_gyroscopic_torque(V3<T> ang_vel, Q4<T> rotation, const Sym3<T>& local_inverse_inertia, T delta_secs) {
    V3<T> local_ang_vel = qrot(qinverse(rotation), ang_vel);
    Sym3<T> tensor = sym_inverse_or_zero(local_inverse_inertia);
    V3<T> local_momentum = smul(tensor, local_ang_vel);
    V3<T> new_local_momentum = local_momentum - delta_secs * cross(local_ang_vel, local_momentum);
    T new_len_sq = length_squared(new_local_momentum);
    if (new_len_sq == T(0)) return vzero<T>();
    new_local_momentum = new_local_momentum * sqrt_t(length_squared(local_momentum) / new_len_sq);
    return qrot(rotation, smul(local_inverse_inertia, new_local_momentum));
}

// integrate_velocities + clamp_velocities for ONE body that has a SolverBody (reference dynamics/integrator/mod.rs:343-391,
// 467-500); v / om are the SolverBody velocities, updated in place; returns true when they were (re)written.
// `dq`: where the body's current SolverBody::delta_rotation lives (HBM or an LDS block); only the gyroscopic branch reads it.
template <class T>
__device__ __forceinline__ bool integrate_velocities_one(const DW<T>& w, const StepParams<T>& p, uint32_t i, uint32_t sbf, V3<T>& v, V3<T>& om, const Vec4<T>* dq) {
    uint32_t meta = w.bmeta[i];
    bool touched = false;
    // apply_local_acceleration (forces/plugin.rs:207-241; in IntegrationSystems::Velocity IN FRONT of integrate_velocities, :34-38): every body with a
    // SolverBody and no CustomVelocityIntegration, kinematic ones included; LockedAxes::apply_to_vec (the translation locks, locked_axes.rs:230-243)
    // masks BOTH world-space vectors, as written there.  A per-body system: running it here is the reference's order for this body.
    if (w.lacc_l && !(meta_flags(meta) & AVN_BODY_CUSTOM_VELOCITY_INTEGRATION)) {
        Q4<T> rotation = qmul(quat<T>(*dq), quat<T>(w.rot[i]));
        V3<T> wl = qrot(rotation, xyz<T>(w.lacc_l[i])), wa = qrot(rotation, xyz<T>(w.lacc_a[i]));
        uint32_t locked = meta_locked(meta);
        if (locked & 0x20u) { wl.x = T(0); wa.x = T(0); }
        if (locked & 0x10u) { wl.y = T(0); wa.y = T(0); }
        if (locked & 0x08u) { wl.z = T(0); wa.z = T(0); }
        v = v + wl * p.h_f64cast;
        om = om + wa * p.h_f64cast;
        touched = true;
    }
    if (!(meta_flags(meta) & AVN_BODY_CUSTOM_VELOCITY_INTEGRATION) && !(sbf & AVN_SB_KINEMATIC)) {
        Vec4<T> il = w.vid_l[i], ia = w.vid_a[i];
        v = v * il.w;
        om = om * ia.w;
        v = v + xyz<T>(il);
        om = om + xyz<T>(ia);
        if (sbf & AVN_SB_GYROSCOPIC) {
            Vec4<T> la = w.iloc_a[i], lb = w.iloc_b[i];
            Sym3<T> local{la.x, la.y, la.z, la.w, lb.x, lb.y};
            Q4<T> rotation = qmul(quat<T>(*dq), quat<T>(w.rot[i]));
            om = solve_gyroscopic_torque(om, rotation, local, p.h_f64cast);
        }
        touched = true;
    }
    // clamp_velocities (MaxLinearSpeed / MaxAngularSpeed; negative = component absent)
    Vec4<T> lb = w.iloc_b[i];
    T max_lin = lb.z, max_ang = lb.w;
    if (max_lin >= T(0)) {
        T sq = length_squared(v);
        if (sq > max_lin * max_lin) { v = v * (max_lin / sqrt_t(sq)); touched = true; }
    }
    if (max_ang >= T(0)) {
        T sq = length_squared(om);
        if (sq > max_ang * max_ang) { om = om * (max_ang / sqrt_t(sq)); touched = true; }
    }
    return touched;
}

// integrate_positions (reference dynamics/integrator/mod.rs:503-535) + update_solver_body_angular_inertia
// (solver/solver_body/plugin.rs:287-295: recomputed from the STEP-START Rotation every substep) for one body that has a
// SolverBody; all four records are updated in place (the caller stores them where they live).
template <class T>
__device__ __forceinline__ void integrate_positions_one(const DW<T>& w, const StepParams<T>& p, uint32_t i, V3<T> v, V3<T> om, Vec4<T>& dp4, Vec4<T>& dq4,
                                                        Vec4<T>& sa, Vec4<T>& sb) {
    uint32_t meta = w.bmeta[i];
    T delta_secs = p.h_adj;
    if (!(meta_flags(meta) & AVN_BODY_CUSTOM_POSITION_INTEGRATION)) {
        V3<T> dp = xyz<T>(dp4) + v * delta_secs;
        Q4<T> dq = qmul(from_scaled_axis(om * delta_secs), quat<T>(dq4));
        dp4 = make4<T>(dp, dp4.w);
        dq4 = make4<T>(dq);
    }
    Vec4<T> la = w.iloc_a[i], lb = w.iloc_b[i];
    Sym3<T> local{la.x, la.y, la.z, la.w, lb.x, lb.y};
    Sym3<T> t = rotated_inverse_inertia(local, quat<T>(w.rot[i]));
    uint32_t iflags = scalar_to_bits(sb.w);
    lock_rotation_axes(t, iflags & 0x3Fu);
    sa = make4<T>(sa.x, t.m00, t.m01, t.m02);
    sb = make4<T>(t.m11, t.m12, t.m22, sb.w);
}

}  // namespace avn
