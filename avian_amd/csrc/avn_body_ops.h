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
template <class T>
__device__ __forceinline__ bool integrate_velocities_one(const DW<T>& w, const StepParams<T>& p, uint32_t i, uint32_t sbf, V3<T>& v, V3<T>& om) {
    uint32_t meta = w.bmeta[i];
    bool touched = false;
    if (!(meta_flags(meta) & AVN_BODY_CUSTOM_VELOCITY_INTEGRATION) && !(sbf & AVN_SB_KINEMATIC)) {
        Vec4<T> il = w.vid_l[i], ia = w.vid_a[i];
        v = v * il.w;
        om = om * ia.w;
        v = v + xyz<T>(il);
        om = om + xyz<T>(ia);
        if (sbf & AVN_SB_GYROSCOPIC) {
            Vec4<T> la = w.iloc_a[i], lb = w.iloc_b[i];
            Sym3<T> local{la.x, la.y, la.z, la.w, lb.x, lb.y};
            Q4<T> rotation = qmul(quat<T>(w.sb_dq[i]), quat<T>(w.rot[i]));
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

}  // namespace avn
