// k_transfer.hip — layout conversion between the ABI's host arrays (interleaved xyz per entity, i.e. the ECS
// component tables) and the device's 16/32-byte Vec4 records.  Streaming, HBM-bound, once per upload/download.
#include "avn_kernels.h"

namespace avn {

template <class T> __device__ __forceinline__ V3<T> ld3(const T* p, size_t i) { return p ? V3<T>{p[3 * i], p[3 * i + 1], p[3 * i + 2]} : vzero<T>(); }
template <class T> __device__ __forceinline__ void st3(T* p, size_t i, V3<T> v) { if (p) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; } }
template <class T> __device__ __forceinline__ void st4(T* p, size_t i, Vec4<T> v) { if (p) { p[4 * i] = v.x; p[4 * i + 1] = v.y; p[4 * i + 2] = v.z; p[4 * i + 3] = v.w; } }

template <class T>
__global__ __launch_bounds__(256) void k_pack_bodies(DW<T> w, BodyStage<T> s) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n_bodies) return;
    w.pos[i] = make4<T>(ld3(s.position, i), s.inv_mass[i]);
    w.rot[i] = make4<T>(s.rotation[4 * i], s.rotation[4 * i + 1], s.rotation[4 * i + 2], s.rotation[4 * i + 3]);
    w.lvel[i] = make4<T>(ld3(s.linear_velocity, i), s.gravity_scale ? s.gravity_scale[i] : T(1));
    w.avel[i] = make4<T>(ld3(s.angular_velocity, i), s.linear_damping ? s.linear_damping[i] : T(0));
    w.com[i] = make4<T>(ld3(s.center_of_mass, i), s.angular_damping ? s.angular_damping[i] : T(0));
    const T* t = s.inv_inertia_local + 6 * (size_t)i;
    w.iloc_a[i] = make4<T>(t[0], t[1], t[2], t[3]);
    w.iloc_b[i] = make4<T>(t[4], t[5], s.max_linear_speed ? s.max_linear_speed[i] : T(-1), s.max_angular_speed ? s.max_angular_speed[i] : T(-1));
    w.acc_l[i] = make4<T>(ld3(s.accel_linear, i), 0);
    w.acc_a[i] = make4<T>(ld3(s.accel_angular, i), 0);
    uint32_t meta = (uint32_t)s.rb_type[i] & 3u;
    if (s.locked_axes) meta |= ((uint32_t)s.locked_axes[i] & 0x3Fu) << 8;
    if (s.body_flags) meta |= (uint32_t)s.body_flags[i] << 16;
    if (s.dominance) meta |= (uint32_t)(uint8_t)s.dominance[i] << 24;
    w.bmeta[i] = meta;
    // fresh components: SolverBody::default(), SolverBodyInertia::DUMMY, VelocityIntegrationData::default()
    w.sb_lin[i] = make4<T>(0, 0, 0, 0);
    w.sb_ang[i] = make4<T>(0, 0, 0, 0);
    w.sb_dp[i] = make4<T>(0, 0, 0, 0);
    w.sb_dq[i] = make4<T>(0, 0, 0, 1);
    w.si_a[i] = make4<T>(0, 0, 0, 0);
    w.si_b[i] = make4<T>(0, 0, 0, bits_to_scalar(0xC0u | (128u << 16), T(0)));
    w.vid_l[i] = make4<T>(0, 0, 0, 0);
    w.vid_a[i] = make4<T>(0, 0, 0, 0);
    w.pre_dp[i] = make4<T>(0, 0, 0, 0);
    w.pre_dq[i] = make4<T>(0, 0, 0, 1);
    w.sb_flags[i] = meta_has_solver_body(meta) ? 0u : AVN_SBF_NO_SOLVER_BODY;
}

// avn_local_accelerations_upload: AccumulatedLocalAcceleration::{linear, angular} per body (either array may be absent = zero)
template <class T>
__global__ __launch_bounds__(256) void k_pack_local_accelerations(Vec4<T>* lin, Vec4<T>* ang, const T* s_lin, const T* s_ang, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    lin[i] = make4<T>(ld3(s_lin, i), 0);
    ang[i] = make4<T>(ld3(s_ang, i), 0);
}

template <class T>
__global__ __launch_bounds__(256) void k_pack_manifolds(DW<T> w, ManifoldStage<T> s) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.n_manifolds) return;
    w.m_bodies[m] = make_int2(s.body1[m], s.body2[m]);
    w.m_n[m] = make4<T>(ld3(s.normal, m), s.friction[m]);
    w.m_tv[m] = make4<T>(ld3(s.tangent_velocity, m), s.restitution[m]);
    uint32_t np = s.point_count[m];
    uint32_t fl = s.manifold_flags ? s.manifold_flags[m] : (uint32_t)AVN_MANIFOLD_GENERATES_CONSTRAINTS;
    w.m_meta[m] = np | (fl << 8);
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        size_t src = 4 * (size_t)m + k;
        size_t dst = (size_t)k * w.m_stride + m;
        w.mp_a1[dst] = make4<T>(ld3(s.anchor1, src), s.penetration[src]);
        w.mp_a2[dst] = make4<T>(ld3(s.anchor2, src), s.normal_speed[src]);
        w.mp_w[dst] = make4<T>(s.warm_n ? s.warm_n[src] : T(0), s.warm_t ? s.warm_t[2 * src] : T(0), s.warm_t ? s.warm_t[2 * src + 1] : T(0), T(0));
    }
    w.c_h1[m] = make4<T>(0, 0, 0, bits_to_scalar(0u, T(0)));  // no constraint until prepare runs
    w.c_reldom[m] = 0;
}

template <class T>
__global__ __launch_bounds__(256) void k_pack_joints(DW<T> w, JointStage<T> s) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    w.j_bodies[j] = make_int2(s.body1[j], s.body2[j]);
    uint32_t type = s.joint_type[j], lf = s.limit_flags ? s.limit_flags[j] : 0u;
    w.j_a1[j] = make4<T>(ld3(s.local_anchor1, j), s.limit_min ? s.limit_min[j] : T(0));
    w.j_a2[j] = make4<T>(ld3(s.local_anchor2, j), s.limit_max ? s.limit_max[j] : T(0));
    bool damp = s.damping_linear && s.damping_angular;
    w.j_par[j] = make4<T>(s.compliance[3 * j], damp ? s.damping_linear[j] : T(0), damp ? s.damping_angular[j] : T(0),
                          bits_to_scalar((damp ? 1u : 0u) | (type << 8) | (lf << 16), T(0)));
    w.j_b1[j] = s.local_basis1 ? make4<T>(s.local_basis1[4 * j], s.local_basis1[4 * j + 1], s.local_basis1[4 * j + 2], s.local_basis1[4 * j + 3]) : make4<T>(0, 0, 0, 1);
    w.j_b2[j] = s.local_basis2 ? make4<T>(s.local_basis2[4 * j], s.local_basis2[4 * j + 1], s.local_basis2[4 * j + 2], s.local_basis2[4 * j + 3]) : make4<T>(0, 0, 0, 1);
    V3<T> ax = s.axis ? ld3(s.axis, j)
                      : (type == AVN_JOINT_REVOLUTE ? V3<T>{T(0), T(0), T(1)} : type == AVN_JOINT_SPHERICAL ? V3<T>{T(0), T(1), T(0)} : V3<T>{T(1), T(0), T(0)});
    w.j_ax[j] = make4<T>(ax, s.compliance[3 * j + 1]);
    w.j_l2[j] = make4<T>(s.limit2_min ? s.limit2_min[j] : T(0), s.limit2_max ? s.limit2_max[j] : T(0), s.compliance[3 * j + 2], T(0));
    Vec4<T> z = make4<T>(0, 0, 0, 0);
    w.j_r1[j] = z; w.j_r2[j] = z; w.j_cd[j] = z; w.j_lag[j] = z;
    w.j_s0[j] = make4<T>(0, 0, 0, 1); w.j_s1[j] = z; w.j_s2[j] = z; w.j_s3[j] = z;
    w.j_rl0[j] = z; w.j_rl1[j] = z; w.j_force[j] = z; w.j_torque[j] = z;
}

template <class T>
__global__ __launch_bounds__(256) void k_pack_colliders(BP<T> bp, ColliderStage<T> s) {
    uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= bp.n_colliders) return;
    uint32_t cf = s.cflags ? s.cflags[c] : 0u;
    bp.col_info[c] = make_uint4(s.entity[c], (uint32_t)s.body[c], (uint32_t)s.shape[c] | (cf << 8), 0u);
    bp.col_he[c] = make4<T>(ld3(s.half_extents, c), s.collision_margin ? s.collision_margin[c] : T(0));
    bp.col_spec[c] = s.speculative_margin ? s.speculative_margin[c] : T(-1);
    bp.col_layers[c] = make_uint2(s.memberships ? s.memberships[c] : 1u, s.filters ? s.filters[c] : 0xFFFFFFFFu);
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_bodies(DW<T> w, T* position, T* rotation, T* lv, T* av) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n_bodies) return;
    st3(position, i, xyz<T>(w.pos[i]));
    st4<T>(rotation, i, w.rot[i]);
    st3(lv, i, xyz<T>(w.lvel[i]));
    st3(av, i, xyz<T>(w.avel[i]));
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_solver_bodies(DW<T> w, SolverBodiesStage<T> o) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n_bodies) return;
    st3(o.linear_velocity, i, xyz<T>(w.sb_lin[i]));
    st3(o.angular_velocity, i, xyz<T>(w.sb_ang[i]));
    st3(o.delta_position, i, xyz<T>(w.sb_dp[i]));
    st4<T>(o.delta_rotation, i, w.sb_dq[i]);
    if (o.flags) o.flags[i] = w.sb_flags[i];
    Vec4<T> a = w.si_a[i], b = w.si_b[i];
    if (o.inv_mass) o.inv_mass[i] = a.x;
    if (o.inv_inertia_world) { T* t = o.inv_inertia_world + 6 * (size_t)i; t[0] = a.y; t[1] = a.z; t[2] = a.w; t[3] = b.x; t[4] = b.y; t[5] = b.z; }
    if (o.dominance) o.dominance[i] = (int16_t)(scalar_to_bits(b.w) >> 16);
    Vec4<T> l = w.vid_l[i], g = w.vid_a[i];
    st3(o.linear_increment, i, xyz<T>(l));
    st3(o.angular_increment, i, xyz<T>(g));
    if (o.linear_damping_rhs) o.linear_damping_rhs[i] = l.w;
    if (o.angular_damping_rhs) o.angular_damping_rhs[i] = g.w;
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_impulses(DW<T> w, T* warm_n, T* warm_t, T* normal_impulse) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.n_manifolds) return;
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        size_t dst = 4 * (size_t)m + k;
        Vec4<T> v = w.mp_w[(size_t)k * w.m_stride + m];
        if (warm_n) warm_n[dst] = v.x;
        if (warm_t) { warm_t[2 * dst] = v.y; warm_t[2 * dst + 1] = v.z; }
        if (normal_impulse) normal_impulse[dst] = v.w;
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_constraints(DW<T> w, ConstraintsStage<T> o) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.n_manifolds) return;
    Vec4<T> h1 = w.c_h1[m];
    uint32_t cm = scalar_to_bits(h1.w), np = cm & 7u;
    if (o.point_count) o.point_count[m] = (uint8_t)np;
    if (np == 0) return;
    if (o.relative_dominance) o.relative_dominance[m] = w.c_reldom[m];
    if (o.softness_non_dynamic) o.softness_non_dynamic[m] = (cm & AVN_CM_SOFT_ND) ? 1 : 0;
    st3(o.tangent1, m, xyz<T>(h1));
    for (uint32_t k = 0; k < np; ++k) {
        size_t src = (size_t)k * w.m_stride + m, dst = 4 * (size_t)m + k;
        Vec4<T> a = w.c_pa[src], b = w.c_pb[src], c = w.c_pc[src], d = w.c_pd[src];
        st3(o.anchor1, dst, xyz<T>(a));
        if (o.initial_separation) o.initial_separation[dst] = a.w;
        if (o.normal_impulse) o.normal_impulse[dst] = d.x;
        if (o.total_impulse) o.total_impulse[dst] = d.y;
        if (o.normal_effective_mass) o.normal_effective_mass[dst] = b.w;
        if (o.tangent_impulse) { o.tangent_impulse[2 * dst] = d.z; o.tangent_impulse[2 * dst + 1] = d.w; }
        if (o.tangent_k) { o.tangent_k[3 * dst] = c.x; o.tangent_k[3 * dst + 1] = c.y; o.tangent_k[3 * dst + 2] = c.z; }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_joints(DW<T> w, T* r1, T* r2, T* cd, T* lag, T* force, T* rot_lag, T* torque) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w.n_joints) return;
    st3(r1, j, xyz<T>(w.j_r1[j]));
    st3(r2, j, xyz<T>(w.j_r2[j]));
    st3(cd, j, xyz<T>(w.j_cd[j]));
    st3(lag, j, xyz<T>(w.j_lag[j]));
    st3(force, j, xyz<T>(w.j_force[j]));
    st3(rot_lag, j, xyz<T>(w.j_rl0[j]) + xyz<T>(w.j_rl1[j]));
    st3(torque, j, xyz<T>(w.j_torque[j]));
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack_aabbs(BP<T> bp, T* mn, T* mx, uint32_t* interval_entities) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < bp.n_colliders) { st3(mn, i, xyz<T>(bp.aabb_min[i])); st3(mx, i, xyz<T>(bp.aabb_max[i])); }
    if (i < bp.n_intervals && interval_entities) interval_entities[i] = bp.col_info[bp.iv_collider[i]].x;
}

static inline dim3 g256(uint32_t n) { return dim3((n + 255) / 256); }
template <class T> void launch_pack_bodies(const DW<T>& w, const BodyStage<T>& s, hipStream_t st) { if (w.n_bodies) hipLaunchKernelGGL(k_pack_bodies<T>, g256(w.n_bodies), dim3(256), 0, st, w, s); }
template <class T> void launch_pack_manifolds(const DW<T>& w, const ManifoldStage<T>& s, hipStream_t st) { if (w.n_manifolds) hipLaunchKernelGGL(k_pack_manifolds<T>, g256(w.n_manifolds), dim3(256), 0, st, w, s); }
template <class T> void launch_pack_joints(const DW<T>& w, const JointStage<T>& s, hipStream_t st) { if (w.n_joints) hipLaunchKernelGGL(k_pack_joints<T>, g256(w.n_joints), dim3(256), 0, st, w, s); }
template <class T> void launch_pack_colliders(const BP<T>& bp, const ColliderStage<T>& s, hipStream_t st) { if (bp.n_colliders) hipLaunchKernelGGL(k_pack_colliders<T>, g256(bp.n_colliders), dim3(256), 0, st, bp, s); }
template <class T> void launch_pack_local_accelerations(Vec4<T>* lin, Vec4<T>* ang, const T* s_lin, const T* s_ang, uint32_t n, hipStream_t st) { if (n) hipLaunchKernelGGL(k_pack_local_accelerations<T>, g256(n), dim3(256), 0, st, lin, ang, s_lin, s_ang, n); }
template <class T> void launch_unpack_bodies(const DW<T>& w, T* p, T* r, T* l, T* a, hipStream_t st) { if (w.n_bodies) hipLaunchKernelGGL(k_unpack_bodies<T>, g256(w.n_bodies), dim3(256), 0, st, w, p, r, l, a); }
template <class T> void launch_unpack_solver_bodies(const DW<T>& w, const SolverBodiesStage<T>& o, hipStream_t st) { if (w.n_bodies) hipLaunchKernelGGL(k_unpack_solver_bodies<T>, g256(w.n_bodies), dim3(256), 0, st, w, o); }
template <class T> void launch_unpack_impulses(const DW<T>& w, T* a, T* b, T* c, hipStream_t st) { if (w.n_manifolds) hipLaunchKernelGGL(k_unpack_impulses<T>, g256(w.n_manifolds), dim3(256), 0, st, w, a, b, c); }
template <class T> void launch_unpack_constraints(const DW<T>& w, const ConstraintsStage<T>& o, hipStream_t st) { if (w.n_manifolds) hipLaunchKernelGGL(k_unpack_constraints<T>, g256(w.n_manifolds), dim3(256), 0, st, w, o); }
template <class T> void launch_unpack_joints(const DW<T>& w, T* a, T* b, T* c, T* d, T* e, T* f, T* g, hipStream_t st) { if (w.n_joints) hipLaunchKernelGGL(k_unpack_joints<T>, g256(w.n_joints), dim3(256), 0, st, w, a, b, c, d, e, f, g); }
template <class T> void launch_unpack_aabbs(const BP<T>& bp, T* mn, T* mx, uint32_t* ents, hipStream_t st) {
    uint32_t n = bp.n_colliders > bp.n_intervals ? bp.n_colliders : bp.n_intervals;
    if (n) hipLaunchKernelGGL(k_unpack_aabbs<T>, g256(n), dim3(256), 0, st, bp, mn, mx, ents);
}

// ---- level-2 sharding: boundary-body velocities in and out of a contiguous exchange buffer -------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_halo_pack(DW<T> w, const int32_t* __restrict__ bodies, uint32_t n, Vec4<T>* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = bodies[i];
    out[2 * i] = w.sb_lin[b]; out[2 * i + 1] = w.sb_ang[b];
}
template <class T>
__global__ __launch_bounds__(256) void k_halo_unpack(DW<T> w, const int32_t* __restrict__ bodies, uint32_t n, const Vec4<T>* __restrict__ in) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = bodies[i];
    w.sb_lin[b] = in[2 * i]; w.sb_ang[b] = in[2 * i + 1];
}
// the joint slot's records (header: avn_halo_joint_slot_set): the whole SolverBody -- delta position / rotation and both velocities
template <class T>
__global__ __launch_bounds__(256) void k_halo_pack_joint(DW<T> w, const int32_t* __restrict__ bodies, uint32_t n, Vec4<T>* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = bodies[i];
    out[4 * i] = w.sb_dp[b]; out[4 * i + 1] = w.sb_dq[b]; out[4 * i + 2] = w.sb_lin[b]; out[4 * i + 3] = w.sb_ang[b];
}
template <class T>
__global__ __launch_bounds__(256) void k_halo_unpack_joint(DW<T> w, const int32_t* __restrict__ bodies, uint32_t n, const Vec4<T>* __restrict__ in) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = bodies[i];
    w.sb_dp[b] = in[4 * i]; w.sb_dq[b] = in[4 * i + 1]; w.sb_lin[b] = in[4 * i + 2]; w.sb_ang[b] = in[4 * i + 3];
}
template <class T> void launch_halo_pack_joint(const DW<T>& w, const int32_t* bodies, uint32_t n, Vec4<T>* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_halo_pack_joint<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, out);
}
template <class T> void launch_halo_unpack_joint(const DW<T>& w, const int32_t* bodies, uint32_t n, const Vec4<T>* in, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_halo_unpack_joint<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, in);
}
template void launch_halo_pack_joint<float>(const DW<float>&, const int32_t*, uint32_t, Vec4<float>*, hipStream_t);
template void launch_halo_pack_joint<double>(const DW<double>&, const int32_t*, uint32_t, Vec4<double>*, hipStream_t);
template void launch_halo_unpack_joint<float>(const DW<float>&, const int32_t*, uint32_t, const Vec4<float>*, hipStream_t);
template void launch_halo_unpack_joint<double>(const DW<double>&, const int32_t*, uint32_t, const Vec4<double>*, hipStream_t);
template <class T> void launch_halo_pack(const DW<T>& w, const int32_t* bodies, uint32_t n, Vec4<T>* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_halo_pack<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, out);
}
template <class T> void launch_halo_unpack(const DW<T>& w, const int32_t* bodies, uint32_t n, const Vec4<T>* in, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_halo_unpack<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, in);
}
template void launch_halo_pack<float>(const DW<float>&, const int32_t*, uint32_t, Vec4<float>*, hipStream_t);
template void launch_halo_pack<double>(const DW<double>&, const int32_t*, uint32_t, Vec4<double>*, hipStream_t);
template void launch_halo_unpack<float>(const DW<float>&, const int32_t*, uint32_t, const Vec4<float>*, hipStream_t);
template void launch_halo_unpack<double>(const DW<double>&, const int32_t*, uint32_t, const Vec4<double>*, hipStream_t);

// ---- sharded closed loop (avn_dshard_enable): FOREIGN flags, and the bodies' components for the per-step all-gather --------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_dsh_set_foreign(DW<T> w, const int32_t* __restrict__ owner, uint32_t rank) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= w.n_bodies) return;
    const uint32_t m = w.bmeta[b];
    const bool foreign = owner && owner[b] >= 0 && owner[b] != (int)rank;
    w.bmeta[b] = meta_with_flags(m, foreign ? (meta_flags(m) | AVN_BODY_FOREIGN) : (meta_flags(m) & ~(uint32_t)AVN_BODY_FOREIGN));
}
// Position / Rotation / LinearVelocity / AngularVelocity records of the listed bodies, four records per body (the w lanes are constants of the upload: identical on every rank)
template <class T>
__global__ __launch_bounds__(256) void k_dsh_pack(DW<T> w, const uint32_t* __restrict__ bodies, uint32_t n, Vec4<T>* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = bodies[i];
    out[4 * (size_t)i] = w.pos[b]; out[4 * (size_t)i + 1] = w.rot[b]; out[4 * (size_t)i + 2] = w.lvel[b]; out[4 * (size_t)i + 3] = w.avel[b];
}
template <class T>
__global__ __launch_bounds__(256) void k_dsh_unpack(DW<T> w, const uint32_t* __restrict__ bodies, uint32_t n, const Vec4<T>* __restrict__ in) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = bodies[i];
    w.pos[b] = in[4 * (size_t)i]; w.rot[b] = in[4 * (size_t)i + 1]; w.lvel[b] = in[4 * (size_t)i + 2]; w.avel[b] = in[4 * (size_t)i + 3];
}
template <class T> void launch_dsh_set_foreign(const DW<T>& w, const int32_t* owner, uint32_t rank, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_dsh_set_foreign<T>, dim3((w.n_bodies + 255) / 256), dim3(256), 0, s, w, owner, rank);
}
template <class T> void launch_dsh_pack(const DW<T>& w, const uint32_t* bodies, uint32_t n, void* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_dsh_pack<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, (Vec4<T>*)out);
}
template <class T> void launch_dsh_unpack(const DW<T>& w, const uint32_t* bodies, uint32_t n, const void* in, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_dsh_unpack<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, (const Vec4<T>*)in);
}

#define INST(T)                                                                                     \
    template void launch_pack_bodies<T>(const DW<T>&, const BodyStage<T>&, hipStream_t);            \
    template void launch_pack_manifolds<T>(const DW<T>&, const ManifoldStage<T>&, hipStream_t);     \
    template void launch_pack_joints<T>(const DW<T>&, const JointStage<T>&, hipStream_t);           \
    template void launch_pack_colliders<T>(const BP<T>&, const ColliderStage<T>&, hipStream_t);     \
    template void launch_unpack_bodies<T>(const DW<T>&, T*, T*, T*, T*, hipStream_t);               \
    template void launch_pack_local_accelerations<T>(Vec4<T>*, Vec4<T>*, const T*, const T*, uint32_t, hipStream_t); \
    template void launch_unpack_solver_bodies<T>(const DW<T>&, const SolverBodiesStage<T>&, hipStream_t); \
    template void launch_unpack_impulses<T>(const DW<T>&, T*, T*, T*, hipStream_t);                 \
    template void launch_unpack_constraints<T>(const DW<T>&, const ConstraintsStage<T>&, hipStream_t); \
    template void launch_unpack_joints<T>(const DW<T>&, T*, T*, T*, T*, T*, T*, T*, hipStream_t);           \
    template void launch_unpack_aabbs<T>(const BP<T>&, T*, T*, uint32_t*, hipStream_t);           \
    template void launch_dsh_set_foreign<T>(const DW<T>&, const int32_t*, uint32_t, hipStream_t);  \
    template void launch_dsh_pack<T>(const DW<T>&, const uint32_t*, uint32_t, void*, hipStream_t); \
    template void launch_dsh_unpack<T>(const DW<T>&, const uint32_t*, uint32_t, const void*, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
