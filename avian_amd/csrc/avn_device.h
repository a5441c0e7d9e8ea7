// avn_device.h — the batched Structure-of-Arrays world resident in HBM, and the launch interface of
// the hand-written gfx950 kernels.  Every record is a 16-byte (f32) / 32-byte (f64) Vec4 so that
//  - streaming kernels (one thread per body / manifold / collider) issue one fully coalesced
//    dwordx4 load per lane per record (16 B/lane x 64 lanes = 1 KiB per wave instruction), and
//  - gather kernels (contacts, joints) fetch a body's state in 6 aligned vector loads that are served
//    by L2 / Infinity Cache (100k bodies x 96 B = 9.6 MB).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/avian_mi355x.h"
#include "avn_math.h"

namespace avn {

// Environment switches -- A/B runs of older code paths, debugging aids, test hooks that force rarely taken paths -- exist only in the `make measure`
// build (-DAVN_MEASURE, csrc/measure/libavian_mi355x.so: what tools/ and the switch tests load through AVN_LIB_PATH).  The release library reads NO
// environment variable: a stray AVN_* in a user's environment cannot change which code runs.
#ifdef AVN_MEASURE
inline const char* avn_env(const char* name) { return std::getenv(name); }
#else
inline const char* avn_env(const char*) { return nullptr; }
#endif


// body meta word
//   bits 0-1  rb_type (AVN_RB_*)      bits 8-13 locked axes     bits 16-23 body_flags   bits 24-31 dominance (i8)
AVN_HD uint32_t meta_rb_type(uint32_t m) { return m & 3u; }
AVN_HD uint32_t meta_locked(uint32_t m) { return (m >> 8) & 0x3Fu; }
AVN_HD uint32_t meta_flags(uint32_t m) { return (m >> 16) & 0xFFu; }
AVN_HD int meta_dominance(uint32_t m) { return (int)(int8_t)(m >> 24); }
AVN_HD uint32_t meta_with_flags(uint32_t m, uint32_t flags) { return (m & ~(0xFFu << 16)) | ((flags & 0xFFu) << 16); }
AVN_HD bool meta_active(uint32_t m) { return (meta_flags(m) & (AVN_BODY_SLEEPING | AVN_BODY_DISABLED)) == 0; }
// internal body flag (bit 7 of the flags byte; the ABI's AVN_BODY_* use bits 0-3): the body is simulated by ANOTHER rank of a sharded closed loop (avn_dshard_enable).  Broad phase,
// narrow phase and the ContactGraph / ConstraintGraph bookkeeping treat it like any body -- they are replicated on every rank --; it owns no SolverBody here, its components arrive
// from its owner after every step.
#define AVN_BODY_FOREIGN 0x80u
AVN_HD bool meta_has_solver_body(uint32_t m) { return meta_rb_type(m) != AVN_RB_STATIC && meta_active(m) && !(meta_flags(m) & AVN_BODY_FOREIGN); }

// SolverBodyFlags word kept per body: reference bits 0-7, plus
#define AVN_SBF_NO_SOLVER_BODY 0x80000000u

// constraint meta word (packed in the w lane of c_h1)
//   bits 0-2 point count   bit 4 inertia1 is DUMMY (body1 dominant)   bit 5 inertia2 is DUMMY
//   bit 6 non-dynamic softness   bit 7 has tangent part   bit 8 body1 has no SolverBody   bit 9 body2 has none
#define AVN_CM_DOM1 0x10u
#define AVN_CM_DOM2 0x20u
#define AVN_CM_SOFT_ND 0x40u
#define AVN_CM_TANGENT 0x80u
#define AVN_CM_NOBODY1 0x100u
#define AVN_CM_NOBODY2 0x200u

template <class T> struct SoftCoef { T bias, mass_scale, impulse_scale; };

template <class T> struct StepParams {
    T dt_f64cast, h_f64cast;  // Duration::as_secs_f64() as Scalar
    T dt_adj, h_adj;          // delta_seconds_adjusted()
    T gravity[3];
    T max_overlap_solve_speed;  // SolverConfig::max_overlap_solve_speed * length_unit
    T warm_start_coefficient;
    T restitution_threshold;    // * length_unit
    T contact_tolerance;        // * length_unit
    T default_speculative_margin;  // * length_unit (Limits::max = unbounded)
    T substeps_as_scalar;
    T length_unit;              // PhysicsLengthUnit
    SoftCoef<T> soft_dynamic, soft_non_dynamic;
    uint32_t restitution_iterations;
    uint32_t match_contacts;
    uint32_t np_debug;   // measurement aid (AVN_NP_DEBUG): 1 = the narrow phase stops after its input loads, 2 = after the SAT; 3 / 4 / 5 = the heavy kernel stops after its loads / after the clipping / after the pruning (results are wrong: timing only; AVN_NP_DEBUG_STEP = the one closed-loop step it applies to)
};

// Two Vec4 records of a body that are always touched together (linear|angular velocity, delta position|rotation, the
// two halves of SolverBodyInertia) share one 32-byte (f32) / 64-byte (f64) slot: a scatter of the pair dirties one
// sector instead of two partial ones, and a gather of a body touches 3 cache lines instead of 6.  Indexing stays
// `field[body]`.
template <class V> struct Pair2 {
    V* p;
    __host__ __device__ __forceinline__ V& operator[](size_t i) const { return p[2 * i]; }
};

template <class T> struct DW {
    using V = Vec4<T>;
    // ---- rigid-body components (persist across steps) ----
    uint32_t n_bodies;
    V* pos;        // (position.xyz, inv_mass)
    V* rot;        // rotation xyzw
    V* lvel;       // (LinearVelocity.xyz, gravity_scale)
    V* avel;       // (AngularVelocity.xyz, linear_damping)
    V* com;        // (center_of_mass.xyz, angular_damping)
    V* iloc_a;     // local inverse inertia (m00, m01, m02, m11)
    V* iloc_b;     // (m12, m22, max_linear_speed, max_angular_speed)
    V* acc_l;      // (accumulated linear acceleration.xyz, 0)
    V* acc_a;      // (accumulated angular acceleration.xyz, 0)
    // AccumulatedLocalAcceleration (forces/mod.rs:661-673; avn_local_accelerations_upload): nullptr = no body has one
    const V* lacc_l;   // (local linear acceleration.xyz, 0)
    const V* lacc_a;   // (local angular acceleration.xyz, 0)
    uint32_t* bmeta;
    // ---- solver bodies (SolverBody / SolverBodyInertia / VelocityIntegrationData) ----
    Pair2<V> sb_lin;   // (linear_velocity.xyz, 0)      \ one 2-record slot per body
    Pair2<V> sb_ang;   // (angular_velocity.xyz, 0)     /
    Pair2<V> sb_dp;    // (delta_position.xyz, 0)       \ one slot
    Pair2<V> sb_dq;    // delta_rotation xyzw           /
    Pair2<V> si_a;     // (inv_mass, m00, m01, m02)   world-space effective inverse inertia   \ one slot
    Pair2<V> si_b;     // (m11, m12, m22, bits(InertiaFlags | dominance << 16))               /
    V* vid_l;      // (linear_increment.xyz, linear_damping_rhs)
    V* vid_a;      // (angular_increment.xyz, angular_damping_rhs)
    V* pre_dp;     // PreSolveDeltaPosition
    V* pre_dq;     // PreSolveDeltaRotation
    uint32_t* sb_flags;
    // Island-level concurrency inside the substep loop: bodies of islands that hold joints but NO contact manifold ("side" islands: hanging
    // chains, mechanisms clear of everything) run their whole substep loop on a second stream next to the contact passes of the other
    // islands -- islands exchange nothing inside the solver (islands/mod.rs:1-10), so every body sees exactly its own operation sequence.
    // side_group[body] = 1 for bodies of side islands; body_group selects what a per-body kernel of the substep loop processes:
    // 0 = every body (no split), 1 = the main group only, 2 = the side group only.
    const uint8_t* side_group;
    uint32_t body_group;
    // ---- contact manifolds (the ContactGraph side; colour-major) ----
    uint32_t n_manifolds, m_stride;   // m_stride: element stride between point slots ([p][m] layout)
    int2* m_bodies;
    V* m_n;        // (normal.xyz, friction)            == constraint header H0
    V* m_tv;       // (tangent_velocity.xyz, restitution) == constraint header H2
    uint32_t* m_meta;   // point_count | manifold_flags << 8
    V* mp_a1;      // [p][m] (anchor1.xyz, penetration)
    V* mp_a2;      // [p][m] (anchor2.xyz, normal_speed)
    V* mp_w;       // [p][m] (warm_start_normal, warm_start_tangent.x, .y, normal_impulse)
    // ---- contact constraints (rebuilt every step) ----
    V* c_h1;       // (tangent1.xyz, bits(constraint meta))
    V* c_pa;       // [p][m] (anchor1.xyz, initial_separation)
    V* c_pb;       // [p][m] (anchor2.xyz, normal effective_mass)
    V* c_pc;       // [p][m] (tangent k0, k1, k2, normal_speed)
    V* c_pd;       // [p][m] (normal impulse, total_impulse, tangent impulse.x, .y)
    int16_t* c_reldom;  // inspection only
    uint32_t* color_offsets;  // [25] device copy
    uint32_t* constraint_count;  // [1]
    // body -> incident (manifold, side) entries for the body-centric warm start, in SOLVE order:
    //  - colours 0..22: a body is in at most ONE manifold per colour, so the incidence is a slot table inc_slot[colour][body]
    //    (colour-major planes of inc_stride entries; EMPTY = ~0), built on the device from the manifold arrays;
    //  - the overflow colour (solved first, list order): a small CSR over bodies, built by the host.
    //  entry = manifold | side << 31  (side 0: the body is the manifold's body1, 1: body2)
    const uint32_t* inc_off;   // [n_bodies + 1]  overflow entries only
    const uint32_t* inc_ent;
    uint32_t* inc_slot;        // [AVN_COLOR_OVERFLOW_INDEX][inc_stride]
    uint32_t inc_stride;
    // ---- XPBD joints (all five types; the reference's per-type components + solver data) ----
    uint32_t n_joints;
    int2* j_bodies;
    V* j_a1;       // (local_anchor1.xyz, limit_min)
    V* j_a2;       // (local_anchor2.xyz, limit_max)
    V* j_par;      // (compliance0, damping_linear, damping_angular, bits(1 has damping | type << 8 | limit_flags << 16))
    V* j_b1;       // local_basis1 quaternion
    V* j_b2;       // local_basis2 quaternion
    V* j_ax;       // (hinge | twist | slider axis.xyz, compliance1)
    V* j_l2;       // (limit2_min, limit2_max, compliance2, 0)
    V* j_r1;       // (world_r1.xyz, 0)
    V* j_r2;       // (world_r2.xyz, 0)
    V* j_cd;       // (center_difference.xyz, 0)
    V* j_lag;      // (total position lagrange.xyz, 0)
    V* j_s0;       // fixed/prismatic: rotation_difference quat | revolute: a1 | spherical: swing_axis1
    V* j_s1;       // prismatic: free_axis1 | revolute: a2 | spherical: swing_axis2
    V* j_s2;       // revolute: b1 | spherical: twist_axis1
    V* j_s3;       // revolute: b2 | spherical: twist_axis2
    V* j_rl0;      // total align | swing | angle lagrange
    V* j_rl1;      // total limit | twist lagrange
    V* j_force;    // (force.xyz, 0)
    V* j_torque;   // (torque.xyz, 0)
};
#ifdef __HIPCC__
template <class T> __device__ __forceinline__ bool body_in_group(const DW<T>& w, uint32_t body) {
    return w.body_group == 0u || (uint32_t)w.side_group[body] == w.body_group - 1u;
}
#endif


// Block index remap so that each XCD (block b runs on XCD b % 8) walks one contiguous eighth of the
// work: neighbouring work items share bodies / sweep ranges, and each XCD has a private 4 MiB L2.
// Placement only changes speed, never results.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t nb) {
    uint32_t per = (nb + 7u) >> 3;
    return (b & 7u) * per + (b >> 3);
}

}  // namespace avn
