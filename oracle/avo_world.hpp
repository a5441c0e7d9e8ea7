// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
//
// avo_world.hpp: CPU restatement of Avian's 3D hot path, in the reference's own operation order
// (SURVEY.md §3.2, Appendix A).  Serial by default; AVO_THREADS=n runs the loops the reference itself
// parallelises (avo_parallel.hpp) on n threads with bit-identical results.  Every function cites the reference file:line it
// follows (paths relative to /root/reference/src).
//
// PARITY PINNING: the reference cannot be built here (no cargo/rustc; bevy/glam/parry not
// vendored), so this restatement is pinned only by the reference's own known-answer tests that
// touch the path (integrator/mod.rs:561-629 `semi_implicit_euler`, tests/mod.rs:93-142
// `body_with_velocity_moves`, forces/tests.rs:53-96,249-292,552-601, solver_body/plugin.rs:318-353 `add_remove_solver_bodies`,
// physics_material.rs:398-446 `coefficient_combine_works`) — transcribed in
// tests/golden/reference_kats.json and checked by tests/test_oracle_golden.py.  For contacts, colouring,
// broad phase pair lists and XPBD joints the reference holds NO golden vectors: "parity unpinned" there;
// the restatement is reviewed line by line against the cited source instead.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <map>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/avian_mi355x.h"
#include "avo_parallel.hpp"
#include "avo_islands.hpp"
#include "avo_math.hpp"
#include "avo_narrow.hpp"

namespace avo {

// ---- softness (solver/softness_parameters/mod.rs:36-41,64-79) ----------------------------------
template <class S> struct SoftnessCoefficients { S bias, mass_scale, impulse_scale; };
template <class S> inline SoftnessCoefficients<S> softness_coefficients(S damping_ratio, S frequency_hz, S delta_secs) {
    const S TAU = S(6.283185307179586476925286766559);
    S double_damping_ratio = S(2) * damping_ratio;
    S angular_frequency = TAU * frequency_hz;
    S a1 = double_damping_ratio + angular_frequency * delta_secs;
    S a2 = angular_frequency * delta_secs * a1;
    S a3 = S(1) / (S(1) + a2);
    return {angular_frequency / a1, a2 * a3, a3};
}

// ---- solver_body/mod.rs:59-91 ---------------------------------------------------------------------
template <class S> struct SolverBody {
    V3<S> linear_velocity{0, 0, 0}, angular_velocity{0, 0, 0}, delta_position{0, 0, 0};
    Q4<S> delta_rotation{0, 0, 0, 1};
    uint32_t flags = 0;
    V3<S> velocity_at_point(V3<S> p) const { return linear_velocity + cross(angular_velocity, p); }  // :107-116
    bool is_kinematic() const { return flags & AVN_SB_KINEMATIC; }
    bool is_gyroscopic() const { return flags & AVN_SB_GYROSCOPIC; }
};
// solver_body/mod.rs:218-276
template <class S> struct SolverBodyInertia {
    S inv_mass = 0;
    Sym3<S> inv_inertia{0, 0, 0, 0, 0, 0};
    int16_t dominance = 128;  // DUMMY: i8::MAX + 1
    uint16_t flags = 0xC0;    // InertiaFlags::STATIC
    // :437-451
    V3<S> effective_inv_mass() const {
        V3<S> m{inv_mass, inv_mass, inv_mass};
        if (flags & 0b100000) m.x = 0;
        if (flags & 0b010000) m.y = 0;
        if (flags & 0b001000) m.z = 0;
        return m;
    }
};
template <class S> inline void lock_rotation_axes(Sym3<S>& t, uint32_t locked) {  // :400-414 / :483-499
    if (locked & 0b100) { t.m00 = 0; t.m01 = 0; t.m02 = 0; }
    if (locked & 0b010) { t.m01 = 0; t.m11 = 0; t.m12 = 0; }
    if (locked & 0b001) { t.m02 = 0; t.m12 = 0; t.m22 = 0; }
}
// integrator/mod.rs:216-233
template <class S> struct VelocityIntegrationData {
    V3<S> linear_increment{0, 0, 0}, angular_increment{0, 0, 0};
    S linear_damping_rhs = 0, angular_damping_rhs = 0;  // #[derive(Default)]
};

template <class S> struct Body {
    V3<S> position, linear_velocity, angular_velocity, center_of_mass;
    Q4<S> rotation;
    S inv_mass;
    Sym3<S> inv_inertia_local;
    S linear_damping, angular_damping, gravity_scale, max_linear_speed, max_angular_speed;
    uint8_t rb_type, locked_axes, body_flags;
    int8_t dominance;
    // solver components
    bool has_solver_body;  // SolverBody exists for awake, enabled dynamic/kinematic bodies (solver_body/plugin.rs:38-96)
    SolverBody<S> sb;
    SolverBodyInertia<S> si;
    VelocityIntegrationData<S> vid;
    V3<S> pre_solve_delta_position;
    Q4<S> pre_solve_delta_rotation;
    bool active() const { return !(body_flags & (AVN_BODY_SLEEPING | AVN_BODY_DISABLED)); }  // RigidBodyActiveFilter
};

// ---- contacts -------------------------------------------------------------------------------------
template <class S> struct ContactPoint {  // contact_types/mod.rs:603-660
    V3<S> anchor1, anchor2;
    S penetration, normal_speed, warm_start_normal_impulse, normal_impulse;
    V2<S> warm_start_tangent_impulse;
};
template <class S> struct ContactManifold {  // contact_types/mod.rs:342-378 (+ pair body handles)
    int32_t body1, body2;
    V3<S> normal, tangent_velocity;
    S friction, restitution;
    uint8_t point_count, flags;
    ContactPoint<S> points[AVN_MAX_MANIFOLD_POINTS];
};
template <class S> struct ContactNormalPart { S impulse, total_impulse, effective_mass; SoftnessCoefficients<S> softness; };
template <class S> struct ContactTangentPart { V2<S> impulse; S effective_inverse_mass[3]; };
template <class S> struct ContactConstraintPoint {  // contact/mod.rs:32-54
    ContactNormalPart<S> normal_part;
    bool has_tangent;
    ContactTangentPart<S> tangent_part;
    V3<S> anchor1, anchor2;
    S normal_speed, initial_separation;
};
template <class S> struct ContactConstraint {  // contact/mod.rs:64-106
    int32_t body1, body2;
    int16_t relative_dominance;
    S friction, restitution;
    V3<S> tangent_velocity, normal, tangent1;
    int point_count;
    ContactConstraintPoint<S> points[AVN_MAX_MANIFOLD_POINTS];
    uint32_t manifold;  // index of the source manifold (contact_id, manifold_index in the reference)
    bool softness_non_dynamic;
};

// contact/normal_part.rs:39-112
template <class S>
inline ContactNormalPart<S> normal_part_generate(V3<S> w_sum, const Sym3<S>& i1, const Sym3<S>& i2, V3<S> r1, V3<S> r2,
                                                 V3<S> normal, bool warm, S warm_impulse, SoftnessCoefficients<S> soft) {
    V3<S> r1_cross_n = cross(r1, normal);
    V3<S> r2_cross_n = cross(r2, normal);
    S k_linear = dot(normal, cmul(w_sum, normal));
    S k = k_linear + dot(r1_cross_n, smul(i1, r1_cross_n)) + dot(r2_cross_n, smul(i2, r2_cross_n));
    return {warm ? warm_impulse : S(0), S(0), recip_or_zero(k), soft};
}
// contact/normal_part.rs:116-166
template <class S>
inline S normal_part_solve_impulse(ContactNormalPart<S>& p, S separation, V3<S> relative_velocity, V3<S> normal, bool use_bias,
                                   S max_overlap_solve_speed, S delta_secs) {
    S normal_speed = dot(relative_velocity, normal);
    S impulse;
    if (separation > S(0)) {
        impulse = -p.effective_mass * (normal_speed + separation / delta_secs);
    } else if (use_bias) {
        S bias = smax(p.softness.bias * separation, -max_overlap_solve_speed);
        S scaled_mass = p.softness.mass_scale * p.effective_mass;
        S scaled_impulse = p.softness.impulse_scale * p.impulse;
        impulse = -scaled_mass * (normal_speed + bias) - scaled_impulse;
    } else {
        impulse = -p.effective_mass * normal_speed;
    }
    S new_impulse = smax(p.impulse + impulse, S(0));
    impulse = new_impulse - p.impulse;
    p.impulse = new_impulse;
    p.total_impulse += new_impulse;  // sic: adds the new ACCUMULATED value (:162)
    return impulse;
}
// contact/tangent_part.rs:35-151 (3D)
template <class S>
inline ContactTangentPart<S> tangent_part_generate(V3<S> w_sum, const Sym3<S>& i1, const Sym3<S>& i2, V3<S> r1, V3<S> r2,
                                                   V3<S> t0, V3<S> t1, bool warm, V2<S> warm_impulse) {
    ContactTangentPart<S> part;
    part.impulse = warm ? warm_impulse : V2<S>{0, 0};
    V3<S> rt11 = cross(r1, t0), rt12 = cross(r2, t0), rt21 = cross(r1, t1), rt22 = cross(r2, t1);
    V3<S> i1_rt11 = smul(i1, rt11), i2_rt12 = smul(i2, rt12), i1_rt21 = smul(i1, rt21), i2_rt22 = smul(i2, rt22);
    S k_linear1 = dot(t0, cmul(w_sum, t0));
    S k_linear2 = dot(t1, cmul(w_sum, t1));
    S k1 = k_linear1 + dot(rt11, i1_rt11) + dot(rt12, i2_rt12);
    S k2 = k_linear2 + dot(rt21, i1_rt21) + dot(rt22, i2_rt22);
    part.effective_inverse_mass[0] = k1;
    part.effective_inverse_mass[1] = k2;
    part.effective_inverse_mass[2] = S(2) * (dot(rt11, i1_rt21) + dot(rt12, i2_rt22));
    return part;
}
// contact/tangent_part.rs:155-244 (3D)
template <class S>
inline V3<S> tangent_part_solve_impulse(ContactTangentPart<S>& p, V3<S> t0, V3<S> t1, V3<S> relative_velocity,
                                        V3<S> surface_velocity, S friction, S normal_impulse) {
    S impulse_limit = friction * normal_impulse;
    V3<S> rv = relative_velocity + surface_velocity;
    S tangent_speed1 = dot(rv, t0);
    S tangent_speed2 = dot(rv, t1);
    S t11 = tangent_speed1 * tangent_speed1;  // powi(2)
    S t22 = tangent_speed2 * tangent_speed2;
    S t12 = tangent_speed1 * tangent_speed2;
    S inv = t11 * p.effective_inverse_mass[0] + t22 * p.effective_inverse_mass[1] + t12 * p.effective_inverse_mass[2];
    S effective_mass = (t11 + t22) * (S(1) / inv);
    if (!std::isfinite(effective_mass)) return vzero<S>();
    V2<S> delta_impulse{effective_mass * tangent_speed1, effective_mass * tangent_speed2};
    V2<S> new_impulse = clamp_length_max(V2<S>{p.impulse.x - delta_impulse.x, p.impulse.y - delta_impulse.y}, impulse_limit);
    V2<S> impulse{new_impulse.x - p.impulse.x, new_impulse.y - p.impulse.y};
    p.impulse = new_impulse;
    return impulse.x * t0 + impulse.y * t1;
}
// contact/mod.rs:427-449
template <class S> inline void compute_tangent_directions(V3<S> normal, V3<S> velocity1, V3<S> velocity2, V3<S>& t0, V3<S>& t1) {
    V3<S> force_direction = -normal;
    V3<S> relative_velocity = velocity1 - velocity2;
    V3<S> tangent_velocity = relative_velocity - force_direction * dot(force_direction, relative_velocity);
    V3<S> tangent;
    if (!try_normalize(tangent_velocity, tangent)) tangent = any_orthonormal_vector(force_direction);
    t0 = tangent;
    t1 = cross(force_direction, tangent);
}

// ---- XPBD -------------------------------------------------------------------------------------------
// One struct for the five XPBD joint types (dynamics/joints/{fixed,revolute,spherical,prismatic,distance}.rs) and their
// solver data (solver/xpbd/joints/*.rs).
template <class S> struct Joint {
    uint8_t type = AVN_JOINT_DISTANCE;
    int32_t body1, body2;
    V3<S> local_anchor1, local_anchor2;
    Q4<S> local_basis1{0, 0, 0, 1}, local_basis2{0, 0, 0, 1};
    V3<S> axis{0, 0, 0};               // hinge_axis / twist_axis / slider_axis
    S limit_min = 0, limit_max = 0;    // DistanceLimit or AngleLimit (revolute angle_limit, spherical swing_limit)
    S limit2_min = 0, limit2_max = 0;  // spherical twist_limit
    uint8_t limit_flags = 0;           // AVN_JOINT_HAS_LIMIT*
    S compliance = 0, compliance1 = 0, compliance2 = 0;  // see avn_joints.compliance
    bool has_damping;
    S damping_linear, damping_angular;
    bool collision_disabled;
    // PointConstraintShared / DistanceJointSolverData / PrismaticJointSolverData
    V3<S> world_r1{0, 0, 0}, world_r2{0, 0, 0}, center_difference{0, 0, 0}, total_lagrange{0, 0, 0};
    // FixedAngleConstraintShared (fixed, prismatic)
    Q4<S> rotation_difference{0, 0, 0, 1};
    // RevoluteJointSolverData a1,a2,b1,b2 / SphericalJointSolverData swing_axis1,2 twist_axis1,2 / prismatic free_axis1 (in v0)
    V3<S> v0{0, 0, 0}, v1{0, 0, 0}, v2{0, 0, 0}, v3{0, 0, 0};
    // total_align|swing|angle lagrange, total_limit|twist lagrange
    V3<S> total_rot0{0, 0, 0}, total_rot1{0, 0, 0};
    V3<S> force{0, 0, 0}, torque{0, 0, 0};
};
// xpbd/mod.rs:393-413
template <class S> inline S compute_lagrange_update(S lagrange, S c, S w1, S w2, S compliance, S dt) {
    S w_sum = S(0) + w1 + w2;  // iter().copied().sum() starts from 0.0
    if (w_sum <= std::numeric_limits<S>::epsilon()) return S(0);
    S tilde_compliance = compliance / (dt * dt);  // dt.powi(2)
    return (-c - tilde_compliance * lagrange) / (w_sum + tilde_compliance);
}

// ---- broad phase --------------------------------------------------------------------------------------
template <class S> struct ColliderAabb { V3<S> min, max; };
template <class S> struct Collider {
    uint32_t entity;
    int32_t body;
    uint8_t shape, cflags;
    V3<S> half_extents;
    uint32_t memberships, filters;
    S collision_margin, speculative_margin;  // speculative < 0 = absent
    ColliderAabb<S> aabb;
    // a collider on a CHILD entity of its rigid body (avn_collider_transforms_upload): ColliderTransform::translation / rotation, scale applied by the host
    bool child = false;
    V3<S> local_translation{0, 0, 0};
    Q4<S> local_rotation{0, 0, 0, 1};
};
struct AabbInterval {  // broad_phase.rs:177-185 (aabb/layers looked up through the collider slot)
    uint32_t collider;  // slot in `colliders`
    uint8_t flags;
};

inline uint64_t pair_key(uint32_t a, uint32_t b) {  // data_structures/pair_key.rs:14-21
    return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
}

// ---- the world ------------------------------------------------------------------------------------------
struct WorldBase {
    std::string error;
    virtual ~WorldBase() {}
    virtual avn_status config_set(const avn_config*) = 0;
    virtual avn_status bodies_upload(const avn_bodies*) = 0;
    virtual avn_status bodies_download(const avn_bodies_out*) = 0;
    virtual avn_status solver_bodies_download(const avn_solver_bodies_out*) = 0;
    virtual avn_status local_accelerations_upload(uint32_t, const void*, const void*) = 0;
    virtual avn_status manifolds_upload(const avn_manifolds*) = 0;
    virtual avn_status impulses_download(const avn_impulses_out*) = 0;
    virtual avn_status constraints_download(const avn_constraints_out*) = 0;
    virtual avn_status distance_joints_upload(const avn_distance_joints*) = 0;
    virtual avn_status joints_upload(const avn_joints*) = 0;
    virtual avn_status joints_download(const avn_joints_out*) = 0;
    virtual avn_status colliders_upload(const avn_colliders*) = 0;
    virtual avn_status collider_transforms_upload(const avn_collider_transforms*) = 0;
    virtual avn_status existing_pairs_upload(const uint64_t*, size_t) = 0;
    virtual avn_status pairs_get(const avn_pair**, size_t*) = 0;
    virtual avn_status aabbs_download(void*, void*, uint32_t*, size_t*) = 0;
    virtual avn_status run_system(avn_system) = 0;
    virtual avn_status step() = 0;
    virtual avn_status timers(avn_timers*) = 0;
    virtual avn_status diagnostics(avn_diagnostics*) = 0;
    virtual avn_status islands_get(uint32_t*, uint32_t*) = 0;
    virtual avn_status sleep_update(const avn_sleep_params*, avn_sleep_stats*) = 0;
    virtual avn_status sleep_get(const avn_sleep_out*) = 0;
    virtual avn_status sleep_reset(const uint32_t*, size_t) = 0;
    virtual avn_status host_shapes_set(avn_host_aabb_fn, avn_host_manifolds_fn, void*) = 0;
    virtual avn_status host_shape_stats_get(avn_host_shape_stats*) = 0;
    virtual avn_status collision_hooks_set(avn_filter_pairs_fn, avn_modify_contacts_fn, void*) = 0;
    virtual avn_status collision_hook_stats_get(avn_collision_hook_stats*) = 0;
    virtual avn_status halo_plan_upload(const avn_halo_plan*) = 0;
    virtual avn_status halo_overflow_levels_upload(uint32_t, const uint32_t*, size_t) = 0;
    virtual avn_status halo_joint_slot_set(uint32_t, uint32_t) = 0;
    virtual avn_status run_color_pass(avn_system, uint32_t) = 0;
    virtual avn_status halo_pack(uint32_t, uint32_t, void*, size_t*) = 0;
    virtual avn_status halo_unpack(uint32_t, uint32_t, const void*, size_t) = 0;
    virtual avn_status profile_system(avn_system, uint32_t, double*, uint32_t*) = 0;
    virtual avn_status dynamic_bounds(double*, double*) = 0;
    virtual avn_status contact_manifolds(const avn_shape_pairs*, const avn_query_manifolds_out*) = 0;
    virtual avn_status collider_materials_upload(const avn_collider_materials*) = 0;
    virtual avn_status contact_pairs_add(const avn_contact_pairs*) = 0;
    virtual avn_status contact_pairs_remove(const uint32_t*, size_t) = 0;
    virtual avn_status active_pairs_set(const uint32_t*, size_t) = 0;
    virtual avn_status contact_changes_get(const avn_contact_change**, size_t*) = 0;
    virtual avn_status manifold_handles_upload(const uint32_t*, const uint32_t*) = 0;
    virtual avn_status contacts_download(const uint32_t*, size_t, const avn_contacts_out*) = 0;
    virtual avn_status contacts_upload(const uint32_t*, size_t, const avn_contacts_in*) = 0;
    virtual avn_status pipeline_enable(int) = 0;
    virtual avn_status pipeline_stats_get(avn_pipeline_stats*) = 0;
    virtual avn_status pipeline_handles_get(uint32_t*, const uint32_t**, size_t*) = 0;
    virtual avn_status pipeline_new_pair_ids_get(const uint32_t**, size_t*) = 0;
    virtual avn_status sleeping_enable(const avn_sleep_params*) = 0;
    virtual avn_status sleeping_stats_get(avn_sleeping_stats*) = 0;
    virtual avn_status sleeping_state_get(const avn_sleeping_out*) = 0;
    virtual avn_status wake_bodies(const uint32_t*, size_t) = 0;
    virtual avn_status despawn(const avn_despawn_list*) = 0;
    virtual avn_status dshard_enable(const avn_dshard_config*) = 0;
    virtual avn_status dshard_bodies_pack(void*, size_t, size_t*) = 0;
    virtual avn_status dshard_bodies_unpack(uint32_t, const void*, size_t) = 0;
    virtual avn_status dshard_stats_get(avn_dshard_stats*) = 0;
};

struct ConstraintGraph;  // defined below (solver/constraint_graph.rs restatement)
struct PipelineState;    // standalone closed-loop bookkeeping (defined after ConstraintGraph)
PipelineState* pipeline_new();
void pipeline_delete(PipelineState*);

template <class S> struct World : WorldBase {
    avn_config cfg;
    Pool pool{Pool::from_env()};   // CPU-baseline threads (AVO_THREADS; 1 = the plain serial restatement)
    // time scalars (SURVEY.md Appendix A addenda "Time scalars")
    S dt_f64cast, h_f64cast;  // Duration::as_secs_f64() as Scalar   (integrator/mod.rs:275,354; plugin.rs:333-334)
    S dt_adj, h_adj;          // delta_seconds_adjusted()            (schedule/time.rs:282-291)
    SoftnessCoefficients<S> soft_dynamic, soft_non_dynamic;

    std::vector<Body<S>> bodies;
    std::vector<V3<S>> accel_linear, accel_angular;
    std::vector<ContactManifold<S>> manifolds;
    uint32_t color_offsets[AVN_GRAPH_COLOR_COUNT + 1];
    std::vector<ContactConstraint<S>> color_constraints[AVN_GRAPH_COLOR_COUNT];  // GraphColor::contact_constraints
    std::vector<Joint<S>> joints;
    std::vector<uint32_t> joint_order;  // solve order: stable by type (xpbd/plugin.rs:77-82), then array (= spawn) order
    std::vector<Collider<S>> colliders;
    std::unordered_map<uint32_t, uint32_t> collider_slot;  // entity -> slot
    std::vector<AabbInterval> intervals;                    // AabbIntervals, kept sorted across frames
    std::unordered_set<uint64_t> pair_set;
    std::unordered_set<uint64_t> collision_disabled_bodies;
    std::vector<avn_pair> pairs;
    avn_timers last_timers;
    bool have_colliders = false;
    // ---- the ContactGraph side of the narrow phase (rows indexed by ContactId) ----
    struct CtPoint {  // ContactPoint, contact_types/mod.rs:603-660
        V3<S> anchor1, anchor2;
        S penetration, normal_speed, warm_start_normal_impulse, normal_impulse;
        V2<S> warm_start_tangent_impulse;
        uint32_t feature_id1, feature_id2;
    };
    struct CtRow {    // ContactPair (+ its single manifold: convex shapes)
        bool used = false;
        uint32_t collider1 = 0, collider2 = 0;  // Entity::index()
        uint32_t flags = 0;                      // AVN_CP_*
        int32_t manifold_count_change = 0;
        uint32_t n_manifolds = 0;                // 0 | 1
        V3<S> normal{0, 0, 0};
        V3<S> tangent_velocity{0, 0, 0};         // ContactManifold::tangent_velocity: zero unless CollisionHooks::modify_contacts set it
        S friction = 0, restitution = 0;
        int point_count = 0;
        CtPoint pts[AVN_MAX_MANIFOLD_POINTS];
    };
    std::vector<CtRow> contact_rows;
    std::vector<uint32_t> active_pairs;
    std::vector<avn_contact_change> contact_changes;
    std::vector<uint32_t> manifold_handles;  // colour-major contact ids; empty = manifolds come from manifolds_upload
    bool use_handles = false;
    struct Material { S friction, restitution; uint8_t friction_combine, restitution_combine; };
    std::vector<Material> materials;         // per collider slot

    PipelineState* pipe = nullptr;
    World() { std::memset(&last_timers, 0, sizeof last_timers); std::memset(color_offsets, 0, sizeof color_offsets); }
    ~World() override { pipeline_delete(pipe); delete slp; }
    avn_status pipeline_enable(int on) override;
    avn_status pipeline_stats_get(avn_pipeline_stats* o) override;
    avn_status pipeline_handles_get(uint32_t* off, const uint32_t** ids, size_t* n) override;
    std::vector<uint32_t> new_pair_ids;   // IdPool::alloc_id results of the last closed-loop step, in emission order (avn_pipeline_new_pair_ids_get)
    avn_status pipeline_new_pair_ids_get(const uint32_t** ids, size_t* n) override {
        if (!ids || !n) return AVN_ERR_BAD_ARG;
        if (!pipe) { error = "pipeline_new_pair_ids_get: needs avn_pipeline_enable"; return AVN_ERR_STATE; }
        *ids = new_pair_ids.data(); *n = new_pair_ids.size();
        return AVN_OK;
    }
    avn_status pipeline_step();
    avn_status pipeline_refresh_handles();
    // ---- the closed loop sharded by islands (header: avn_dshard_*): the whole front of the step on every body, a SolverBody only for the bodies this rank simulates,
    //      the solver's handle list = the replicated colour lists restricted to those bodies' manifolds (order kept), the other ranks' bodies copied in after every step ----
    bool dsh_on = false;
    uint32_t dsh_ranks = 1, dsh_rank = 0, dsh_own_manifolds = 0, dsh_global_manifolds = 0;
    std::vector<int32_t> dsh_owner;
    bool dsh_foreign(size_t b) const { return dsh_on && b < dsh_owner.size() && dsh_owner[b] >= 0 && (uint32_t)dsh_owner[b] != dsh_rank; }
    void dsh_refresh_solver_bodies() { for (size_t b = 0; b < bodies.size(); ++b) bodies[b].has_solver_body = bodies[b].rb_type != AVN_RB_STATIC && bodies[b].active() && !dsh_foreign(b); }
    avn_status dshard_enable(const avn_dshard_config* c) override {
        if (!c) { dsh_on = false; dsh_refresh_solver_bodies(); return AVN_OK; }
        if (c->struct_size != sizeof(avn_dshard_config) || !c->n_ranks || c->rank >= c->n_ranks || !c->body_owner) { error = "dshard_enable: bad argument"; return AVN_ERR_BAD_ARG; }
        if (!pipe) { error = "dshard_enable: needs the closed loop (avn_pipeline_enable)"; return AVN_ERR_STATE; }
        if (slp) { error = "dshard_enable: not combined with avn_sleeping_enable (the island manager is per world)"; return AVN_ERR_STATE; }
        for (size_t b = 0; b < bodies.size(); ++b) {
            if (c->body_owner[b] >= (int32_t)c->n_ranks) { error = "dshard_enable: body_owner names a rank that does not exist"; return AVN_ERR_BAD_ARG; }
            if (c->body_owner[b] < 0 && bodies[b].rb_type != AVN_RB_STATIC) { error = "dshard_enable: every non-static body needs an owner"; return AVN_ERR_BAD_ARG; }
        }
        dsh_owner.assign(c->body_owner, c->body_owner + bodies.size());
        dsh_ranks = c->n_ranks; dsh_rank = c->rank; dsh_on = true;
        dsh_refresh_solver_bodies();
        return AVN_OK;
    }
    avn_status dshard_bodies_pack(void* out, size_t cap, size_t* bytes) override {
        if (!dsh_on) { error = "dshard_bodies_pack: avn_dshard_enable first"; return AVN_ERR_STATE; }
        size_t n = 0;
        for (size_t b = 0; b < bodies.size(); ++b) n += dsh_owner[b] == (int32_t)dsh_rank;
        if (bytes) *bytes = n * 16 * sizeof(S);
        if (!out || cap < n * 16 * sizeof(S)) { error = "dshard_bodies_pack: the buffer is too small"; return AVN_ERR_BAD_ARG; }
        S* o = (S*)out;
        for (size_t b = 0; b < bodies.size(); ++b) {
            if (dsh_owner[b] != (int32_t)dsh_rank) continue;
            const Body<S>& B = bodies[b];
            const S rec[16] = {B.position.x, B.position.y, B.position.z, B.inv_mass, B.rotation.x, B.rotation.y, B.rotation.z, B.rotation.w,
                               B.linear_velocity.x, B.linear_velocity.y, B.linear_velocity.z, B.gravity_scale, B.angular_velocity.x, B.angular_velocity.y, B.angular_velocity.z, B.linear_damping};
            std::memcpy(o, rec, sizeof rec); o += 16;
        }
        return AVN_OK;
    }
    avn_status dshard_bodies_unpack(uint32_t from, const void* in, size_t bytes) override {
        if (!dsh_on) { error = "dshard_bodies_unpack: avn_dshard_enable first"; return AVN_ERR_STATE; }
        if (from >= dsh_ranks || from == dsh_rank) { error = "dshard_bodies_unpack: another rank of the shard"; return AVN_ERR_BAD_ARG; }
        size_t n = 0;
        for (size_t b = 0; b < bodies.size(); ++b) n += dsh_owner[b] == (int32_t)from;
        if (bytes != n * 16 * sizeof(S) || (n && !in)) { error = "dshard_bodies_unpack: the byte count is not that rank's bodies x 16 scalars"; return AVN_ERR_BAD_ARG; }
        const S* r = (const S*)in;
        for (size_t b = 0; b < bodies.size(); ++b) {
            if (dsh_owner[b] != (int32_t)from) continue;
            Body<S>& B = bodies[b];
            B.position = {r[0], r[1], r[2]}; B.rotation = {r[4], r[5], r[6], r[7]}; B.linear_velocity = {r[8], r[9], r[10]}; B.angular_velocity = {r[12], r[13], r[14]};
            r += 16;
        }
        return AVN_OK;
    }
    avn_status dshard_stats_get(avn_dshard_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        std::memset(o, 0, sizeof *o);
        if (!dsh_on) return AVN_OK;
        uint32_t own = 0;
        for (size_t b = 0; b < bodies.size(); ++b) own += dsh_owner[b] == (int32_t)dsh_rank;
        o->n_ranks = dsh_ranks; o->rank = dsh_rank; o->own_bodies = own; o->own_manifolds = dsh_own_manifolds; o->global_manifolds = dsh_global_manifolds;
        o->bytes_sent_per_step = (uint64_t)own * 16 * sizeof(S);
        return AVN_OK;
    }
    // ---- persistent islands + sleeping in the closed loop (header: avn_sleeping_enable; avo_islands.hpp) ----
    struct Sleeping {
        avn_sleep_params p;
        std::vector<float> lin, ang;          // per-body SleepThreshold (empty: the world-level value)
        std::vector<uint8_t> disabled;        // SleepingDisabled
        IslandManager isl;
        std::vector<float> timer;             // SleepTimer
        uint32_t n_awake = 0, last_slept = 0, last_woken = 0, last_popped = 0, last_pushed = 0;
        double host_ms = 0;
    };
    Sleeping* slp = nullptr;
    avn_status sleeping_enable(const avn_sleep_params* p) override;
    avn_status sleeping_stats_get(avn_sleeping_stats* o) override;
    avn_status sleeping_state_get(const avn_sleeping_out* o) override;
    avn_status wake_bodies(const uint32_t* ids, size_t n) override;
    avn_status despawn(const avn_despawn_list* d) override;
    bool despawn_needs_bodies = false, despawn_needs_colliders = false;   // avn_despawn happened: the host owes the remaining bodies / colliders
    void sleeping_apply(bool count);          // the ConstraintGraph / Sleeping-component side of the manager's last result
    void sleeping_systems();                  // split_island + the Sleeping set, after the solver

    // -- time: Duration arithmetic of run_physics_schedule / run_substep_schedule (schedule/mod.rs:240-284,
    //    solver/schedule.rs:194-200).  sub_delta = delta.div_f64(substeps) = from_secs_f64(secs/substeps)
    //    (rounds to the nearest nanosecond).
    static S as_secs_adjusted(uint64_t ns) {
        if (sizeof(S) == 8) return (S)((double)(ns / 1000000000ull) + (double)(ns % 1000000000ull) / 1e9);
        // Duration::as_secs_f32: (secs as f32) + (nanos as f32) / (NANOS_PER_SEC as f32)
        return (S)((float)(ns / 1000000000ull) + (float)(ns % 1000000000ull) / 1e9f);
    }
    static double as_secs_f64(uint64_t ns) { return (double)(ns / 1000000000ull) + (double)(ns % 1000000000ull) / 1e9; }
    avn_status config_set(const avn_config* c) override {
        if (!c || c->substeps == 0 || c->dt_ns == 0) { error = "bad config"; return AVN_ERR_BAD_ARG; }
        cfg = *c;
        if (cfg.solver_iterations == 0) cfg.solver_iterations = 1;
        uint64_t dt_ns = cfg.dt_ns;
        double sub = as_secs_f64(dt_ns) / (double)cfg.substeps;
        uint64_t h_ns = (uint64_t)std::llround(sub * 1e9);
        dt_f64cast = (S)as_secs_f64(dt_ns);
        h_f64cast = (S)as_secs_f64(h_ns);
        dt_adj = as_secs_adjusted(dt_ns);
        h_adj = as_secs_adjusted(h_ns);
        update_contact_softness();
        return AVN_OK;
    }
    // solver/plugin.rs:326-350
    void update_contact_softness() {
        S dt = dt_f64cast, h = h_f64cast;
        S max_hz = S(1) / (dt * S(2));
        S hz = (S)cfg.contact_frequency_factor * smin(max_hz, S(0.25) / h);
        soft_dynamic = softness_coefficients<S>((S)cfg.contact_damping_ratio, hz, h);
        soft_non_dynamic = softness_coefficients<S>((S)cfg.contact_damping_ratio, S(2) * hz, h);
    }

    // ---------------------------------------------------------------------------------------------
    template <class T> static T rd(const void* p, size_t i, T dflt) { return p ? ((const T*)p)[i] : dflt; }
    static V3<S> rd3(const void* p, size_t i) { if (!p) return vzero<S>(); const S* a = (const S*)p + 3 * i; return {a[0], a[1], a[2]}; }
    static void wr3(void* p, size_t i, V3<S> v) { if (!p) return; S* a = (S*)p + 3 * i; a[0] = v.x; a[1] = v.y; a[2] = v.z; }

    avn_status bodies_upload(const avn_bodies* b) override {
        if (!b || (b->count && (!b->position || !b->rotation || !b->linear_velocity || !b->angular_velocity || !b->inv_mass ||
                                !b->inv_inertia_local || !b->rb_type))) { error = "bodies_upload: null array"; return AVN_ERR_BAD_ARG; }
        size_t n = b->count;
        if (despawn_needs_bodies && n != bodies.size()) { error = "bodies_upload: after avn_despawn exactly the remaining bodies must be uploaded"; return AVN_ERR_STATE; }
        despawn_needs_bodies = false;
        if (n != bodies.size()) { local_acc_linear.clear(); local_acc_angular.clear(); }   // (header: dropped by an upload with another body count)
        bodies.resize(n);
        accel_linear.resize(n);
        accel_angular.resize(n);
        for (size_t i = 0; i < n; ++i) {
            Body<S>& o = bodies[i];
            o.position = rd3(b->position, i);
            const S* q = (const S*)b->rotation + 4 * i;
            o.rotation = {q[0], q[1], q[2], q[3]};
            o.linear_velocity = rd3(b->linear_velocity, i);
            o.angular_velocity = rd3(b->angular_velocity, i);
            o.inv_mass = ((const S*)b->inv_mass)[i];
            const S* t = (const S*)b->inv_inertia_local + 6 * i;
            o.inv_inertia_local = {t[0], t[1], t[2], t[3], t[4], t[5]};
            o.center_of_mass = rd3(b->center_of_mass, i);
            o.linear_damping = rd<S>(b->linear_damping, i, 0);
            o.angular_damping = rd<S>(b->angular_damping, i, 0);
            o.gravity_scale = rd<S>(b->gravity_scale, i, 1);
            o.max_linear_speed = rd<S>(b->max_linear_speed, i, -1);
            o.max_angular_speed = rd<S>(b->max_angular_speed, i, -1);
            o.rb_type = b->rb_type[i];
            o.locked_axes = rd<uint8_t>(b->locked_axes, i, 0);
            o.dominance = rd<int8_t>(b->dominance, i, 0);
            o.body_flags = rd<uint8_t>(b->body_flags, i, 0);
            o.has_solver_body = o.rb_type != AVN_RB_STATIC && o.active() && !dsh_foreign(i);
            o.sb = SolverBody<S>();
            o.si = SolverBodyInertia<S>();
            o.vid = VelocityIntegrationData<S>();
            o.pre_solve_delta_position = vzero<S>();
            o.pre_solve_delta_rotation = qidentity<S>();
            accel_linear[i] = rd3(b->accel_linear, i);
            accel_angular[i] = rd3(b->accel_angular, i);
            if (slp) {   // the Sleeping component is the island manager's (SleepIslands / WakeIslands), not the uploader's
                if (o.rb_type != AVN_RB_STATIC && !(o.body_flags & AVN_BODY_DISABLED) && !slp->isl.has_node((uint32_t)i)) slp->isl.body_add((uint32_t)i);
                const bool asleep = slp->isl.has_node((uint32_t)i) && slp->isl.body_sleeping[i];
                o.body_flags = asleep ? (o.body_flags | AVN_BODY_SLEEPING) : (o.body_flags & (uint8_t)~AVN_BODY_SLEEPING);
                o.has_solver_body = o.rb_type != AVN_RB_STATIC && o.active();
                if (slp->timer.size() < n) slp->timer.resize(n, 0.0f);   // spawned inside the loop: SleepTimer 0, the world's thresholds, not SleepingDisabled
                if (!slp->lin.empty() && slp->lin.size() < n) slp->lin.resize(n, slp->p.linear_threshold);
                if (!slp->ang.empty() && slp->ang.size() < n) slp->ang.resize(n, slp->p.angular_threshold);
                if (!slp->disabled.empty() && slp->disabled.size() < n) slp->disabled.resize(n, 0);
            }
        }
        return AVN_OK;
    }
    // AccumulatedLocalAcceleration per body (forces/mod.rs:661-673); empty = no body has one (header: avn_local_accelerations_upload)
    std::vector<V3<S>> local_acc_linear, local_acc_angular;
    avn_status local_accelerations_upload(uint32_t count, const void* linear, const void* angular) override {
        if (count == 0 || (!linear && !angular)) { local_acc_linear.clear(); local_acc_angular.clear(); return AVN_OK; }
        if (count != bodies.size()) { error = "local_accelerations_upload: count differs from the last bodies_upload"; return AVN_ERR_BAD_ARG; }
        local_acc_linear.resize(count); local_acc_angular.resize(count);
        for (size_t i = 0; i < count; ++i) { local_acc_linear[i] = rd3(linear, i); local_acc_angular[i] = rd3(angular, i); }
        return AVN_OK;
    }
    avn_status bodies_download(const avn_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < bodies.size(); ++i) {
            wr3(o->position, i, bodies[i].position);
            if (o->rotation) { S* q = (S*)o->rotation + 4 * i; Q4<S> r = bodies[i].rotation; q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
            wr3(o->linear_velocity, i, bodies[i].linear_velocity);
            wr3(o->angular_velocity, i, bodies[i].angular_velocity);
        }
        return AVN_OK;
    }
    avn_status solver_bodies_download(const avn_solver_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < bodies.size(); ++i) {
            const Body<S>& b = bodies[i];
            wr3(o->linear_velocity, i, b.sb.linear_velocity);
            wr3(o->angular_velocity, i, b.sb.angular_velocity);
            wr3(o->delta_position, i, b.sb.delta_position);
            if (o->delta_rotation) { S* q = (S*)o->delta_rotation + 4 * i; Q4<S> r = b.sb.delta_rotation; q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
            if (o->flags) o->flags[i] = b.sb.flags | (b.has_solver_body ? 0u : 0x80000000u);
            if (o->inv_mass) ((S*)o->inv_mass)[i] = b.si.inv_mass;
            if (o->inv_inertia_world) { S* t = (S*)o->inv_inertia_world + 6 * i; const Sym3<S>& s = b.si.inv_inertia; t[0] = s.m00; t[1] = s.m01; t[2] = s.m02; t[3] = s.m11; t[4] = s.m12; t[5] = s.m22; }
            if (o->dominance) o->dominance[i] = b.si.dominance;
            wr3(o->linear_increment, i, b.vid.linear_increment);
            wr3(o->angular_increment, i, b.vid.angular_increment);
            if (o->linear_damping_rhs) ((S*)o->linear_damping_rhs)[i] = b.vid.linear_damping_rhs;
            if (o->angular_damping_rhs) ((S*)o->angular_damping_rhs)[i] = b.vid.angular_damping_rhs;
        }
        return AVN_OK;
    }

    avn_status manifolds_upload(const avn_manifolds* m) override {
        if (!m || !m->color_offsets || (m->count && (!m->body1 || !m->body2 || !m->normal || !m->friction || !m->restitution ||
                                                    !m->point_count || !m->anchor1 || !m->anchor2 || !m->penetration || !m->normal_speed))) {
            error = "manifolds_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        if (m->color_offsets[0] != 0 || m->color_offsets[AVN_GRAPH_COLOR_COUNT] != m->count) { error = "manifolds_upload: bad color_offsets"; return AVN_ERR_BAD_ARG; }
        use_handles = false;  // the manifolds come from the host again (not from the contact table)
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            if (m->color_offsets[c] > m->color_offsets[c + 1]) { error = "manifolds_upload: color_offsets not monotone"; return AVN_ERR_BAD_ARG; }
        std::memcpy(color_offsets, m->color_offsets, sizeof color_offsets);
        manifolds.resize(m->count);
        for (size_t i = 0; i < m->count; ++i) {
            ContactManifold<S>& o = manifolds[i];
            o.body1 = m->body1[i]; o.body2 = m->body2[i];
            if (o.body1 < 0 || o.body2 < 0 || (size_t)o.body1 >= bodies.size() || (size_t)o.body2 >= bodies.size()) { error = "manifolds_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
            o.normal = rd3(m->normal, i);
            o.tangent_velocity = rd3(m->tangent_velocity, i);
            o.friction = ((const S*)m->friction)[i];
            o.restitution = ((const S*)m->restitution)[i];
            o.point_count = m->point_count[i];
            if (o.point_count > AVN_MAX_MANIFOLD_POINTS) { error = "manifolds_upload: point_count > 4"; return AVN_ERR_BAD_ARG; }
            o.flags = rd<uint8_t>(m->manifold_flags, i, AVN_MANIFOLD_GENERATES_CONSTRAINTS);
            for (int p = 0; p < AVN_MAX_MANIFOLD_POINTS; ++p) {
                size_t s = 4 * i + p;
                ContactPoint<S>& cp = o.points[p];
                cp.anchor1 = rd3(m->anchor1, s);
                cp.anchor2 = rd3(m->anchor2, s);
                cp.penetration = ((const S*)m->penetration)[s];
                cp.normal_speed = ((const S*)m->normal_speed)[s];
                cp.warm_start_normal_impulse = rd<S>(m->warm_start_normal_impulse, s, 0);
                cp.normal_impulse = 0;
                if (m->warm_start_tangent_impulse) { const S* t = (const S*)m->warm_start_tangent_impulse + 2 * s; cp.warm_start_tangent_impulse = {t[0], t[1]}; }
                else cp.warm_start_tangent_impulse = {0, 0};
            }
        }
        for (auto& v : color_constraints) v.clear();
        return AVN_OK;
    }
    avn_status impulses_download(const avn_impulses_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < manifolds.size(); ++i)
            for (int p = 0; p < AVN_MAX_MANIFOLD_POINTS; ++p) {
                size_t s = 4 * i + p;
                const ContactPoint<S>& cp = manifolds[i].points[p];
                if (o->warm_start_normal_impulse) ((S*)o->warm_start_normal_impulse)[s] = cp.warm_start_normal_impulse;
                if (o->warm_start_tangent_impulse) { S* t = (S*)o->warm_start_tangent_impulse + 2 * s; t[0] = cp.warm_start_tangent_impulse.x; t[1] = cp.warm_start_tangent_impulse.y; }
                if (o->normal_impulse) ((S*)o->normal_impulse)[s] = cp.normal_impulse;
            }
        return AVN_OK;
    }
    avn_status constraints_download(const avn_constraints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t M = manifolds.size();
        if (o->point_count) std::memset(o->point_count, 0, M);
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            for (const ContactConstraint<S>& k : color_constraints[c]) {
                size_t i = k.manifold;
                if (o->point_count) o->point_count[i] = (uint8_t)k.point_count;
                if (o->relative_dominance) o->relative_dominance[i] = k.relative_dominance;
                wr3(o->tangent1, i, k.tangent1);
                if (o->softness_non_dynamic) o->softness_non_dynamic[i] = k.softness_non_dynamic;
                for (int p = 0; p < k.point_count; ++p) {
                    size_t s = 4 * i + p;
                    const ContactConstraintPoint<S>& cp = k.points[p];
                    wr3(o->anchor1, s, cp.anchor1);
                    if (o->initial_separation) ((S*)o->initial_separation)[s] = cp.initial_separation;
                    if (o->normal_impulse) ((S*)o->normal_impulse)[s] = cp.normal_part.impulse;
                    if (o->total_impulse) ((S*)o->total_impulse)[s] = cp.normal_part.total_impulse;
                    if (o->normal_effective_mass) ((S*)o->normal_effective_mass)[s] = cp.normal_part.effective_mass;
                    if (o->tangent_impulse) { S* t = (S*)o->tangent_impulse + 2 * s; t[0] = cp.tangent_part.impulse.x; t[1] = cp.tangent_part.impulse.y; }
                    if (o->tangent_effective_inverse_mass) { S* t = (S*)o->tangent_effective_inverse_mass + 3 * s; for (int q = 0; q < 3; ++q) t[q] = cp.tangent_part.effective_inverse_mass[q]; }
                }
            }
        return AVN_OK;
    }

    avn_status distance_joints_upload(const avn_distance_joints* j) override {
        if (!j || (j->count && (!j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->limit_min || !j->limit_max || !j->compliance))) {
            error = "distance_joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        std::vector<uint8_t> types(j->count, (uint8_t)AVN_JOINT_DISTANCE);
        std::vector<S> comp(3 * (size_t)j->count, S(0));
        for (size_t i = 0; i < j->count; ++i) comp[3 * i] = ((const S*)j->compliance)[i];
        avn_joints g;
        std::memset(&g, 0, sizeof g);
        g.count = j->count; g.joint_type = types.data(); g.body1 = j->body1; g.body2 = j->body2;
        g.local_anchor1 = j->local_anchor1; g.local_anchor2 = j->local_anchor2; g.limit_min = j->limit_min; g.limit_max = j->limit_max;
        g.compliance = comp.data(); g.damping_linear = j->damping_linear; g.damping_angular = j->damping_angular;
        g.collision_disabled = j->collision_disabled;
        return joints_upload(&g);
    }
    avn_status joints_upload(const avn_joints* j) override {
        if (!j || (j->count && (!j->joint_type || !j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->compliance))) {
            error = "joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        if (slp) {   // PhysicsIslands::add_joint links when a joint is added (islands/mod.rs:668-735): with sleeping on an upload may only APPEND joints
            if (j->count < joints.size()) { error = "joints_upload: with avn_sleeping_enable on, joints can only be appended (avn_sleeping_enable(NULL) first to change the set)"; return AVN_ERR_STATE; }
            for (size_t i = 0; i < joints.size(); ++i)
                if (joints[i].body1 != j->body1[i] || joints[i].body2 != j->body2[i]) { error = "joints_upload: with avn_sleeping_enable on, the bodies of an existing joint cannot change"; return AVN_ERR_STATE; }
            for (size_t i = joints.size(); i < j->count; ++i) {
                if (j->body1[i] < 0 || j->body2[i] < 0 || (size_t)j->body1[i] >= bodies.size() || (size_t)j->body2[i] >= bodies.size() || j->body1[i] == j->body2[i]) { error = "joints_upload: bad body index"; return AVN_ERR_BAD_ARG; }
                slp->isl.joint_add((uint32_t)i, (uint32_t)j->body1[i], (uint32_t)j->body2[i]);
            }
        }
        joints.resize(j->count);
        collision_disabled_bodies.clear();
        for (size_t i = 0; i < j->count; ++i) {
            Joint<S>& o = joints[i];
            o = Joint<S>();
            o.type = j->joint_type[i];
            if (o.type >= AVN_JOINT_TYPE_COUNT) { error = "joints_upload: bad joint_type"; return AVN_ERR_BAD_ARG; }
            o.body1 = j->body1[i]; o.body2 = j->body2[i];
            if (o.body1 < 0 || o.body2 < 0 || (size_t)o.body1 >= bodies.size() || (size_t)o.body2 >= bodies.size() || o.body1 == o.body2) { error = "joints_upload: bad body index"; return AVN_ERR_BAD_ARG; }
            o.local_anchor1 = rd3(j->local_anchor1, i);
            o.local_anchor2 = rd3(j->local_anchor2, i);
            if (j->local_basis1) { const S* q = (const S*)j->local_basis1 + 4 * i; o.local_basis1 = {q[0], q[1], q[2], q[3]}; }
            if (j->local_basis2) { const S* q = (const S*)j->local_basis2 + 4 * i; o.local_basis2 = {q[0], q[1], q[2], q[3]}; }
            if (j->axis) o.axis = rd3(j->axis, i);
            else o.axis = o.type == AVN_JOINT_REVOLUTE ? V3<S>{0, 0, 1} : o.type == AVN_JOINT_SPHERICAL ? V3<S>{0, 1, 0} : V3<S>{1, 0, 0};  // DEFAULT_*_AXIS
            o.limit_min = rd<S>(j->limit_min, i, 0); o.limit_max = rd<S>(j->limit_max, i, 0);
            o.limit2_min = rd<S>(j->limit2_min, i, 0); o.limit2_max = rd<S>(j->limit2_max, i, 0);
            o.limit_flags = rd<uint8_t>(j->limit_flags, i, 0);
            const S* c = (const S*)j->compliance + 3 * i;
            o.compliance = c[0]; o.compliance1 = c[1]; o.compliance2 = c[2];
            o.has_damping = j->damping_linear && j->damping_angular;
            o.damping_linear = rd<S>(j->damping_linear, i, 0);
            o.damping_angular = rd<S>(j->damping_angular, i, 0);
            o.collision_disabled = rd<uint8_t>(j->collision_disabled, i, 0) != 0;
            if (o.collision_disabled) collision_disabled_bodies.insert(pair_key((uint32_t)o.body1, (uint32_t)o.body2));
        }
        joint_order.resize(j->count);
        for (uint32_t i = 0; i < j->count; ++i) joint_order[i] = i;
        std::stable_sort(joint_order.begin(), joint_order.end(), [&](uint32_t a, uint32_t b) { return joints[a].type < joints[b].type; });
        return AVN_OK;
    }
    avn_status joints_download(const avn_joints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < joints.size(); ++i) {
            wr3(o->world_r1, i, joints[i].world_r1);
            wr3(o->world_r2, i, joints[i].world_r2);
            wr3(o->center_difference, i, joints[i].center_difference);
            wr3(o->total_lagrange, i, joints[i].total_lagrange);
            wr3(o->total_rotation_lagrange, i, joints[i].total_rot0 + joints[i].total_rot1);
            wr3(o->torque, i, joints[i].torque);
            wr3(o->force, i, joints[i].force);
        }
        return AVN_OK;
    }

    // =============================================================================================
    //                                      SOLVER BODIES
    // =============================================================================================
    // solver_body/plugin.rs:173-251
    // Query::par_iter_mut of the per-body systems: disjoint bodies, any order
    template <class F> void par_bodies(F f) {
        pool.par_for_each(bodies.size(), 64, [&](size_t b0, size_t b1) { for (size_t i = b0; i < b1; ++i) f(i, bodies[i]); });
    }
    void prepare_solver_bodies() {
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body) return;
            b.sb.linear_velocity = b.linear_velocity;
            b.sb.angular_velocity = b.angular_velocity;
            b.sb.delta_position = vzero<S>();
            b.sb.delta_rotation = qidentity<S>();
            // SolverBodyInertia::new (solver_body/mod.rs:378-423)
            Sym3<S> inv_inertia = rotated_inverse_inertia(b.inv_inertia_local, b.rotation);
            uint16_t flags = b.locked_axes;
            if (b.inv_mass == S(0)) flags |= 1 << 6;
            if (sym_is_zero(inv_inertia)) flags |= 1 << 7;
            lock_rotation_axes(inv_inertia, b.locked_axes);
            b.si.inv_mass = b.inv_mass;
            b.si.inv_inertia = inv_inertia;
            b.si.dominance = b.rb_type == AVN_RB_DYNAMIC ? (int16_t)b.dominance : (int16_t)128;
            b.si.flags = flags;
            b.sb.flags = b.locked_axes;
            if (b.rb_type == AVN_RB_KINEMATIC) b.sb.flags |= AVN_SB_KINEMATIC;
            bool rotation_locked = (b.locked_axes & 0b111) == 0b111;
            bool is_gyroscopic = !rotation_locked && !sym_is_isotropic(b.inv_inertia_local, S(1e-6));
            if (is_gyroscopic) b.sb.flags |= AVN_SB_GYROSCOPIC;
        });
    }
    // integrator/mod.rs:260-313.  VelocityIntegrationData.{linear,angular}_increment hold the accumulated
    // accelerations written by ForcePlugin before this system (here: the uploaded accel_* arrays).
    void pre_process_velocity_increments() {
        S delta_secs = h_f64cast;
        V3<S> gravity{(S)cfg.gravity[0], (S)cfg.gravity[1], (S)cfg.gravity[2]};
        par_bodies([&](size_t i, Body<S>& b) {
            if (b.rb_type != AVN_RB_DYNAMIC) return;
            b.vid.linear_increment = accel_linear[i];
            b.vid.angular_increment = accel_angular[i];
            b.vid.linear_damping_rhs = S(1) / (S(1) + delta_secs * b.linear_damping);
            b.vid.angular_damping_rhs = S(1) / (S(1) + delta_secs * b.angular_damping);
            b.vid.linear_increment = b.vid.linear_increment + gravity * b.gravity_scale;
            if (b.locked_axes & 0b100000) b.vid.linear_increment.x = 0;
            if (b.locked_axes & 0b010000) b.vid.linear_increment.y = 0;
            if (b.locked_axes & 0b001000) b.vid.linear_increment.z = 0;
            if (b.locked_axes & 0b000100) b.vid.angular_increment.x = 0;
            if (b.locked_axes & 0b000010) b.vid.angular_increment.y = 0;
            if (b.locked_axes & 0b000001) b.vid.angular_increment.z = 0;
            b.vid.linear_increment = b.vid.linear_increment * delta_secs;
            b.vid.angular_increment = b.vid.angular_increment * delta_secs;
        });
    }
    // integrator/mod.rs:316-328
    void clear_velocity_increments() {
        par_bodies([&](size_t, Body<S>& b) {
            if (b.has_solver_body) { b.vid.linear_increment = vzero<S>(); b.vid.angular_increment = vzero<S>(); }
        });
    }
    // integrator/mod.rs:403-460
    static void solve_gyroscopic_torque(V3<S>& ang_vel, Q4<S> rotation, const Sym3<S>& local_inverse_inertia, S delta_secs) {
        V3<S> local_ang_vel = qrot(qinverse(rotation), ang_vel);
        Sym3<S> tensor = sym_inverse_or_zero(local_inverse_inertia);  // ComputedAngularInertia::tensor() :617-619
        V3<S> local_momentum = smul(tensor, local_ang_vel);
        V3<S> new_local_momentum = local_momentum - delta_secs * cross(local_ang_vel, local_momentum);
        S new_len_sq = length_squared(new_local_momentum);
        if (new_len_sq == S(0)) { ang_vel = vzero<S>(); return; }
        new_local_momentum = new_local_momentum * std::sqrt(length_squared(local_momentum) / new_len_sq);
        ang_vel = qrot(rotation, smul(local_inverse_inertia, new_local_momentum));
    }
    // integrator/mod.rs:343-391, then clamp_velocities :467-500
    void integrate_velocities() {
        S delta_secs = h_f64cast;
        // apply_local_acceleration, forces/plugin.rs:207-241 (ForceSystems::ApplyLocalAcceleration: in IntegrationSystems::Velocity, before integrate_velocities,
        // :34-38): every (SolverBody, AccumulatedLocalAcceleration, Rotation) Without<CustomVelocityIntegration> -- kinematic bodies too; apply_to_vec (the
        // TRANSLATION locks, locked_axes.rs:230-243) masks both vectors, as written there
        if (!local_acc_linear.empty())
            par_bodies([&](size_t i, Body<S>& b) {
                if (!b.has_solver_body || (b.body_flags & AVN_BODY_CUSTOM_VELOCITY_INTEGRATION)) return;
                const Q4<S> rotation = qmul(b.sb.delta_rotation, b.rotation);
                const uint32_t locked = b.sb.flags & 0x3Fu;
                auto apply_to_vec = [&](V3<S> v) { if (locked & 0x20u) v.x = S(0); if (locked & 0x10u) v.y = S(0); if (locked & 0x08u) v.z = S(0); return v; };
                const V3<S> world_linear_acceleration = apply_to_vec(qrot(rotation, local_acc_linear[i]));
                const V3<S> world_angular_acceleration = apply_to_vec(qrot(rotation, local_acc_angular[i]));
                b.sb.linear_velocity = b.sb.linear_velocity + world_linear_acceleration * delta_secs;
                b.sb.angular_velocity = b.sb.angular_velocity + world_angular_acceleration * delta_secs;
            });
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body || (b.body_flags & AVN_BODY_CUSTOM_VELOCITY_INTEGRATION)) return;
            if (b.sb.is_kinematic()) return;
            b.sb.linear_velocity = b.sb.linear_velocity * b.vid.linear_damping_rhs;
            b.sb.angular_velocity = b.sb.angular_velocity * b.vid.angular_damping_rhs;
            b.sb.linear_velocity = b.sb.linear_velocity + b.vid.linear_increment;
            b.sb.angular_velocity = b.sb.angular_velocity + b.vid.angular_increment;
            if (b.sb.is_gyroscopic()) {
                Q4<S> rotation = qmul(b.sb.delta_rotation, b.rotation);
                solve_gyroscopic_torque(b.sb.angular_velocity, rotation, b.inv_inertia_local, delta_secs);
            }
        });
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body) return;
            if (b.max_linear_speed >= S(0)) {
                S sq = length_squared(b.sb.linear_velocity);
                if (sq > b.max_linear_speed * b.max_linear_speed) b.sb.linear_velocity = b.sb.linear_velocity * (b.max_linear_speed / std::sqrt(sq));
            }
        });
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body) return;
            if (b.max_angular_speed >= S(0)) {
                S sq = length_squared(b.sb.angular_velocity);
                if (sq > b.max_angular_speed * b.max_angular_speed) b.sb.angular_velocity = b.sb.angular_velocity * (b.max_angular_speed / std::sqrt(sq));
            }
        });
    }
    // integrator/mod.rs:503-535, then update_solver_body_angular_inertia solver_body/plugin.rs:287-295
    void integrate_positions() {
        S delta_secs = h_adj;
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body || (b.body_flags & AVN_BODY_CUSTOM_POSITION_INTEGRATION)) return;
            b.sb.delta_position = b.sb.delta_position + b.sb.linear_velocity * delta_secs;
            b.sb.delta_rotation = qmul(from_scaled_axis(b.sb.angular_velocity * delta_secs), b.sb.delta_rotation);
        });
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body) return;
            // update_effective_inv_angular_inertia (solver_body/mod.rs:473-501): uses the step-start Rotation
            Sym3<S> t = rotated_inverse_inertia(b.inv_inertia_local, b.rotation);
            lock_rotation_axes(t, b.si.flags & 0x3F);
            b.si.inv_inertia = t;
        });
    }
    // solver_body/plugin.rs:255-284
    void writeback_solver_bodies() {
        par_bodies([&](size_t, Body<S>& b) {
            if (!b.has_solver_body) return;
            V3<S> old_world_com = qrot(b.rotation, b.center_of_mass);
            b.rotation = fast_renormalize(qmul(b.sb.delta_rotation, b.rotation));
            V3<S> new_world_com = qrot(b.rotation, b.center_of_mass);
            b.position = b.position + ((b.sb.delta_position + old_world_com) - new_world_com);
            b.linear_velocity = b.sb.linear_velocity;
            b.angular_velocity = b.sb.angular_velocity;
        });
    }

    // =============================================================================================
    //                                      CONTACT SOLVER
    // =============================================================================================
    // Dummy bodies (solver/plugin.rs:491-505): static/sleeping/missing bodies use a fresh local DUMMY.
    struct BodyRef { SolverBody<S>* body; const SolverBodyInertia<S>* inertia; };
    struct Dummies { SolverBody<S> body[2]; };   // the caller's two local `let mut dummy = SolverBody::DUMMY` (plugin.rs:491-505)
    SolverBodyInertia<S> dummy_inertia;
    BodyRef solver_ref(int32_t idx, int which, Dummies& d) {
        Body<S>& b = bodies[idx];
        if (b.has_solver_body) return {&b.sb, &b.si};
        d.body[which] = SolverBody<S>();
        return {&d.body[which], &dummy_inertia};
    }

    // ContactConstraint::generate, contact/mod.rs:110-220; driver solver/plugin.rs:363-448
    void prepare_contact_constraints() {
        if (use_handles) gather_manifolds_from_handles();
        update_contact_softness();  // runs .before(NarrowPhase) every step, plugin.rs:108,326-350
        bool warm = cfg.match_contacts != 0;
        uint32_t counts[AVN_GRAPH_COLOR_COUNT] = {};
        // "parallelizing over graph colors": par_for_each(&mut active_colors, 2, ..) (plugin.rs:387-388) -- one task per colour chunk
        pool.par_for_each(AVN_GRAPH_COLOR_COUNT, 2, [&](size_t c0, size_t c1) {
        for (int c = (int)c0; c < (int)c1; ++c) {
            uint32_t count = 0;
            color_constraints[c].clear();
            for (uint32_t mi = color_offsets[c]; mi < color_offsets[c + 1]; ++mi) {
                const ContactManifold<S>& m = manifolds[mi];
                if (!(m.flags & AVN_MANIFOLD_GENERATES_CONSTRAINTS)) continue;
                const Body<S>& b1 = bodies[m.body1];
                const Body<S>& b2 = bodies[m.body2];
                if (!b1.active() || !b2.active()) continue;  // bodies.get() with RigidBodyActiveFilter fails
                if (b1.rb_type != AVN_RB_DYNAMIC && b2.rb_type != AVN_RB_DYNAMIC) continue;
                static const SolverBodyInertia<S> DUMMY;
                const SolverBodyInertia<S>& inertia1 = b1.has_solver_body ? b1.si : DUMMY;
                const SolverBodyInertia<S>& inertia2 = b2.has_solver_body ? b2.si : DUMMY;
                int16_t relative_dominance = inertia1.dominance - inertia2.dominance;
                V3<S> inv_mass1, inv_mass2;
                Sym3<S> i1, i2;
                if (relative_dominance == 0) { inv_mass1 = inertia1.effective_inv_mass(); i1 = inertia1.inv_inertia; inv_mass2 = inertia2.effective_inv_mass(); i2 = inertia2.inv_inertia; }
                else if (relative_dominance > 0) { inv_mass1 = vzero<S>(); i1 = sym_zero<S>(); inv_mass2 = inertia2.effective_inv_mass(); i2 = inertia2.inv_inertia; }
                else { inv_mass1 = inertia1.effective_inv_mass(); i1 = inertia1.inv_inertia; inv_mass2 = vzero<S>(); i2 = sym_zero<S>(); }
                SoftnessCoefficients<S> softness = relative_dominance != 0 ? soft_non_dynamic : soft_dynamic;
                V3<S> w_sum = inv_mass1 + inv_mass2;
                V3<S> t0, t1;
                compute_tangent_directions(m.normal, b1.linear_velocity, b2.linear_velocity, t0, t1);
                ContactConstraint<S> k;
                k.body1 = m.body1; k.body2 = m.body2;
                k.relative_dominance = relative_dominance;
                k.friction = m.friction; k.restitution = m.restitution;
                k.tangent_velocity = m.tangent_velocity;
                k.normal = m.normal; k.tangent1 = t0;
                k.point_count = m.point_count;
                k.manifold = mi;
                k.softness_non_dynamic = relative_dominance != 0;
                for (int p = 0; p < m.point_count; ++p) {
                    const ContactPoint<S>& cp = m.points[p];
                    ContactConstraintPoint<S>& o = k.points[p];
                    o.normal_part = normal_part_generate(w_sum, i1, i2, cp.anchor1, cp.anchor2, m.normal, warm, cp.warm_start_normal_impulse, softness);
                    o.has_tangent = m.friction > S(0);
                    if (o.has_tangent) o.tangent_part = tangent_part_generate(w_sum, i1, i2, cp.anchor1, cp.anchor2, t0, t1, warm, cp.warm_start_tangent_impulse);
                    else o.tangent_part = ContactTangentPart<S>{{0, 0}, {0, 0, 0}};
                    o.anchor1 = cp.anchor1; o.anchor2 = cp.anchor2;
                    o.normal_speed = cp.normal_speed;
                    o.initial_separation = -cp.penetration - dot(cp.anchor2 - cp.anchor1, m.normal);
                }
                if (k.point_count > 0) { color_constraints[c].push_back(k); ++count; }
            }
            counts[c] = count;
        }
        });
        uint32_t count = 0;
        for (uint32_t n : counts) count += n;
        last_timers.contact_constraint_count = count;
    }

    // Colour iteration order shared by warm_start / solve_contacts / solve_restitution:
    // overflow colour serially FIRST, then colours 0..22 (solver/plugin.rs:461-479).
    // ... the constraints of one colour 0..22 touch disjoint bodies: par_for_each(&mut color.contact_constraints, 64, ..)
    // (plugin.rs:476,564,662)
    int only_color = -1;   // >= 0: the pass functions below visit this colour only (avn_run_color_pass, level-2 sharding)
    int only_level = -1;   // with only_color == overflow and levels uploaded (avn_halo_overflow_levels_upload): this level of the overflow colour only, in list order
    std::vector<uint32_t> l2_level_of;
    uint32_t l2_levels = 1;
    template <class F> void for_each_constraint_in_solver_order(F f) {
        if (only_color < 0 || only_color == AVN_COLOR_OVERFLOW_INDEX) {
            std::vector<ContactConstraint<S>>& ov = color_constraints[AVN_COLOR_OVERFLOW_INDEX];
            for (size_t i = 0; i < ov.size(); ++i) if (only_level < 0 || (i < l2_level_of.size() && (int)l2_level_of[i] == only_level)) f(ov[i]);
        }
        for (int c = 0; c < AVN_COLOR_OVERFLOW_INDEX; ++c) {
            if (only_color >= 0 && only_color != c) continue;
            std::vector<ContactConstraint<S>>& v = color_constraints[c];
            pool.par_for_each(v.size(), 64, [&](size_t b0, size_t b1) { for (size_t i = b0; i < b1; ++i) f(v[i]); });
        }
    }
    void resolve(ContactConstraint<S>& k, BodyRef& r1, BodyRef& r2, Dummies& d) {
        r1 = solver_ref(k.body1, 0, d);
        r2 = solver_ref(k.body2, 1, d);
        if (k.relative_dominance > 0) r1.inertia = &dummy_inertia;       // plugin.rs:508-512
        else if (k.relative_dominance < 0) r2.inertia = &dummy_inertia;
    }
    static void tangent_directions(const ContactConstraint<S>& k, V3<S>& t0, V3<S>& t1) {  // contact/mod.rs:411-421
        t0 = k.tangent1;
        t1 = cross(k.tangent1, k.normal);
    }
    // contact/mod.rs:223-264
    void warm_start() {
        S coeff = (S)cfg.warm_start_coefficient;
        for_each_constraint_in_solver_order([&](ContactConstraint<S>& k) {
            Dummies dm; BodyRef r1, r2; resolve(k, r1, r2, dm);
            V3<S> inv_mass1 = r1.inertia->effective_inv_mass(), inv_mass2 = r2.inertia->effective_inv_mass();
            const Sym3<S>& ii1 = r1.inertia->inv_inertia; const Sym3<S>& ii2 = r2.inertia->inv_inertia;
            V3<S> t0, t1; tangent_directions(k, t0, t1);
            for (int p = 0; p < k.point_count; ++p) {
                ContactConstraintPoint<S>& pt = k.points[p];
                V3<S> ra = pt.anchor1, rb = pt.anchor2;
                V2<S> ti = pt.has_tangent ? pt.tangent_part.impulse : V2<S>{0, 0};
                V3<S> imp = coeff * ((pt.normal_part.impulse * k.normal + ti.x * t0) + ti.y * t1);
                r1.body->linear_velocity = r1.body->linear_velocity - cmul(imp, inv_mass1);
                r1.body->angular_velocity = r1.body->angular_velocity - smul(ii1, cross(ra, imp));
                r2.body->linear_velocity = r2.body->linear_velocity + cmul(imp, inv_mass2);
                r2.body->angular_velocity = r2.body->angular_velocity + smul(ii2, cross(rb, imp));
            }
        });
    }
    // contact/mod.rs:267-354
    void solve_contacts(bool use_bias) {
        S delta_secs = h_adj;
        S max_overlap_solve_speed = (S)cfg.max_overlap_solve_speed * (S)cfg.length_unit;
        for_each_constraint_in_solver_order([&](ContactConstraint<S>& k) {
            Dummies dm; BodyRef r1, r2; resolve(k, r1, r2, dm);
            SolverBody<S>& body1 = *r1.body; SolverBody<S>& body2 = *r2.body;
            V3<S> inv_mass1 = r1.inertia->effective_inv_mass(), inv_mass2 = r2.inertia->effective_inv_mass();
            const Sym3<S>& ii1 = r1.inertia->inv_inertia; const Sym3<S>& ii2 = r2.inertia->inv_inertia;
            V3<S> delta_translation = body2.delta_position - body1.delta_position;
            for (int p = 0; p < k.point_count; ++p) {
                ContactConstraintPoint<S>& pt = k.points[p];
                V3<S> r1w = qrot(body1.delta_rotation, pt.anchor1);
                V3<S> r2w = qrot(body2.delta_rotation, pt.anchor2);
                V3<S> delta_separation = delta_translation + (r2w - r1w);
                S separation = dot(delta_separation, k.normal) + pt.initial_separation;
                V3<S> ra = pt.anchor1, rb = pt.anchor2;
                V3<S> relative_velocity = body2.velocity_at_point(rb) - body1.velocity_at_point(ra);
                S mag = normal_part_solve_impulse(pt.normal_part, separation, relative_velocity, k.normal, use_bias, max_overlap_solve_speed, delta_secs);
                V3<S> imp = mag * k.normal;
                body1.linear_velocity = body1.linear_velocity - cmul(imp, inv_mass1);
                body1.angular_velocity = body1.angular_velocity - smul(ii1, cross(ra, imp));
                body2.linear_velocity = body2.linear_velocity + cmul(imp, inv_mass2);
                body2.angular_velocity = body2.angular_velocity + smul(ii2, cross(rb, imp));
            }
            V3<S> t0, t1; tangent_directions(k, t0, t1);
            for (int p = 0; p < k.point_count; ++p) {
                ContactConstraintPoint<S>& pt = k.points[p];
                if (!pt.has_tangent) continue;
                V3<S> ra = pt.anchor1, rb = pt.anchor2;
                V3<S> relative_velocity = body2.velocity_at_point(rb) - body1.velocity_at_point(ra);
                V3<S> imp = tangent_part_solve_impulse(pt.tangent_part, t0, t1, relative_velocity, k.tangent_velocity, k.friction, pt.normal_part.impulse);
                body1.linear_velocity = body1.linear_velocity - cmul(imp, inv_mass1);
                body1.angular_velocity = body1.angular_velocity - smul(ii1, cross(ra, imp));
                body2.linear_velocity = body2.linear_velocity + cmul(imp, inv_mass2);
                body2.angular_velocity = body2.angular_velocity + smul(ii2, cross(rb, imp));
            }
        });
    }
    // contact/mod.rs:358-407; driver solver/plugin.rs:630-718
    void solve_restitution() {
        S threshold = (S)cfg.restitution_threshold * (S)cfg.length_unit;
        for_each_constraint_in_solver_order([&](ContactConstraint<S>& k) {
            if (k.restitution == S(0)) return;
            Dummies dm; BodyRef r1, r2; resolve(k, r1, r2, dm);
            SolverBody<S>& body1 = *r1.body; SolverBody<S>& body2 = *r2.body;
            V3<S> inv_mass1 = r1.inertia->effective_inv_mass(), inv_mass2 = r2.inertia->effective_inv_mass();
            const Sym3<S>& ii1 = r1.inertia->inv_inertia; const Sym3<S>& ii2 = r2.inertia->inv_inertia;
            uint32_t iterations = k.point_count > 1 ? cfg.restitution_iterations : 1;
            for (uint32_t it = 0; it < iterations; ++it)
                for (int p = 0; p < k.point_count; ++p) {
                    ContactConstraintPoint<S>& pt = k.points[p];
                    if (pt.normal_speed > -threshold || pt.normal_part.total_impulse == S(0)) continue;
                    V3<S> ra = pt.anchor1, rb = pt.anchor2;
                    V3<S> relative_velocity = body2.velocity_at_point(rb) - body1.velocity_at_point(ra);
                    S normal_speed = dot(relative_velocity, k.normal);
                    S impulse = -pt.normal_part.effective_mass * (normal_speed + k.restitution * pt.normal_speed);
                    S new_impulse = smax(pt.normal_part.impulse + impulse, S(0));
                    impulse = new_impulse - pt.normal_part.impulse;
                    pt.normal_part.impulse = new_impulse;
                    pt.normal_part.total_impulse += impulse;
                    V3<S> imp = impulse * k.normal;
                    body1.linear_velocity = body1.linear_velocity - cmul(imp, inv_mass1);
                    body1.angular_velocity = body1.angular_velocity - smul(ii1, cross(ra, imp));
                    body2.linear_velocity = body2.linear_velocity + cmul(imp, inv_mass2);
                    body2.angular_velocity = body2.angular_velocity + smul(ii2, cross(rb, imp));
                }
        });
    }
    // solver/plugin.rs:722-755
    void store_contact_impulses() {
        store_contact_impulses_to_manifolds();
        if (use_handles) scatter_impulses_to_contacts();
    }
    void store_contact_impulses_to_manifolds() {
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
            std::vector<ContactConstraint<S>>& v = color_constraints[c];
            pool.par_for_each(v.size(), 64, [&](size_t b0, size_t b1) {
                for (size_t i = b0; i < b1; ++i) {
                    ContactConstraint<S>& k = v[i];
                    ContactManifold<S>& m = manifolds[k.manifold];
                    for (int p = 0; p < k.point_count; ++p) {
                        m.points[p].warm_start_normal_impulse = k.points[p].normal_part.impulse;
                        m.points[p].warm_start_tangent_impulse = k.points[p].has_tangent ? k.points[p].tangent_part.impulse : V2<S>{0, 0};
                        m.points[p].normal_impulse = k.points[p].normal_part.total_impulse;
                    }
                }
            });
        }
    }

    // =============================================================================================
    //                                      XPBD (DistanceJoint)
    // =============================================================================================
    // xpbd/plugin.rs:125-142 prepare_xpbd_joint<T> + the per-type prepare():
    //   point constraint  xpbd/joints/shared/point_constraint.rs:38-53     fixed angle  shared/fixed_angle_constraint.rs:38-57
    //   fixed fixed.rs:39-72   revolute revolute.rs:48-90   spherical spherical.rs:44-83   prismatic prismatic.rs:43-81
    //   distance distance.rs:36-59
    void prepare_joints() {
        for (Joint<S>& j : joints) {
            j.total_lagrange = vzero<S>(); j.total_rot0 = vzero<S>(); j.total_rot1 = vzero<S>();  // clear_lagrange_multipliers
            const Body<S>& b1 = bodies[j.body1];
            const Body<S>& b2 = bodies[j.body2];
            if ((b1.body_flags & AVN_BODY_DISABLED) || (b2.body_flags & AVN_BODY_DISABLED)) continue;  // Without<RigidBodyDisabled>
            V3<S> cd = (b2.position - b1.position) + (qrot(b2.rotation, b2.center_of_mass) - qrot(b1.rotation, b1.center_of_mass));
            if (j.type == AVN_JOINT_SPHERICAL) {
                M3<S> rot1_mat = mat3_from_quat(b1.rotation), rot2_mat = mat3_from_quat(b2.rotation);
                j.world_r1 = mmul(rot1_mat, j.local_anchor1 - b1.center_of_mass);
                j.world_r2 = mmul(rot2_mat, j.local_anchor2 - b2.center_of_mass);
                j.center_difference = cd;
                V3<S> swing_axis = any_orthonormal_vector(j.axis);
                j.v0 = mmul(rot1_mat, qrot(j.local_basis1, swing_axis));   // swing_axis1
                j.v1 = mmul(rot2_mat, qrot(j.local_basis2, swing_axis));   // swing_axis2
                j.v2 = mmul(rot1_mat, qrot(j.local_basis1, j.axis));       // twist_axis1
                j.v3 = mmul(rot2_mat, qrot(j.local_basis2, j.axis));       // twist_axis2
                continue;
            }
            j.world_r1 = qrot(b1.rotation, j.local_anchor1 - b1.center_of_mass);
            j.world_r2 = qrot(b2.rotation, j.local_anchor2 - b2.center_of_mass);
            j.center_difference = cd;
            if (j.type == AVN_JOINT_FIXED || j.type == AVN_JOINT_PRISMATIC)
                j.rotation_difference = qmul(qmul(b1.rotation, j.local_basis1), qinverse(qmul(b2.rotation, j.local_basis2)));
            if (j.type == AVN_JOINT_PRISMATIC) j.v0 = qrot(qmul(b1.rotation, j.local_basis1), j.axis);  // free_axis1
            if (j.type == AVN_JOINT_REVOLUTE) {
                Q4<S> q1 = qmul(b1.rotation, j.local_basis1), q2 = qmul(b2.rotation, j.local_basis2);
                V3<S> ortho = any_orthonormal_vector(j.axis);
                j.v0 = qrot(q1, j.axis); j.v1 = qrot(q2, j.axis); j.v2 = qrot(q1, ortho); j.v3 = qrot(q2, ortho);  // a1, a2, b1, b2
            }
        }
    }
    // XPBD queries use Without<RigidBodyDisabled> only (a sleeping body keeps no SolverBody anyway).
    SolverBody<S> xpbd_dummy[2];  // declared outside the joint loop in the reference (xpbd/plugin.rs:155-156)

    struct JointCtx {  // the [body1, body2] / [inertia1, inertia2] of one solve call
        SolverBody<S>* body1; SolverBody<S>* body2;
        V3<S> inv_mass1, inv_mass2;
        Sym3<S> ii1, ii2;
    };
    // positional_constraint.rs:10-51 apply_positional_impulse
    static void apply_positional_impulse(JointCtx& c, V3<S> impulse, V3<S> r1, V3<S> r2) {
        c.body1->delta_position = c.body1->delta_position + cmul(impulse, c.inv_mass1);
        c.body1->delta_rotation = qmul(from_scaled_axis(smul(c.ii1, cross(r1, impulse))), c.body1->delta_rotation);
        c.body2->delta_position = c.body2->delta_position - cmul(impulse, c.inv_mass2);
        c.body2->delta_rotation = qmul(from_scaled_axis(smul(c.ii2, cross(r2, -impulse))), c.body2->delta_rotation);
    }
    // positional_constraint.rs:68-82 compute_generalized_inverse_mass
    static S positional_w(S inv_mass_max, const Sym3<S>& ii, V3<S> r, V3<S> dir) {
        V3<S> rc = cross(r, dir);
        return inv_mass_max + dot(rc, smul(ii, rc));
    }
    // shared/point_constraint.rs:56-108 PointConstraintShared::solve
    void point_constraint_solve(Joint<S>& j, JointCtx& c, S compliance, S dt) {
        V3<S> world_r1 = qrot(c.body1->delta_rotation, j.world_r1);
        V3<S> world_r2 = qrot(c.body2->delta_rotation, j.world_r2);
        V3<S> separation = ((c.body2->delta_position - c.body1->delta_position) + (world_r2 - world_r1)) + j.center_difference;
        S magnitude_squared = length_squared(separation);
        if (magnitude_squared == S(0)) return;
        S magnitude = std::sqrt(magnitude_squared);
        V3<S> dir = (-separation) / magnitude;
        S w1 = positional_w(max_element(c.inv_mass1), c.ii1, world_r1, dir);
        S w2 = positional_w(max_element(c.inv_mass2), c.ii2, world_r2, dir);
        S delta_lagrange = compute_lagrange_update<S>(S(0), magnitude, w1, w2, compliance, dt);
        V3<S> impulse = delta_lagrange * dir;
        j.total_lagrange = j.total_lagrange + impulse;
        apply_positional_impulse(c, impulse, world_r1, world_r2);
    }
    // angular_constraint.rs:146-184 align_orientation (3D) with :50-93 apply_angular_lagrange_update / apply_angular_impulse
    static V3<S> align_orientation(JointCtx& c, V3<S> rotation_difference, S lagrange, S compliance, S dt) {
        S angle = length(rotation_difference);
        if (angle <= std::numeric_limits<S>::epsilon()) return vzero<S>();
        V3<S> axis = rotation_difference / angle;
        S w1 = dot(axis, smul(c.ii1, axis)), w2 = dot(axis, smul(c.ii2, axis));
        S delta_lagrange = compute_lagrange_update<S>(lagrange, angle, w1, w2, compliance, dt);
        if (!(std::fabs(delta_lagrange) <= std::numeric_limits<S>::epsilon())) {
            V3<S> impulse = (-delta_lagrange) * axis;
            c.body1->delta_rotation = qmul(from_scaled_axis(smul(c.ii1, impulse)), c.body1->delta_rotation);
            c.body2->delta_rotation = qmul(from_scaled_axis(smul(c.ii2, -impulse)), c.body2->delta_rotation);
        }
        return delta_lagrange * axis;
    }
    // shared/fixed_angle_constraint.rs:60-95 FixedAngleConstraintShared::solve (3D)
    void fixed_angle_solve(Joint<S>& j, JointCtx& c, S compliance, S dt) {
        Q4<S> q = qmul(qmul(j.rotation_difference, c.body1->delta_rotation), qinverse(c.body2->delta_rotation));
        V3<S> difference = S(-2) * V3<S>{q.x, q.y, q.z};
        j.total_rot0 = j.total_rot0 + align_orientation(c, difference, S(0), compliance, dt);
    }
    // dynamics/joints/mod.rs:427-472 AngleLimit::compute_correction (3D)
    static bool angle_limit_correction(S lim_min, S lim_max, V3<S> limit_axis, V3<S> axis1, V3<S> axis2, S max_correction, V3<S>& out) {
        const S PI = S(3.14159265358979323846264338327950288), TAU = S(6.28318530717958647692528676655900577);
        S phi = asin_s(dot(cross(axis1, axis2), limit_axis));
        if (dot(axis1, axis2) < S(0)) phi = PI - phi;
        if (phi > PI) phi -= TAU;
        if (phi < lim_min || phi > lim_max) {
            phi = clamp_s(phi, lim_min, lim_max);
            Q4<S> rot = from_axis_angle(limit_axis, phi);
            out = clamp_length_max(cross(qrot(rot, axis1), axis2), max_correction);
            return true;
        }
        return false;
    }
    // dynamics/joints/mod.rs:345-357 DistanceLimit::compute_correction_along_axis
    static V3<S> correction_along_axis(S lim_min, S lim_max, V3<S> separation, V3<S> axis) {
        S a = dot(separation, axis);
        if (a < lim_min) return axis * (lim_min - a);
        if (a > lim_max) return (-axis) * (a - lim_max);
        return vzero<S>();
    }
    void solve_fixed(Joint<S>& j, JointCtx& c, S dt) {      // fixed.rs:74-91: angle, then point
        fixed_angle_solve(j, c, j.compliance1, dt);
        point_constraint_solve(j, c, j.compliance, dt);
    }
    void solve_revolute(Joint<S>& j, JointCtx& c, S dt) {   // revolute.rs:92-183
        const S PI = S(3.14159265358979323846264338327950288);
        {
            V3<S> a1 = qrot(c.body1->delta_rotation, j.v0), a2 = qrot(c.body2->delta_rotation, j.v1);
            j.total_rot0 = j.total_rot0 + align_orientation(c, cross(a1, a2), S(0), j.compliance1, dt);
        }
        if (j.limit_flags & AVN_JOINT_HAS_LIMIT1) {
            V3<S> a1 = qrot(c.body1->delta_rotation, j.v0), b1 = qrot(c.body1->delta_rotation, j.v2), b2 = qrot(c.body2->delta_rotation, j.v3);
            V3<S> correction;
            if (angle_limit_correction(j.limit_min, j.limit_max, a1, b1, b2, PI, correction))
                j.total_rot1 = j.total_rot1 + align_orientation(c, correction, S(0), j.compliance2, dt);
        }
        point_constraint_solve(j, c, j.compliance, dt);
    }
    void solve_spherical(Joint<S>& j, JointCtx& c, S dt) {  // spherical.rs:85-209
        const S PI = S(3.14159265358979323846264338327950288), EPS = std::numeric_limits<S>::epsilon();
        point_constraint_solve(j, c, j.compliance, dt);
        if (j.limit_flags & AVN_JOINT_HAS_LIMIT1) {             // apply_swing_limits
            V3<S> a1 = qrot(c.body1->delta_rotation, j.v0), a2 = qrot(c.body2->delta_rotation, j.v1);
            V3<S> n = cross(a1, a2);
            S n_magnitude = length(n);
            if (!(n_magnitude <= EPS)) {
                n = n / n_magnitude;
                V3<S> correction;
                if (angle_limit_correction(j.limit_min, j.limit_max, n, a1, a2, PI, correction))
                    j.total_rot0 = j.total_rot0 + align_orientation(c, correction, S(0), j.compliance1, dt);
            }
        }
        if (j.limit_flags & AVN_JOINT_HAS_LIMIT2) {             // apply_twist_limits
            V3<S> a1 = qrot(c.body1->delta_rotation, j.v0), a2 = qrot(c.body2->delta_rotation, j.v1);
            V3<S> n = a1 + a2;
            S n_magnitude = length(n);
            if (n_magnitude <= EPS) return;
            V3<S> b1 = qrot(c.body1->delta_rotation, j.v2), b2 = qrot(c.body2->delta_rotation, j.v3);
            n = n / n_magnitude;
            V3<S> n1 = b1 - dot(n, b1) * n, n2 = b2 - dot(n, b2) * n;
            S n1_magnitude = length(n1), n2_magnitude = length(n2);
            if (n1_magnitude <= EPS || n2_magnitude <= EPS) return;
            n1 = n1 / n1_magnitude; n2 = n2 / n2_magnitude;
            S max_correction = dot(a1, a2) > S(-0.5) ? S(2) * PI : dt;
            V3<S> correction;
            if (angle_limit_correction(j.limit2_min, j.limit2_max, n, n1, n2, max_correction, correction))
                j.total_rot1 = j.total_rot1 + align_orientation(c, correction, S(0), j.compliance2, dt);
        }
    }
    void solve_prismatic(Joint<S>& j, JointCtx& c, S dt) {  // prismatic.rs:83-192
        fixed_angle_solve(j, c, j.compliance1, dt);
        V3<S> world_r1 = qrot(c.body1->delta_rotation, j.world_r1);
        V3<S> world_r2 = qrot(c.body2->delta_rotation, j.world_r2);
        V3<S> delta_x = vzero<S>();
        V3<S> axis1 = qrot(c.body1->delta_rotation, j.v0);
        V3<S> separation = ((c.body2->delta_position - c.body1->delta_position) + (world_r2 - world_r1)) + j.center_difference;
        if (j.limit_flags & AVN_JOINT_HAS_LIMIT1) delta_x = delta_x + correction_along_axis(j.limit_min, j.limit_max, separation, axis1);
        V3<S> axis2 = any_orthogonal_vector(axis1);
        V3<S> axis3 = cross(axis1, axis2);
        delta_x = delta_x + correction_along_axis(S(0), S(0), separation, axis2);   // DistanceLimit::ZERO
        delta_x = delta_x + correction_along_axis(S(0), S(0), separation, axis3);
        S magnitude = length(delta_x);
        if (magnitude <= std::numeric_limits<S>::epsilon()) return;
        V3<S> dir = delta_x / magnitude;
        S w1 = positional_w(max_element(c.inv_mass1), c.ii1, world_r1, dir);
        S w2 = positional_w(max_element(c.inv_mass2), c.ii2, world_r2, dir);
        S delta_lagrange = compute_lagrange_update<S>(S(0), magnitude, w1, w2, j.compliance, dt);
        V3<S> impulse = delta_lagrange * dir;
        j.total_lagrange = j.total_lagrange + impulse;
        apply_positional_impulse(c, impulse, world_r1, world_r2);
    }
    void solve_distance(Joint<S>& j, JointCtx& c, S dt) {   // distance.rs:61-117
        V3<S> world_r1 = qrot(c.body1->delta_rotation, j.world_r1);
        V3<S> world_r2 = qrot(c.body2->delta_rotation, j.world_r2);
        V3<S> separation = ((c.body2->delta_position - c.body1->delta_position) + (world_r2 - world_r1)) + j.center_difference;
        // DistanceLimit::compute_correction, dynamics/joints/mod.rs:321-340
        V3<S> dir = vzero<S>(); S distance = 0;
        S dsq = length_squared(separation);
        if (!(dsq <= std::numeric_limits<S>::epsilon())) {
            S d = std::sqrt(dsq);
            if (d < j.limit_min) { dir = separation / d; distance = j.limit_min - d; }
            else if (d > j.limit_max) { dir = (-separation) / d; distance = d - j.limit_max; }
        }
        if (distance <= std::numeric_limits<S>::epsilon()) return;
        S w1 = positional_w(max_element(c.inv_mass1), c.ii1, world_r1, dir);
        S w2 = positional_w(max_element(c.inv_mass2), c.ii2, world_r2, dir);
        S delta_lagrange = compute_lagrange_update<S>(S(0), distance, w1, w2, j.compliance, dt);
        V3<S> impulse = delta_lagrange * dir;
        j.total_lagrange = j.total_lagrange + impulse;
        apply_positional_impulse(c, impulse, world_r1, world_r2);
    }
    // xpbd/plugin.rs:61-76 (snapshot) + :145-189 solve_xpbd_joint<T> for T in the order of :77-82
    void xpbd_solve() {
        for (Body<S>& b : bodies) {
            if (!b.has_solver_body) continue;
            b.pre_solve_delta_position = b.sb.delta_position;
            b.pre_solve_delta_rotation = b.sb.delta_rotation;
        }
        S dt = h_adj;
        static const SolverBodyInertia<S> DUMMY;
        int cur_type = -1;
        for (uint32_t idx : joint_order) {
            Joint<S>& j = joints[idx];
            if ((int)j.type != cur_type) { cur_type = j.type; xpbd_dummy[0] = SolverBody<S>(); xpbd_dummy[1] = SolverBody<S>(); }  // one system per type
            Body<S>& B1 = bodies[j.body1]; Body<S>& B2 = bodies[j.body2];
            SolverBody<S>* body1 = &xpbd_dummy[0]; SolverBody<S>* body2 = &xpbd_dummy[1];
            const SolverBodyInertia<S>* inertia1 = &DUMMY; const SolverBodyInertia<S>* inertia2 = &DUMMY;
            if (B1.has_solver_body) { body1 = &B1.sb; inertia1 = &B1.si; }
            if (B2.has_solver_body) { body2 = &B2.sb; inertia2 = &B2.si; }
            int rel = (int)inertia1->dominance - (int)inertia2->dominance;
            if (rel > 0) inertia1 = &DUMMY; else if (rel < 0) inertia2 = &DUMMY;
            JointCtx c{body1, body2, inertia1->effective_inv_mass(), inertia2->effective_inv_mass(), inertia1->inv_inertia, inertia2->inv_inertia};
            switch (j.type) {
                case AVN_JOINT_FIXED: solve_fixed(j, c, dt); break;
                case AVN_JOINT_REVOLUTE: solve_revolute(j, c, dt); break;
                case AVN_JOINT_SPHERICAL: solve_spherical(j, c, dt); break;
                case AVN_JOINT_PRISMATIC: solve_prismatic(j, c, dt); break;
                default: solve_distance(j, c, dt); break;
            }
        }
    }
    // xpbd/plugin.rs:192-240 (RigidBodyActiveFilter; runs over ALL active solver bodies)
    void xpbd_velocity_projection() {
        S delta_secs = h_adj;
        for (Body<S>& b : bodies) {
            if (!b.has_solver_body) continue;
            V3<S> new_lin_vel = (b.sb.delta_position - b.pre_solve_delta_position) / delta_secs;
            b.sb.linear_velocity = b.sb.linear_velocity + new_lin_vel;
        }
        for (Body<S>& b : bodies) {
            if (!b.has_solver_body) continue;
            Q4<S> delta_rot = qmul(b.sb.delta_rotation, qinverse(b.pre_solve_delta_rotation));
            V3<S> new_ang_vel = (S(2) * V3<S>{delta_rot.x, delta_rot.y, delta_rot.z}) / delta_secs;
            if (delta_rot.w < S(0)) new_ang_vel = -new_ang_vel;
            b.sb.angular_velocity = b.sb.angular_velocity + new_ang_vel;
        }
    }
    // solver/plugin.rs:759-806 (dummy bodies shared across the loop, no dominance swap here)
    void joint_damping() {
        S delta_secs = h_adj;
        static const SolverBodyInertia<S> DUMMY;
        SolverBody<S> d1, d2;
        int cur_type = -1;
        for (uint32_t idx : joint_order) {   // joint_damping::<T> for T in type order (solver/plugin.rs:139-150), fresh DUMMYs per system
            Joint<S>& j = joints[idx];
            if ((int)j.type != cur_type) { cur_type = j.type; d1 = SolverBody<S>(); d2 = SolverBody<S>(); }
            if (!j.has_damping) continue;
            Body<S>& B1 = bodies[j.body1]; Body<S>& B2 = bodies[j.body2];
            SolverBody<S>* body1 = &d1; SolverBody<S>* body2 = &d2;
            const SolverBodyInertia<S>* inertia1 = &DUMMY; const SolverBodyInertia<S>* inertia2 = &DUMMY;
            if (B1.has_solver_body) { body1 = &B1.sb; inertia1 = &B1.si; }
            if (B2.has_solver_body) { body2 = &B2.sb; inertia2 = &B2.si; }
            V3<S> delta_omega = (body2->angular_velocity - body1->angular_velocity) * smin(j.damping_angular * delta_secs, S(1));
            if (!body1->is_kinematic()) body1->angular_velocity = body1->angular_velocity + delta_omega;
            if (!body2->is_kinematic()) body2->angular_velocity = body2->angular_velocity - delta_omega;
            V3<S> delta_v = (body2->linear_velocity - body1->linear_velocity) * smin(j.damping_linear * delta_secs, S(1));
            V3<S> w1 = inertia1->effective_inv_mass(), w2 = inertia2->effective_inv_mass();
            V3<S> p = cmul(delta_v, recip_or_zero(w1 + w2));
            body1->linear_velocity = body1->linear_velocity + cmul(p, w1);
            body2->linear_velocity = body2->linear_velocity - cmul(p, w2);
        }
    }
    // xpbd/plugin.rs:242-260
    void writeback_joint_forces() {
        S delta_secs = dt_adj;  // Time is Time<Physics> again after the substep loop (solver/schedule.rs:209-212)
        S rhs = recip_or_zero(delta_secs * delta_secs) * (S)cfg.substeps;
        for (Joint<S>& j : joints) { j.force = j.total_lagrange * rhs; j.torque = (j.total_rot0 + j.total_rot1) * rhs; }
    }

    // =============================================================================================
    //                                      BROAD PHASE
    // =============================================================================================
    // child colliders (header: "child colliders"): ColliderTransform of the colliders that are not on their body's entity; a colliders_upload puts every collider back on its body
    avn_status collider_transforms_upload(const avn_collider_transforms* t) override {
        if (!t || !t->count) { for (Collider<S>& c : colliders) c.child = false; return AVN_OK; }
        if (t->count != colliders.size()) { error = "collider_transforms_upload: count differs from the last colliders_upload"; return AVN_ERR_BAD_ARG; }
        if (!t->is_child || !t->translation || !t->rotation) { error = "collider_transforms_upload: null array"; return AVN_ERR_BAD_ARG; }
        for (uint32_t i = 0; i < t->count; ++i) {
            Collider<S>& c = colliders[i];
            c.child = t->is_child[i] != 0;
            c.local_translation = rd3(t->translation, i);
            const S* r = (const S*)t->rotation + 4 * (size_t)i;
            c.local_rotation = {r[0], r[1], r[2], r[3]};
        }
        return AVN_OK;
    }
    // update_child_collider_position (collision/collider/collider_transform/plugin.rs:62-91): Position / Rotation of a collider; its body's for a collider on the body's entity
    static void collider_pose(const Collider<S>& c, const Body<S>& b, V3<S>& pos, Q4<S>& rot) {
        pos = b.position; rot = b.rotation;
        if (!c.child) return;
        pos = b.position + qrot(b.rotation, c.local_translation);
        rot = qnormalize(qmul(b.rotation, c.local_rotation));
    }
    avn_status colliders_upload(const avn_colliders* c) override {
        if (!c || (c->count && (!c->entity_index || !c->body || !c->shape || !c->half_extents))) { error = "colliders_upload: null array"; return AVN_ERR_BAD_ARG; }
        std::vector<Collider<S>> next(c->count);
        std::unordered_map<uint32_t, uint32_t> next_slot;
        uint32_t n_host_here = 0;
        for (uint32_t i = 0; i < c->count; ++i) {
            Collider<S>& o = next[i];
            o.entity = c->entity_index[i];
            o.body = c->body[i];
            if (o.body < 0 || (size_t)o.body >= bodies.size()) { error = "colliders_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
            o.shape = c->shape[i];
            n_host_here += o.shape == AVN_SHAPE_HOST;
            o.half_extents = rd3(c->half_extents, i);
            o.memberships = rd<uint32_t>(c->memberships, i, 1u);
            o.filters = rd<uint32_t>(c->filters, i, 0xFFFFFFFFu);
            o.cflags = rd<uint8_t>(c->collider_flags, i, 0);
            o.collision_margin = rd<S>(c->collision_margin, i, 0);
            o.speculative_margin = rd<S>(c->speculative_margin, i, -1);
            auto it = collider_slot.find(o.entity);
            if (it != collider_slot.end()) o.aabb = colliders[it->second].aabb; else o.aabb = {{0, 0, 0}, {0, 0, 0}};
            if (!next_slot.emplace(o.entity, i).second) { error = "colliders_upload: duplicate entity_index"; return AVN_ERR_BAD_ARG; }
        }
        // retain_mut (broad_phase.rs:230-279): drop intervals of colliders that no longer exist, in place
        std::vector<AabbInterval> kept;
        kept.reserve(intervals.size());
        std::unordered_set<uint32_t> known;
        for (const AabbInterval& iv : intervals) {
            uint32_t ent = colliders[iv.collider].entity;
            auto it = next_slot.find(ent);
            if (it == next_slot.end()) continue;
            kept.push_back({it->second, iv.flags});
            known.insert(ent);
        }
        // add_new_aabb_intervals (broad_phase.rs:296-315): append the new ones at the END, in upload order
        for (uint32_t i = 0; i < c->count; ++i)
            if (!known.count(next[i].entity)) kept.push_back({i, 0});
        colliders.swap(next);
        collider_slot.swap(next_slot);
        intervals.swap(kept);
        n_host_colliders = n_host_here;
        have_colliders = true;
        despawn_needs_colliders = false;
        if (slp)   // colliders spawned inside the loop join their body's RigidBodyColliders (upload order = Add order)
            for (const Collider<S>& o : colliders) {
                if (slp->isl.has_collider(o.entity)) continue;
                const Body<S>& b = bodies[(size_t)o.body];
                const bool node = b.rb_type != AVN_RB_STATIC && !(b.body_flags & AVN_BODY_DISABLED);
                slp->isl.collider_add(o.entity, node ? (uint32_t)o.body : IslandManager::NONE);
            }
        return AVN_OK;
    }
    avn_status existing_pairs_upload(const uint64_t* keys, size_t n) override {
        if (n && !keys) return AVN_ERR_BAD_ARG;
        pair_set.clear();
        pair_set.insert(keys, keys + n);
        return AVN_OK;
    }
    avn_status pairs_get(const avn_pair** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        *out = pairs.data(); *n = pairs.size();
        return AVN_OK;
    }
    // header avn_dynamic_bounds: union of the ColliderAabbs of colliders on non-static bodies
    avn_status dynamic_bounds(double* mn, double* mx) override {
        if (!mn || !mx) return AVN_ERR_BAD_ARG;
        const double inf = std::numeric_limits<double>::infinity();
        for (int k = 0; k < 3; ++k) { mn[k] = inf; mx[k] = -inf; }
        for (const Collider<S>& c : colliders) {
            if (bodies[c.body].rb_type == AVN_RB_STATIC) continue;
            const double lo[3] = {(double)c.aabb.min.x, (double)c.aabb.min.y, (double)c.aabb.min.z};
            const double hi[3] = {(double)c.aabb.max.x, (double)c.aabb.max.y, (double)c.aabb.max.z};
            for (int k = 0; k < 3; ++k) { if (lo[k] < mn[k]) mn[k] = lo[k]; if (hi[k] > mx[k]) mx[k] = hi[k]; }
        }
        return AVN_OK;
    }
    // contact_query::contact_manifolds for a batch of pairs (avo_narrow.hpp)
    avn_status contact_manifolds(const avn_shape_pairs* p, const avn_query_manifolds_out* o) override {
        if (!p || !o || (p->count && (!p->shape1 || !p->shape2 || !p->half_extents1 || !p->half_extents2 || !p->position1 || !p->position2 ||
                                      !p->rotation1 || !p->rotation2 || !p->prediction_distance))) { error = "contact_manifolds: null array"; return AVN_ERR_BAD_ARG; }
        auto rdq = [](const void* a, size_t i) { const S* q = (const S*)a + 4 * i; return Q4<S>{q[0], q[1], q[2], q[3]}; };
        for (size_t i = 0; i < p->count; ++i) {
            if (p->shape1[i] > AVN_SHAPE_BALL || p->shape2[i] > AVN_SHAPE_BALL) { error = "contact_manifolds: unknown shape"; return AVN_ERR_BAD_ARG; }
            QueryManifold<S> m;
            bool has = contact_manifolds_pair<S>(p->shape1[i], rd3(p->half_extents1, i), rd3(p->position1, i), rdq(p->rotation1, i), p->shape2[i],
                                                 rd3(p->half_extents2, i), rd3(p->position2, i), rdq(p->rotation2, i), ((const S*)p->prediction_distance)[i], m);
            int n = has ? m.n : 0;
            if (o->point_count) o->point_count[i] = (uint8_t)n;
            wr3(o->normal, i, has ? m.normal : vzero<S>());
            for (int k = 0; k < AVN_MAX_QUERY_POINTS; ++k) {
                size_t s = (size_t)AVN_MAX_QUERY_POINTS * i + k;
                bool live = k < n;
                wr3(o->anchor1, s, live ? m.pts[k].anchor1 : vzero<S>());
                wr3(o->anchor2, s, live ? m.pts[k].anchor2 : vzero<S>());
                wr3(o->point, s, live ? m.pts[k].point : vzero<S>());
                if (o->penetration) ((S*)o->penetration)[s] = live ? m.pts[k].penetration : S(0);
                if (o->feature_id1) o->feature_id1[s] = live ? m.pts[k].fid1 : 0u;
                if (o->feature_id2) o->feature_id2[s] = live ? m.pts[k].fid2 : 0u;
            }
        }
        return AVN_OK;
    }
    // ---- narrow phase, part 2 ------------------------------------------------------------------------------------------
    avn_status collider_materials_upload(const avn_collider_materials* m) override {
        if (!m || m->count != colliders.size()) { error = "collider_materials_upload: count must equal the collider count"; return AVN_ERR_BAD_ARG; }
        materials.resize(m->count);
        for (uint32_t i = 0; i < m->count; ++i)
            materials[i] = {rd<S>(m->friction, i, S(0.5)), rd<S>(m->restitution, i, S(0)), rd<uint8_t>(m->friction_combine, i, AVN_COMBINE_AVERAGE),
                            rd<uint8_t>(m->restitution_combine, i, AVN_COMBINE_AVERAGE)};
        return AVN_OK;
    }
    Material material_of(uint32_t slot) const { return slot < materials.size() ? materials[slot] : Material{S(0.5), S(0), AVN_COMBINE_AVERAGE, AVN_COMBINE_AVERAGE}; }
    // CoefficientCombine::mix (physics_material.rs:28-36) under the higher-priority rule (:205-214, :372-380)
    static S combine(S a, uint8_t ra, S b, uint8_t rb) {
        uint8_t rule = ra > rb ? ra : rb;
        switch (rule) {
            case AVN_COMBINE_GEOMETRIC_MEAN: return std::sqrt(a * b);
            case AVN_COMBINE_MIN: return smin(a, b);
            case AVN_COMBINE_MULTIPLY: return a * b;
            case AVN_COMBINE_MAX: return smax(a, b);
            default: return (a + b) * S(0.5);
        }
    }
    avn_status contact_pairs_add(const avn_contact_pairs* p) override {  // contact_graph.rs:521-566
        if (!p || (p->count && (!p->contact_id || !p->collider1 || !p->collider2 || !p->pair_flags))) { error = "contact_pairs_add: null array"; return AVN_ERR_BAD_ARG; }
        for (uint32_t i = 0; i < p->count; ++i) {
            if (!collider_slot.count(p->collider1[i]) || !collider_slot.count(p->collider2[i])) { error = "contact_pairs_add: unknown collider"; return AVN_ERR_BAD_ARG; }
            uint32_t id = p->contact_id[i];
            if (id >= contact_rows.size()) contact_rows.resize((size_t)id + 1);
            if (contact_rows[id].used) { error = "contact_pairs_add: contact id in use"; return AVN_ERR_STATE; }
            CtRow r;
            r.used = true; r.collider1 = p->collider1[i]; r.collider2 = p->collider2[i];
            uint32_t f = p->pair_flags[i];
            r.flags = ((f & AVN_PAIR_GENERATE_CONSTRAINTS) ? (uint32_t)AVN_CP_GENERATE_CONSTRAINTS : 0u) | ((f & AVN_PAIR_MODIFY_CONTACTS) ? (uint32_t)AVN_CP_MODIFY_CONTACTS : 0u) |
                      ((f & AVN_PAIR_CONTACT_EVENTS) ? (uint32_t)AVN_CP_CONTACT_EVENTS : 0u);
            contact_rows[id] = r;
        }
        return AVN_OK;
    }
    avn_status contact_pairs_remove(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= contact_rows.size() || !contact_rows[ids[i]].used) { error = "contact_pairs_remove: no such contact"; return AVN_ERR_STATE; }
            CtRow& r = contact_rows[ids[i]];
            pair_set.erase(pair_key(r.collider1, r.collider2));
            r = CtRow();
        }
        return AVN_OK;
    }
    avn_status active_pairs_set(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i)
            if (ids[i] >= contact_rows.size() || !contact_rows[ids[i]].used) { error = "active_pairs_set: no such contact"; return AVN_ERR_STATE; }
        active_pairs.assign(ids, ids + n);
        return AVN_OK;
    }
    avn_status contact_changes_get(const avn_contact_change** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        *out = contact_changes.data(); *n = contact_changes.size();
        return AVN_OK;
    }
    avn_status manifold_handles_upload(const uint32_t* offsets, const uint32_t* ids) override {
        if (!offsets || offsets[0] != 0) { error = "manifold_handles_upload: bad offsets"; return AVN_ERR_BAD_ARG; }
        uint32_t M = offsets[AVN_GRAPH_COLOR_COUNT];
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) if (offsets[c] > offsets[c + 1]) { error = "manifold_handles_upload: offsets not monotone"; return AVN_ERR_BAD_ARG; }
        if (M && !ids) return AVN_ERR_BAD_ARG;
        for (uint32_t i = 0; i < M; ++i)
            if (ids[i] >= contact_rows.size() || !contact_rows[ids[i]].used) { error = "manifold_handles_upload: no such contact"; return AVN_ERR_STATE; }
        manifold_handles.assign(ids, ids + M);
        std::memcpy(color_offsets, offsets, sizeof color_offsets);
        use_handles = true;
        return AVN_OK;
    }
    avn_status contacts_download(const uint32_t* ids, size_t n, const avn_contacts_out* o) override {
        if (!o || (n && !ids)) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= contact_rows.size() || !contact_rows[ids[i]].used) { error = "contacts_download: no such contact"; return AVN_ERR_STATE; }
            const CtRow& r = contact_rows[ids[i]];
            int np = r.n_manifolds ? r.point_count : 0;
            if (o->flags) o->flags[i] = r.flags;
            if (o->point_count) o->point_count[i] = (uint8_t)np;
            wr3(o->normal, i, np ? r.normal : vzero<S>());
            if (o->friction) ((S*)o->friction)[i] = np ? r.friction : S(0);
            if (o->restitution) ((S*)o->restitution)[i] = np ? r.restitution : S(0);
            for (int k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                size_t s = 4 * i + k;
                bool live = k < np;
                const CtPoint& c = r.pts[k];
                wr3(o->anchor1, s, live ? c.anchor1 : vzero<S>()); wr3(o->anchor2, s, live ? c.anchor2 : vzero<S>());
                if (o->penetration) ((S*)o->penetration)[s] = live ? c.penetration : S(0);
                if (o->normal_speed) ((S*)o->normal_speed)[s] = live ? c.normal_speed : S(0);
                if (o->warm_start_normal_impulse) ((S*)o->warm_start_normal_impulse)[s] = live ? c.warm_start_normal_impulse : S(0);
                if (o->warm_start_tangent_impulse) { ((S*)o->warm_start_tangent_impulse)[2 * s] = live ? c.warm_start_tangent_impulse.x : S(0); ((S*)o->warm_start_tangent_impulse)[2 * s + 1] = live ? c.warm_start_tangent_impulse.y : S(0); }
                if (o->normal_impulse) ((S*)o->normal_impulse)[s] = live ? c.normal_impulse : S(0);
                if (o->feature_id1) o->feature_id1[s] = live ? c.feature_id1 : 0u;
                if (o->feature_id2) o->feature_id2[s] = live ? c.feature_id2 : 0u;
            }
        }
        return AVN_OK;
    }
    // the inverse of contacts_download: a ContactPair that moved here with its manifold (what NarrowPhase::update reads of the
    // previous step: flags for the started/stopped events, system_param.rs:560-584; the points for match_contacts,
    // contact_types/mod.rs:568-600)
    avn_status contacts_upload(const uint32_t* ids, size_t n, const avn_contacts_in* in) override {
        if (!in || (n && !ids)) return AVN_ERR_BAD_ARG;
        if (n && (!in->flags || !in->point_count || !in->normal || !in->friction || !in->restitution || !in->anchor1 || !in->anchor2 || !in->penetration || !in->normal_speed ||
                  !in->warm_start_normal_impulse || !in->warm_start_tangent_impulse || !in->normal_impulse || !in->feature_id1 || !in->feature_id2)) {
            error = "contacts_upload: every field of avn_contacts_in is required"; return AVN_ERR_BAD_ARG;
        }
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= contact_rows.size() || !contact_rows[ids[i]].used) { error = "contacts_upload: no such contact (avn_contact_pairs_add first)"; return AVN_ERR_STATE; }
            if (in->point_count[i] > AVN_MAX_MANIFOLD_POINTS) { error = "contacts_upload: point_count > 4"; return AVN_ERR_BAD_ARG; }
        }
        for (size_t i = 0; i < n; ++i) {
            CtRow& r = contact_rows[ids[i]];
            r.flags = in->flags[i];
            r.point_count = in->point_count[i];
            r.n_manifolds = r.point_count ? 1u : 0u;
            r.normal = rd3(in->normal, i);
            r.friction = ((const S*)in->friction)[i];
            r.restitution = ((const S*)in->restitution)[i];
            for (int k = 0; k < r.point_count; ++k) {
                size_t s = 4 * i + k;
                CtPoint& c = r.pts[k];
                c.anchor1 = rd3(in->anchor1, s); c.anchor2 = rd3(in->anchor2, s);
                c.penetration = ((const S*)in->penetration)[s];
                c.normal_speed = ((const S*)in->normal_speed)[s];
                c.warm_start_normal_impulse = ((const S*)in->warm_start_normal_impulse)[s];
                c.warm_start_tangent_impulse = {((const S*)in->warm_start_tangent_impulse)[2 * s], ((const S*)in->warm_start_tangent_impulse)[2 * s + 1]};
                c.normal_impulse = ((const S*)in->normal_impulse)[s];
                c.feature_id1 = in->feature_id1[s]; c.feature_id2 = in->feature_id2[s];
            }
        }
        return AVN_OK;
    }
    // ContactManifold::prune_points (contact_types/mod.rs:477-566) on the index set `idx` (n > 4): returns the kept order
    static int prune_points(const CtPoint* pts, V3<S> normal, int n, int* keep) {
        const S MIN_DISTANCE_SQUARED = S(1e-6);
        V3<S> projected[AVO_MAX_RAW_POINTS];
        S pen_sq[AVO_MAX_RAW_POINTS];
        for (int i = 0; i < n; ++i) {
            projected[i] = pts[i].anchor1 - normal * dot(pts[i].anchor1, normal);  // reject_from_normalized
            pen_sq[i] = smax(pts[i].penetration * pts[i].penetration, MIN_DISTANCE_SQUARED);
        }
        int p1 = 0;
        S value = -std::numeric_limits<S>::max();  // Scalar::MIN
        for (int i = 0; i < n; ++i) {
            S v = smax(length_squared(projected[i]), MIN_DISTANCE_SQUARED) * pen_sq[i];
            if (v > value) { value = v; p1 = i; }
        }
        int p2 = -1;
        S max_distance = -std::numeric_limits<S>::max();
        for (int i = 0; i < n; ++i) {
            if (i == p1) continue;
            S v = smax(length_squared(projected[i] - projected[p1]), MIN_DISTANCE_SQUARED) * pen_sq[i];
            if (v > max_distance) { max_distance = v; p2 = i; }
        }
        int p3 = -1, p4 = -1;
        S min_value = S(0), max_value = S(0);
        V3<S> perp = cross(projected[p2] - projected[p1], normal);
        for (int i = 0; i < n; ++i) {
            if (i == p1 || i == p2) continue;
            S v = dot(perp, projected[i] - projected[p1]);
            if (v < min_value) { min_value = v; p3 = i; }
            else if (v > max_value) { max_value = v; p4 = i; }
        }
        int k = 0;
        keep[k++] = p1;
        if (p3 >= 0) keep[k++] = p3;
        keep[k++] = p2;
        if (p4 >= 0) keep[k++] = p4;
        return k;
    }
    // NarrowPhase::update_contacts, narrow_phase/system_param.rs:437-830 (hooks are not called; no islands / events here)
    void narrow_phase() {
        contact_changes.clear();
        const S delta_secs = dt_adj;
        const S default_speculative_margin = (S)cfg.length_unit * (cfg.default_speculative_margin >= (double)std::numeric_limits<S>::max() ? std::numeric_limits<S>::max() : (S)cfg.default_speculative_margin);
        const S contact_tolerance = (S)cfg.length_unit * (S)cfg.contact_tolerance;
        // one contact pair; pairs own disjoint rows, so the reference runs them on the ComputeTaskPool with thread-local
        // status-change bit vectors (system_param.rs:454-475, feature "parallel"); here: per-chunk change lists
        auto update_pair = [&](uint32_t id, std::vector<avn_contact_change>& changes) {
            CtRow& r = contact_rows[id];
            bool status = false;
            auto i1 = collider_slot.find(r.collider1), i2 = collider_slot.find(r.collider2);
            if (i1 == collider_slot.end() || i2 == collider_slot.end()) return;  // collider_query.get_many failed
            const Collider<S>& c1 = colliders[i1->second]; const Collider<S>& c2 = colliders[i2->second];
            bool overlap = c1.aabb.min.x <= c2.aabb.max.x && c1.aabb.max.x >= c2.aabb.min.x && c1.aabb.min.y <= c2.aabb.max.y && c1.aabb.max.y >= c2.aabb.min.y &&
                           c1.aabb.min.z <= c2.aabb.max.z && c1.aabb.max.z >= c2.aabb.min.z;  // ColliderAabb::intersects, collider/mod.rs:539-544
            bool interacts = (c1.memberships & c2.filters) != 0 && (c2.memberships & c1.filters) != 0;
            if (!overlap || !interacts) {
                r.flags |= AVN_CP_DISJOINT_AABB;
                status = true;
            } else {
                const Body<S>& b1 = bodies[c1.body]; const Body<S>& b2 = bodies[c2.body];
                const bool have1 = !(b1.body_flags & AVN_BODY_DISABLED), have2 = !(b2.body_flags & AVN_BODY_DISABLED);  // Without<RigidBodyDisabled>
                // the collider sits on the body entity: collider.position = body.position
                bool is_static1 = have1 && b1.rb_type == AVN_RB_STATIC, is_static2 = have2 && b2.rb_type == AVN_RB_STATIC;
                V3<S> cpos1, cpos2; Q4<S> crot1, crot2;   // the colliders' own Position / Rotation (a child collider: update_child_collider_position)
                collider_pose(c1, b1, cpos1, crot1); collider_pose(c2, b2, cpos2, crot2);
                V3<S> collider_offset1 = have1 ? cpos1 - b1.position : vzero<S>(), collider_offset2 = have2 ? cpos2 - b2.position : vzero<S>();
                V3<S> world_com1 = have1 ? qrot(b1.rotation, b1.center_of_mass) : vzero<S>(), world_com2 = have2 ? qrot(b2.rotation, b2.center_of_mass) : vzero<S>();
                V3<S> lin_vel1 = have1 ? b1.linear_velocity : vzero<S>(), lin_vel2 = have2 ? b2.linear_velocity : vzero<S>();
                V3<S> ang_vel1 = have1 ? b1.angular_velocity : vzero<S>(), ang_vel2 = have2 ? b2.angular_velocity : vzero<S>();
                r.flags = (r.flags & ~(uint32_t)(AVN_CP_STATIC1 | AVN_CP_STATIC2)) | (is_static1 ? (uint32_t)AVN_CP_STATIC1 : 0u) | (is_static2 ? (uint32_t)AVN_CP_STATIC2 : 0u);
                bool is_disabled = !have1 || !have2 || (c1.cflags & AVN_COLLIDER_SENSOR) || (c2.cflags & AVN_COLLIDER_SENSOR);
                if (!is_disabled && !(r.flags & AVN_CP_GENERATE_CONSTRAINTS)) { r.flags |= AVN_CP_STARTED_GENERATING_CONSTRAINTS; status = true; }
                r.flags = is_disabled ? (r.flags & ~(uint32_t)AVN_CP_GENERATE_CONSTRAINTS) : (r.flags | AVN_CP_GENERATE_CONSTRAINTS);
                Material m1 = material_of(i1->second), m2 = material_of(i2->second);
                S friction = combine(m1.friction, m1.friction_combine, m2.friction, m2.friction_combine);
                S restitution = combine(m1.restitution, m1.restitution_combine, m2.restitution, m2.restitution_combine);
                S collision_margin_sum = c1.collision_margin + c2.collision_margin;
                S speculative_margin1 = c1.speculative_margin >= S(0) ? c1.speculative_margin : default_speculative_margin;
                S speculative_margin2 = c2.speculative_margin >= S(0) ? c2.speculative_margin : default_speculative_margin;
                S inv_delta_secs = S(1) / delta_secs;
                if (speculative_margin1 < std::numeric_limits<S>::max()) lin_vel1 = clamp_length_max(lin_vel1, speculative_margin1 * inv_delta_secs);
                if (speculative_margin2 < std::numeric_limits<S>::max()) lin_vel2 = clamp_length_max(lin_vel2, speculative_margin2 * inv_delta_secs);
                V3<S> relative_linear_velocity = lin_vel2 - lin_vel1;
                S effective_speculative_margin = delta_secs * length(relative_linear_velocity);
                S max_contact_distance = smax(effective_speculative_margin, contact_tolerance) + collision_margin_sum;
                bool was_touching = r.flags & AVN_CP_TOUCHING;
                CtRow old = r;  // old_manifolds = contacts.manifolds.clone()
                QueryManifold<S> qm;
                bool has;
                if (c1.shape == AVN_SHAPE_HOST || c2.shape == AVN_SHAPE_HOST)   // AnyCollider::contact_manifolds_with_context on the host (system_param.rs:700-712; header: "host shapes")
                    has = host_contact_manifold(id, c1, cpos1, crot1, c2, cpos2, crot2, max_contact_distance, qm);
                else
                has = contact_manifolds_pair<S>(c1.shape, c1.half_extents, cpos1, crot1, c2.shape, c2.half_extents, cpos2, crot2, max_contact_distance, qm);
                // retain_mut over the (at most one) manifold
                CtPoint kept[AVO_MAX_RAW_POINTS];
                int nk = 0;
                if (has) {
                    for (int k = 0; k < qm.n; ++k) {
                        CtPoint pt;
                        pt.anchor1 = (qm.pts[k].anchor1 + collider_offset1) - world_com1;
                        pt.anchor2 = (qm.pts[k].anchor2 + collider_offset2) - world_com2;
                        pt.penetration = qm.pts[k].penetration + collision_margin_sum;
                        V3<S> relative_velocity = (relative_linear_velocity + cross(ang_vel2, pt.anchor2)) - cross(ang_vel1, pt.anchor1);
                        pt.normal_speed = dot(relative_velocity, qm.normal);
                        pt.warm_start_normal_impulse = 0; pt.normal_impulse = 0; pt.warm_start_tangent_impulse = {0, 0};
                        pt.feature_id1 = qm.pts[k].fid1; pt.feature_id2 = qm.pts[k].fid2;
                        bool keep = -pt.penetration < effective_speculative_margin || (pt.normal_speed * delta_secs - pt.penetration < effective_speculative_margin);
                        if (keep) kept[nk++] = pt;
                    }
                }
                r.n_manifolds = 0; r.point_count = 0;
                if (nk > 0) {
                    r.normal = qm.normal; r.friction = friction; r.restitution = restitution;
                    if (nk > 4) {
                        int keep[4];
                        int k4 = prune_points(kept, qm.normal, nk, keep);
                        for (int k = 0; k < k4; ++k) r.pts[k] = kept[keep[k]];
                        r.point_count = k4;
                    } else {
                        for (int k = 0; k < nk; ++k) r.pts[k] = kept[k];
                        r.point_count = nk;
                    }
                    r.n_manifolds = 1;
                    r.tangent_velocity = vzero<S>();   // system_param.rs:722-729
                }
                bool touching = r.n_manifolds != 0;
                // CollisionHooks::modify_contacts (system_param.rs:770-778; header: "collision hooks")
                if (touching && (r.flags & AVN_CP_MODIFY_CONTACTS) && hk_modify_fn) touching = hook_modify_contacts(id, r, c1, c2);
                r.flags = touching ? (r.flags | AVN_CP_TOUCHING) : (r.flags & ~(uint32_t)AVN_CP_TOUCHING);
                if (r.n_manifolds <= 4 && cfg.match_contacts && touching && old.n_manifolds) {
                    // ContactManifold::match_contacts (contact_types/mod.rs:425-475)
                    S thr = S(0.1) * (S)cfg.length_unit;
                    S thr2 = thr * thr;  // distance_threshold.powi(2)
                    for (int k = 0; k < r.point_count; ++k) {
                        CtPoint& c = r.pts[k];
                        for (int j = 0; j < old.point_count; ++j) {
                            const CtPoint& pc = old.pts[j];
                            if ((c.feature_id1 == pc.feature_id1 && c.feature_id2 == pc.feature_id2) || (c.feature_id2 == pc.feature_id1 && c.feature_id1 == pc.feature_id2)) {
                                c.warm_start_normal_impulse = pc.warm_start_normal_impulse; c.warm_start_tangent_impulse = pc.warm_start_tangent_impulse;
                                break;
                            }
                            bool unknown = c.feature_id1 == 0u || c.feature_id2 == 0u;
                            if ((unknown && (length_squared(c.anchor1 - pc.anchor1) < thr2 && length_squared(c.anchor2 - pc.anchor2) < thr2)) ||
                                (length_squared(c.anchor1 - pc.anchor2) < thr2 && length_squared(c.anchor2 - pc.anchor1) < thr2)) {
                                c.warm_start_normal_impulse = pc.warm_start_normal_impulse; c.warm_start_tangent_impulse = pc.warm_start_tangent_impulse;
                                break;
                            }
                        }
                    }
                }
                r.manifold_count_change = (int32_t)r.n_manifolds - (int32_t)old.n_manifolds;
                if (touching && !was_touching) { r.flags |= AVN_CP_STARTED_TOUCHING; status = true; }
                else if (!touching && was_touching) { r.flags |= AVN_CP_STOPPED_TOUCHING; status = true; }
                else if (r.manifold_count_change != 0) status = true;
            }
            if (status) changes.push_back({id, r.flags, r.manifold_count_change, r.n_manifolds});
        };
        {
            const size_t n = active_pairs.size(), chunk = std::max<size_t>(n / pool.threads, 1), chunks = n ? (n + chunk - 1) / chunk : 0;
            std::vector<std::vector<avn_contact_change>> per_chunk(pool.threads == 1 || n < 64 ? 1 : chunks);
            // (pairs with a host-shaped collider call back into the host: on the calling thread, in ascending contact id, after the pool's share)
            auto host_pair = [&](uint32_t id) {
                if (hk_modify_fn && (contact_rows[id].flags & AVN_CP_MODIFY_CONTACTS)) return true;   // (a pair whose hook may call back: with the host pairs, on this thread)
                if (n_host_colliders == 0) return false;
                const CtRow& r = contact_rows[id];
                auto i1 = collider_slot.find(r.collider1), i2 = collider_slot.find(r.collider2);
                return i1 != collider_slot.end() && i2 != collider_slot.end() && (colliders[i1->second].shape == AVN_SHAPE_HOST || colliders[i2->second].shape == AVN_SHAPE_HOST);
            };
            if (per_chunk.size() == 1) { for (uint32_t id : active_pairs) if (!host_pair(id)) update_pair(id, per_chunk[0]); }
            else pool.par_for_each(n, 64, [&](size_t b0, size_t b1) { std::vector<avn_contact_change>& out = per_chunk[b0 / chunk]; for (size_t i = b0; i < b1; ++i) if (!host_pair(active_pairs[i])) update_pair(active_pairs[i], out); });
            hk_stats.last_modify_queries = hk_stats.last_modify_rejected = 0;
            if (n_host_colliders || hk_modify_fn) {
                std::vector<uint32_t> hp;
                for (uint32_t id : active_pairs) if (host_pair(id)) hp.push_back(id);
                std::sort(hp.begin(), hp.end());
                hs_stats.last_manifold_queries = (uint32_t)hp.size(); hs_stats.last_manifolds_with_points = 0;
                for (uint32_t id : hp) update_pair(id, per_chunk[0]);
            }
            for (const std::vector<avn_contact_change>& v : per_chunk) contact_changes.insert(contact_changes.end(), v.begin(), v.end());
        }
        std::sort(contact_changes.begin(), contact_changes.end(), [](const avn_contact_change& a, const avn_contact_change& b) { return a.contact_id < b.contact_id; });
        // the status processing of system_param.rs:141-389 clears these once handled (host side); here they are per-step outputs
        for (const avn_contact_change& c : contact_changes)
            contact_rows[c.contact_id].flags &= ~(uint32_t)(AVN_CP_STARTED_TOUCHING | AVN_CP_STOPPED_TOUCHING | AVN_CP_STARTED_GENERATING_CONSTRAINTS);
    }
    // ---- collision hooks (header: "collision hooks"): CollisionHooks::filter_pairs / modify_contacts through the callbacks, one record per call ----
    avn_filter_pairs_fn hk_filter_fn = nullptr; avn_modify_contacts_fn hk_modify_fn = nullptr; void* hk_user = nullptr;
    avn_collision_hook_stats hk_stats{};
    template <class Q> struct HookC { uint32_t contact_id, collider1, collider2, body1, body2, flags, touching, manifold_count, point_count, reserved;
                                      Q normal[3], friction, restitution, tangent_velocity[3], anchor1[12], anchor2[12], penetration[4], normal_speed[4]; uint32_t fid1[4], fid2[4]; };
    static_assert(sizeof(HookC<float>) == sizeof(avn_hook_contact_f32) && sizeof(HookC<double>) == sizeof(avn_hook_contact_f64), "hook record layout");
    avn_status collision_hooks_set(avn_filter_pairs_fn f, avn_modify_contacts_fn m, void* user) override { hk_filter_fn = f; hk_modify_fn = m; hk_user = user; return AVN_OK; }
    avn_status collision_hook_stats_get(avn_collision_hook_stats* o) override { if (!o) return AVN_ERR_BAD_ARG; *o = hk_stats; return AVN_OK; }
    // the pair as the hook sees it -> the callback -> the pair as the hook left it; returns the hook's `touching`
    bool hook_modify_contacts(uint32_t id, CtRow& r, const Collider<S>& c1, const Collider<S>& c2) {
        HookC<S> h;
        std::memset(&h, 0, sizeof h);
        h.contact_id = id; h.collider1 = c1.entity; h.collider2 = c2.entity; h.body1 = (uint32_t)c1.body; h.body2 = (uint32_t)c2.body;
        h.flags = r.flags & 0xFFFFu; h.touching = 1u; h.manifold_count = 1u; h.point_count = (uint32_t)r.point_count;
        h.normal[0] = r.normal.x; h.normal[1] = r.normal.y; h.normal[2] = r.normal.z; h.friction = r.friction; h.restitution = r.restitution;
        for (int k = 0; k < r.point_count; ++k) {
            const CtPoint& pt = r.pts[k];
            h.anchor1[3 * k] = pt.anchor1.x; h.anchor1[3 * k + 1] = pt.anchor1.y; h.anchor1[3 * k + 2] = pt.anchor1.z;
            h.anchor2[3 * k] = pt.anchor2.x; h.anchor2[3 * k + 1] = pt.anchor2.y; h.anchor2[3 * k + 2] = pt.anchor2.z;
            h.penetration[k] = pt.penetration; h.normal_speed[k] = pt.normal_speed; h.fid1[k] = pt.feature_id1; h.fid2[k] = pt.feature_id2;
        }
        hk_modify_fn(hk_user, (uint32_t)(8 * sizeof(S)), 1u, &h);
        ++hk_stats.last_modify_queries; hk_stats.bytes_to_host += sizeof h; hk_stats.bytes_from_host += sizeof h;
        const bool touching = h.touching != 0u;
        hk_stats.last_modify_rejected += !touching;
        r.n_manifolds = touching && h.manifold_count ? 1u : 0u;   // !touching: manifolds.clear()
        r.point_count = r.n_manifolds ? (int)std::min<uint32_t>(h.point_count, AVN_MAX_MANIFOLD_POINTS) : 0;
        if (r.n_manifolds) {
            r.normal = {h.normal[0], h.normal[1], h.normal[2]}; r.friction = h.friction; r.restitution = h.restitution;
            r.tangent_velocity = {h.tangent_velocity[0], h.tangent_velocity[1], h.tangent_velocity[2]};
            for (int k = 0; k < r.point_count; ++k) {
                CtPoint pt;
                pt.anchor1 = {h.anchor1[3 * k], h.anchor1[3 * k + 1], h.anchor1[3 * k + 2]}; pt.anchor2 = {h.anchor2[3 * k], h.anchor2[3 * k + 1], h.anchor2[3 * k + 2]};
                pt.penetration = h.penetration[k]; pt.normal_speed = h.normal_speed[k];
                pt.warm_start_normal_impulse = 0; pt.normal_impulse = 0; pt.warm_start_tangent_impulse = {0, 0};   // (ContactPoint::new's zeros: the record does not carry impulses)
                pt.feature_id1 = h.fid1[k]; pt.feature_id2 = h.fid2[k];
                r.pts[k] = pt;
            }
        }
        return touching;
    }
    // ---- host shapes (header: "host shapes"): the two AnyCollider methods through the callbacks, one query per call ----
    avn_host_aabb_fn hs_aabb_fn = nullptr; avn_host_manifolds_fn hs_manifolds_fn = nullptr; void* hs_user = nullptr;
    uint32_t n_host_colliders = 0;
    avn_host_shape_stats hs_stats{};
    template <class Q> struct HostAabbQ { uint32_t collider, swept; Q start_position[3], start_rotation[4], end_position[3], end_rotation[4]; };
    template <class Q> struct HostAabb { Q min[3], max[3]; };
    template <class Q> struct HostMQ { uint32_t contact_id, collider1, collider2, reserved; Q position1[3], rotation1[4], position2[3], rotation2[4], max_contact_distance; };
    template <class Q> struct HostMM { uint32_t point_count; Q normal[3]; Q anchor1[3 * AVN_MAX_QUERY_POINTS]; Q penetration[AVN_MAX_QUERY_POINTS]; uint32_t fid1[AVN_MAX_QUERY_POINTS], fid2[AVN_MAX_QUERY_POINTS]; };
    static_assert(sizeof(HostMM<float>) == sizeof(avn_host_manifold_f32) && sizeof(HostMM<double>) == sizeof(avn_host_manifold_f64), "host manifold layout");
    static_assert(sizeof(HostMQ<float>) == sizeof(avn_host_manifold_query_f32) && sizeof(HostMQ<double>) == sizeof(avn_host_manifold_query_f64), "host query layout");
    static_assert(sizeof(HostAabbQ<float>) == sizeof(avn_host_aabb_query_f32) && sizeof(HostAabbQ<double>) == sizeof(avn_host_aabb_query_f64), "host aabb query layout");
    avn_status host_shapes_set(avn_host_aabb_fn a, avn_host_manifolds_fn m, void* user) override {
        if ((a == nullptr) != (m == nullptr)) { error = "host_shapes_set: both callbacks or none"; return AVN_ERR_BAD_ARG; }
        hs_aabb_fn = a; hs_manifolds_fn = m; hs_user = user;
        return AVN_OK;
    }
    avn_status host_shape_stats_get(avn_host_shape_stats* o) override { if (!o) return AVN_ERR_BAD_ARG; hs_stats.host_colliders = n_host_colliders; *o = hs_stats; return AVN_OK; }
    bool host_contact_manifold(uint32_t id, const Collider<S>& c1, V3<S> p1, Q4<S> r1, const Collider<S>& c2, V3<S> p2, Q4<S> r2, S max_contact_distance, QueryManifold<S>& qm) {
        HostMQ<S> q{id, c1.entity, c2.entity, 0u, {p1.x, p1.y, p1.z}, {r1.x, r1.y, r1.z, r1.w}, {p2.x, p2.y, p2.z}, {r2.x, r2.y, r2.z, r2.w}, max_contact_distance};
        HostMM<S> m;
        std::memset(&m, 0, sizeof m);
        hs_manifolds_fn(hs_user, (uint32_t)(8 * sizeof(S)), 1u, &q, &m);
        hs_stats.bytes_to_host += sizeof q; hs_stats.bytes_from_host += sizeof q + sizeof m;
        qm.n = (int)std::min<uint32_t>(m.point_count, AVN_MAX_QUERY_POINTS);
        qm.normal = {m.normal[0], m.normal[1], m.normal[2]};
        for (int k = 0; k < qm.n; ++k) {
            V3<S> anchor1{m.anchor1[3 * k], m.anchor1[3 * k + 1], m.anchor1[3 * k + 2]};
            qm.pts[k] = {anchor1, anchor1 + (p1 - p2), p1 + anchor1, m.penetration[k], m.fid1[k], m.fid2[k]};   // contact_query.rs:243-248
        }
        hs_stats.last_manifolds_with_points += qm.n != 0;
        return qm.n != 0;
    }
    // the manifolds prepare_contact_constraints reads through GraphColor::manifold_handles (plugin.rs:389-398)
    void gather_manifolds_from_handles() {
        manifolds.resize(manifold_handles.size());
        // (a copy of the reference's in-place access through get_by_id: an artefact of this restatement, run on the pool)
        pool.par_for_each(manifold_handles.size(), 64, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const CtRow& r = contact_rows[manifold_handles[i]];
            ContactManifold<S>& m = manifolds[i];
            m.body1 = colliders[collider_slot.at(r.collider1)].body;
            m.body2 = colliders[collider_slot.at(r.collider2)].body;
            m.normal = r.normal; m.tangent_velocity = r.tangent_velocity;
            m.friction = r.friction; m.restitution = r.restitution;
            m.point_count = (uint8_t)(r.n_manifolds ? r.point_count : 0);
            m.flags = (r.flags & AVN_CP_GENERATE_CONSTRAINTS) ? AVN_MANIFOLD_GENERATES_CONSTRAINTS : 0;
            for (int k = 0; k < m.point_count; ++k) {
                const CtPoint& c = r.pts[k];
                m.points[k] = {c.anchor1, c.anchor2, c.penetration, c.normal_speed, c.warm_start_normal_impulse, c.normal_impulse, c.warm_start_tangent_impulse};
            }
        }
        });
    }
    void scatter_impulses_to_contacts() {
        pool.par_for_each(manifold_handles.size(), 64, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            CtRow& r = contact_rows[manifold_handles[i]];
            const ContactManifold<S>& m = manifolds[i];
            for (int k = 0; k < m.point_count; ++k) {
                r.pts[k].warm_start_normal_impulse = m.points[k].warm_start_normal_impulse;
                r.pts[k].warm_start_tangent_impulse = m.points[k].warm_start_tangent_impulse;
                r.pts[k].normal_impulse = m.points[k].normal_impulse;
            }
        }
        });
    }
    avn_status aabbs_download(void* mn, void* mx, uint32_t* ents, size_t* n_iv) override {
        for (size_t i = 0; i < colliders.size(); ++i) { wr3(mn, i, colliders[i].aabb.min); wr3(mx, i, colliders[i].aabb.max); }
        if (ents) for (size_t i = 0; i < intervals.size(); ++i) ents[i] = colliders[intervals[i].collider].entity;
        if (n_iv) *n_iv = intervals.size();
        return AVN_OK;
    }
    // parry3d 0.25 Cuboid::aabb / Ball::aabb (third party, un-vendored; restated): centre +- |R| * half_extents
    // with R from nalgebra UnitQuaternion::to_rotation_matrix and a column-major gemv.
    static ColliderAabb<S> shape_aabb(const Collider<S>& c, V3<S> pos, Q4<S> q) {
        V3<S> he;
        if (c.shape == AVN_SHAPE_BALL) he = {c.half_extents.x, c.half_extents.x, c.half_extents.x};
        else {
            S i = q.x, j = q.y, k = q.z, w = q.w;
            S ww = w * w, ii = i * i, jj = j * j, kk = k * k;
            S ij = i * j * S(2), wk = w * k * S(2), wj = w * j * S(2), ik = i * k * S(2), jk = j * k * S(2), wi = w * i * S(2);
            S m00 = std::fabs(ww + ii - jj - kk), m01 = std::fabs(ij - wk), m02 = std::fabs(wj + ik);
            S m10 = std::fabs(wk + ij), m11 = std::fabs(ww - ii + jj - kk), m12 = std::fabs(jk - wi);
            S m20 = std::fabs(ik - wj), m21 = std::fabs(wi + jk), m22 = std::fabs(ww - ii - jj + kk);
            V3<S> h = c.half_extents;
            he = {(m00 * h.x + m01 * h.y) + m02 * h.z, (m10 * h.x + m11 * h.y) + m12 * h.z, (m20 * h.x + m21 * h.y) + m22 * h.z};
        }
        return {pos - he, pos + he};
    }
    // collider/backend.rs:498-624 (collider on the rigid-body entity: Position/Rotation/velocities are the body's)
    void update_aabb() {
        S delta_secs = dt_adj;
        S default_speculative_margin = (S)cfg.length_unit * (cfg.default_speculative_margin >= (double)std::numeric_limits<S>::max() ? std::numeric_limits<S>::max() : (S)cfg.default_speculative_margin);
        S contact_tolerance = (S)cfg.length_unit * (S)cfg.contact_tolerance;
        hs_stats.last_aabb_queries = 0;
        for (Collider<S>& c : colliders) {
            const Body<S>& b = bodies[c.body];
            S speculative_margin = (c.cflags & AVN_COLLIDER_SWEPT_CCD) ? std::numeric_limits<S>::max()
                                   : (c.speculative_margin >= S(0) ? c.speculative_margin : default_speculative_margin);
            S g = contact_tolerance + c.collision_margin;
            // the collider's own Position / Rotation; a child collider of a rotating body orbits it: the velocity at its offset from the centre of mass (backend.rs:569-586)
            V3<S> cpos; Q4<S> crot;
            collider_pose(c, b, cpos, crot);
            V3<S> lin_vel = b.linear_velocity;
            if (c.child) { const V3<S> offset = (cpos - b.position) - qrot(b.rotation, b.center_of_mass); lin_vel = b.linear_velocity + cross(b.angular_velocity, offset); }
            if (c.shape == AVN_SHAPE_HOST) {   // aabb_with_context / swept_aabb_with_context on the host (backend.rs:556-620; header: "host shapes")
                const bool swept = !(speculative_margin <= S(0));
                Q4<S> end_rot = crot; V3<S> end_pos = cpos;
                if (swept) {
                    end_rot = fast_renormalize(qmul(from_scaled_axis(b.angular_velocity * delta_secs), crot));
                    end_pos = cpos + clamp_length_max(lin_vel * delta_secs, smax(speculative_margin, contact_tolerance));
                }
                HostAabbQ<S> q{c.entity, swept ? 1u : 0u, {cpos.x, cpos.y, cpos.z}, {crot.x, crot.y, crot.z, crot.w},
                               {end_pos.x, end_pos.y, end_pos.z}, {end_rot.x, end_rot.y, end_rot.z, end_rot.w}};
                HostAabb<S> a{};
                hs_aabb_fn(hs_user, (uint32_t)(8 * sizeof(S)), 1u, &q, &a);
                hs_stats.bytes_to_host += sizeof q; hs_stats.bytes_from_host += sizeof a; ++hs_stats.last_aabb_queries;
                c.aabb = {V3<S>{a.min[0], a.min[1], a.min[2]} - V3<S>{g, g, g}, V3<S>{a.max[0], a.max[1], a.max[2]} + V3<S>{g, g, g}};
                continue;
            }
            if (speculative_margin <= S(0)) {
                ColliderAabb<S> a = shape_aabb(c, cpos, crot);
                c.aabb = {a.min - V3<S>{g, g, g}, a.max + V3<S>{g, g, g}};
                continue;
            }
            Q4<S> end_rot = fast_renormalize(qmul(from_scaled_axis(b.angular_velocity * delta_secs), crot));
            V3<S> end_pos = cpos + clamp_length_max(lin_vel * delta_secs, smax(speculative_margin, contact_tolerance));
            ColliderAabb<S> a0 = shape_aabb(c, cpos, crot);
            ColliderAabb<S> a1 = shape_aabb(c, end_pos, end_rot);
            ColliderAabb<S> m{vmin(a0.min, a1.min), vmax(a0.max, a1.max)};
            c.aabb = {m.min - V3<S>{g, g, g}, m.max + V3<S>{g, g, g}};
        }
    }
    // broad_phase.rs:214-280 (flags refresh), :373-474 sweep_and_prune, :479-487 insertion_sort
    void collect_collision_pairs() {
        // update_aabb_intervals: drop non-finite AABBs, refresh flags
        std::vector<AabbInterval> kept;
        kept.reserve(intervals.size());
        for (AabbInterval iv : intervals) {
            const Collider<S>& c = colliders[iv.collider];
            if (!is_finite(c.aabb.min) || !is_finite(c.aabb.max)) continue;
            const Body<S>& b = bodies[c.body];
            bool is_static = b.rb_type == AVN_RB_STATIC;
            bool is_sleeping = b.body_flags & AVN_BODY_SLEEPING;
            bool is_disabled = b.body_flags & AVN_BODY_DISABLED;
            uint8_t f = 0;
            if (is_static || is_sleeping) f |= AVN_AABB_IS_INACTIVE;
            if (c.cflags & AVN_COLLIDER_EVENTS) f |= AVN_AABB_CONTACT_EVENTS;
            if (!(c.cflags & AVN_COLLIDER_SENSOR) && !is_disabled) f |= AVN_AABB_GENERATE_CONSTRAINTS;
            if (c.cflags & AVN_COLLIDER_FILTER_PAIRS) f |= AVN_AABB_CUSTOM_FILTER;
            if (c.cflags & AVN_COLLIDER_MODIFY_CONTACTS) f |= AVN_AABB_MODIFY_CONTACTS;
            kept.push_back({iv.collider, f});
        }
        intervals.swap(kept);
        // insertion_sort(|a, b| a.min.x > b.min.x), broad_phase.rs:479-487.  The reference's first frame over an unsorted
        // list is O(n^2); to keep the CHECKER usable at 10^6 colliders the loop hands over to std::stable_sort once it has
        // done more than 64 n swaps.  Same result: an insertion sort that swaps only on strict `>` is a stable ascending sort,
        // every intermediate state keeps equal keys in their original relative order, and the stable sorted permutation is unique.
        {
            size_t swaps = 0, budget = 64 * intervals.size() + 1024;
            bool handed_over = false;
            for (size_t i = 1; i < intervals.size() && !handed_over; ++i) {
                size_t j = i;
                while (j > 0 && colliders[intervals[j - 1].collider].aabb.min.x > colliders[intervals[j].collider].aabb.min.x) {
                    std::swap(intervals[j - 1], intervals[j]);
                    --j;
                    if (++swaps > budget) { handed_over = true; break; }
                }
            }
            if (handed_over)
                std::stable_sort(intervals.begin(), intervals.end(), [&](const AabbInterval& a, const AabbInterval& b) {
                    return colliders[a.collider].aabb.min.x < colliders[b.collider].aabb.min.x;
                });
        }
        pairs.clear();
        hk_stats.last_filter_queries = hk_stats.last_filter_rejected = 0;
        for (size_t i = 0; i < intervals.size(); ++i) {
            const Collider<S>& c1 = colliders[intervals[i].collider];
            uint8_t flags1 = intervals[i].flags;
            for (size_t j = i + 1; j < intervals.size(); ++j) {
                const Collider<S>& c2 = colliders[intervals[j].collider];
                uint8_t flags2 = intervals[j].flags;
                if (c2.aabb.min.x > c1.aabb.max.x) break;
                if (c1.aabb.min.y > c2.aabb.max.y || c1.aabb.max.y < c2.aabb.min.y) continue;
                if (c1.aabb.min.z > c2.aabb.max.z || c1.aabb.max.z < c2.aabb.min.z) continue;
                bool interacts = (c1.memberships & c2.filters) != 0 && (c2.memberships & c1.filters) != 0;  // layers.rs:423-426
                if ((flags1 & flags2 & AVN_AABB_IS_INACTIVE) || !interacts || c1.body == c2.body) continue;
                uint64_t key = pair_key(c1.entity, c2.entity);
                if (pair_set.count(key)) continue;
                if (collision_disabled_bodies.count(pair_key((uint32_t)c1.body, (uint32_t)c2.body))) continue;
                uint8_t u = flags1 | flags2;
                if ((u & AVN_AABB_CUSTOM_FILTER) && hk_filter_fn) {   // CollisionHooks::filter_pairs, broad_phase.rs:431-439 (header: "collision hooks")
                    const avn_hook_pair q{(uint32_t)pairs.size() + hk_stats.last_filter_rejected, c1.entity, c2.entity};
                    uint8_t should_collide = 1;
                    hk_filter_fn(hk_user, 1u, &q, &should_collide);
                    ++hk_stats.last_filter_queries; hk_stats.bytes_to_host += sizeof q; hk_stats.bytes_from_host += 1;
                    if (!should_collide) { ++hk_stats.last_filter_rejected; continue; }
                }
                avn_pair p;
                p.collider1 = c1.entity; p.collider2 = c2.entity; p.body1 = c1.body; p.body2 = c2.body;
                p.flags = 0; p.reserved = 0;
                if (u & AVN_AABB_CONTACT_EVENTS) p.flags |= AVN_PAIR_CONTACT_EVENTS;
                if (u & AVN_AABB_MODIFY_CONTACTS) p.flags |= AVN_PAIR_MODIFY_CONTACTS;
                if (u & AVN_AABB_GENERATE_CONSTRAINTS) p.flags |= AVN_PAIR_GENERATE_CONSTRAINTS;
                if (u & AVN_AABB_CUSTOM_FILTER) p.flags |= AVN_PAIR_NEEDS_CUSTOM_FILTER;
                pairs.push_back(p);
                // add_edge_and_key_with inserts the key (contact_graph.rs:521-566): a later duplicate is skipped
                pair_set.insert(key);
            }
        }
        last_timers.pair_count = (uint32_t)pairs.size();
    }

    // =============================================================================================
    //                                      SCHEDULES
    // =============================================================================================
    // SolverDiagnostics / CollisionDiagnostics of the last step, wall clock like the reference's own Instant::now() pairs (plugin.rs:459,481)
    avn_diagnostics diag{};
    template <class Fn> void timed(double& acc, Fn&& f) {
        auto t0 = std::chrono::steady_clock::now();
        f();
        acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void substep() {  // SubstepSchedule order, solver/schedule.rs:59-69 + xpbd/plugin.rs:30-40
        timed(diag.integrate_velocities_ms, [&] { integrate_velocities(); });
        timed(diag.warm_start_ms, [&] { warm_start(); });
        timed(diag.solve_constraints_ms, [&] { for (uint32_t it = 0; it < cfg.solver_iterations; ++it) solve_contacts(true); });
        timed(diag.integrate_positions_ms, [&] { integrate_positions(); });
        timed(diag.relax_velocities_ms, [&] {
            for (uint32_t it = 0; it < cfg.solver_iterations; ++it) solve_contacts(false);
            for (uint32_t it = 0; it < cfg.solver_iterations; ++it) xpbd_solve_iter(it);
            xpbd_velocity_projection();
            joint_damping();
        });
    }
    // extension: extra joint iterations must not re-take the pre-solve snapshot
    void xpbd_solve_iter(uint32_t it) {
        if (it == 0) { xpbd_solve(); return; }
        std::vector<V3<S>> sp; std::vector<Q4<S>> sq;
        for (Body<S>& b : bodies) { sp.push_back(b.pre_solve_delta_position); sq.push_back(b.pre_solve_delta_rotation); }
        xpbd_solve();
        for (size_t i = 0; i < bodies.size(); ++i) { bodies[i].pre_solve_delta_position = sp[i]; bodies[i].pre_solve_delta_rotation = sq[i]; }
    }
    void solver() {  // SolverSystems, solver/schedule.rs:32-46
        const double bp = diag.broad_phase_ms, np_ms = diag.narrow_phase_ms;
        diag = avn_diagnostics{};
        diag.broad_phase_ms = bp; diag.narrow_phase_ms = np_ms; diag.per_system_valid = 1;
        timed(diag.prepare_constraints_ms, [&] { prepare_solver_bodies(); prepare_joints(); prepare_contact_constraints(); });
        timed(diag.update_velocity_increments_ms, [&] { pre_process_velocity_increments(); });
        timed(diag.substeps_ms, [&] { for (uint32_t s = 0; s < cfg.substeps; ++s) substep(); });
        timed(diag.apply_restitution_ms, [&] { clear_velocity_increments(); solve_restitution(); });
        timed(diag.finalize_ms, [&] { writeback_solver_bodies(); writeback_joint_forces(); });
        timed(diag.store_impulses_ms, [&] { store_contact_impulses(); });
        diag.contact_constraint_count = last_timers.contact_constraint_count;
    }
    avn_status run_system(avn_system sys) override {
        switch (sys) {
            case AVN_SYS_UPDATE_AABB: if (n_host_colliders && !hs_aabb_fn) { error = "the world holds AVN_SHAPE_HOST colliders and no callbacks (avn_host_shapes_set)"; return AVN_ERR_STATE; } update_aabb(); break;
            case AVN_SYS_COLLECT_COLLISION_PAIRS: collect_collision_pairs(); break;
            case AVN_SYS_PREPARE_SOLVER_BODIES: prepare_solver_bodies(); break;
            case AVN_SYS_PREPARE_JOINTS: prepare_joints(); break;
            case AVN_SYS_PREPARE_CONTACT_CONSTRAINTS: prepare_contact_constraints(); break;
            case AVN_SYS_PRE_PROCESS_VELOCITY_INCREMENTS: pre_process_velocity_increments(); break;
            case AVN_SYS_INTEGRATE_VELOCITIES: integrate_velocities(); break;
            case AVN_SYS_WARM_START: warm_start(); break;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: solve_contacts(true); break;
            case AVN_SYS_INTEGRATE_POSITIONS: integrate_positions(); break;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: solve_contacts(false); break;
            case AVN_SYS_XPBD_SOLVE: xpbd_solve(); break;
            case AVN_SYS_XPBD_VELOCITY_PROJECTION: xpbd_velocity_projection(); break;
            case AVN_SYS_JOINT_DAMPING: joint_damping(); break;
            case AVN_SYS_CLEAR_VELOCITY_INCREMENTS: clear_velocity_increments(); break;
            case AVN_SYS_SOLVE_RESTITUTION: solve_restitution(); break;
            case AVN_SYS_WRITEBACK_SOLVER_BODIES: writeback_solver_bodies(); writeback_joint_forces(); break;
            case AVN_SYS_STORE_CONTACT_IMPULSES: store_contact_impulses(); break;
            case AVN_SYS_SUBSTEP: substep(); break;
            case AVN_SYS_SOLVER: solver(); break;
            case AVN_SYS_NARROW_PHASE: if (n_host_colliders && !hs_aabb_fn) { error = "the world holds AVN_SHAPE_HOST colliders and no callbacks (avn_host_shapes_set)"; return AVN_ERR_STATE; } narrow_phase(); break;
            default: error = "run_system: unknown system"; return AVN_ERR_BAD_ARG;
        }
        return AVN_OK;
    }
    avn_status step() override {
        if (n_host_colliders && !hs_aabb_fn) { error = "the world holds AVN_SHAPE_HOST colliders and no callbacks (avn_host_shapes_set)"; return AVN_ERR_STATE; }
        if (pipe) return pipeline_step();
        diag.broad_phase_ms = 0; diag.narrow_phase_ms = 0;
        if (have_colliders) timed(diag.broad_phase_ms, [&] { update_aabb(); collect_collision_pairs(); });
        solver();
        diag.contact_count = (uint32_t)pairs.size();
        return AVN_OK;
    }
    // ---- level-2 sharding (header: avn_halo_plan) ----
    struct Halo { std::vector<int32_t> peers; std::vector<uint32_t> send_off, recv_off; std::vector<int32_t> send, recv; } halo;
    avn_status halo_overflow_levels_upload(uint32_t n_levels, const uint32_t* level_of, size_t count) override {
        if (count && !level_of) { error = "halo_overflow_levels_upload: null array"; return AVN_ERR_BAD_ARG; }
        for (size_t i = 0; i < count; ++i) if (level_of[i] >= std::max(n_levels, 1u)) { error = "halo_overflow_levels_upload: level out of range"; return AVN_ERR_BAD_ARG; }
        l2_levels = std::max(n_levels, 1u);
        l2_level_of.assign(level_of, level_of + count);
        halo = Halo();
        return AVN_OK;
    }
    // the joint slot (header: avn_halo_joint_slot_set): behind the colours and levels, records = the whole SolverBody (16 scalars)
    bool l2_joint_slot = false;
    avn_status halo_joint_slot_set(uint32_t joint_slot, uint32_t /* global_joints: this restatement always runs the XPBD body passes */) override {
        l2_joint_slot = joint_slot != 0;
        halo = Halo();
        return AVN_OK;
    }
    uint32_t halo_slots() const { return (uint32_t)AVN_COLOR_OVERFLOW_INDEX + l2_levels + (l2_joint_slot ? 1u : 0u); }
    bool is_joint_slot(uint32_t slot) const { return l2_joint_slot && slot == (uint32_t)AVN_COLOR_OVERFLOW_INDEX + l2_levels; }
    avn_status halo_plan_upload(const avn_halo_plan* p) override {
        if (!p) return AVN_ERR_BAD_ARG;
        const size_t n = (size_t)halo_slots() * p->n_peers;
        if (p->n_peers && (!p->peer_rank || !p->send_offsets || !p->recv_offsets)) { error = "halo_plan_upload: null array"; return AVN_ERR_BAD_ARG; }
        halo.peers.assign(p->peer_rank, p->peer_rank + p->n_peers);
        halo.send_off.assign(p->send_offsets, p->send_offsets + (p->n_peers ? n + 1 : 0)); halo.recv_off.assign(p->recv_offsets, p->recv_offsets + (p->n_peers ? n + 1 : 0));
        const size_t ns = p->n_peers ? halo.send_off[n] : 0, nr = p->n_peers ? halo.recv_off[n] : 0;
        halo.send.assign(p->send_bodies, p->send_bodies + ns); halo.recv.assign(p->recv_bodies, p->recv_bodies + nr);
        for (int32_t b : halo.send) if (b < 0 || (size_t)b >= bodies.size()) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        for (int32_t b : halo.recv) if (b < 0 || (size_t)b >= bodies.size()) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        return AVN_OK;
    }
    avn_status run_color_pass(avn_system pass, uint32_t color) override {
        if (color >= halo_slots() || is_joint_slot(color)) { error = "run_color_pass: colour / slot out of range"; return AVN_ERR_BAD_ARG; }
        only_color = (int)std::min<uint32_t>(color, AVN_COLOR_OVERFLOW_INDEX);
        only_level = (l2_levels > 1 && color >= (uint32_t)AVN_COLOR_OVERFLOW_INDEX) ? (int)(color - AVN_COLOR_OVERFLOW_INDEX) : -1;
        if (only_level >= 0 && l2_level_of.size() != color_constraints[AVN_COLOR_OVERFLOW_INDEX].size()) { only_color = only_level = -1; error = "level-2: avn_halo_overflow_levels_upload does not name this world's overflow manifolds"; return AVN_ERR_STATE; }
        avn_status st = AVN_OK;
        switch (pass) {
            case AVN_SYS_WARM_START: warm_start(); break;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: solve_contacts(true); break;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: solve_contacts(false); break;
            case AVN_SYS_SOLVE_RESTITUTION: solve_restitution(); break;
            default: error = "run_color_pass: not a contact pass"; st = AVN_ERR_BAD_ARG;
        }
        only_color = -1; only_level = -1;
        return st;
    }
    avn_status halo_pack(uint32_t color, uint32_t peer, void* out, size_t* count) override {
        if (color >= halo_slots() || peer >= halo.peers.size() || !count) return AVN_ERR_BAD_ARG;
        const size_t k = (size_t)color * halo.peers.size() + peer, b0 = halo.send_off[k], b1 = halo.send_off[k + 1];
        *count = b1 - b0;
        S* o = (S*)out;
        if (is_joint_slot(color)) {
            for (size_t i = b0; i < b1 && o; ++i) {
                const SolverBody<S>& sb = bodies[halo.send[i]].sb;
                S* r = o + 16 * (i - b0);
                r[0] = sb.delta_position.x; r[1] = sb.delta_position.y; r[2] = sb.delta_position.z; r[3] = S(0);
                r[4] = sb.delta_rotation.x; r[5] = sb.delta_rotation.y; r[6] = sb.delta_rotation.z; r[7] = sb.delta_rotation.w;
                r[8] = sb.linear_velocity.x; r[9] = sb.linear_velocity.y; r[10] = sb.linear_velocity.z; r[11] = S(0);
                r[12] = sb.angular_velocity.x; r[13] = sb.angular_velocity.y; r[14] = sb.angular_velocity.z; r[15] = S(0);
            }
            return AVN_OK;
        }
        for (size_t i = b0; i < b1 && o; ++i) {
            const SolverBody<S>& sb = bodies[halo.send[i]].sb;
            S* r = o + 8 * (i - b0);
            r[0] = sb.linear_velocity.x; r[1] = sb.linear_velocity.y; r[2] = sb.linear_velocity.z; r[3] = S(0);
            r[4] = sb.angular_velocity.x; r[5] = sb.angular_velocity.y; r[6] = sb.angular_velocity.z; r[7] = S(0);
        }
        return AVN_OK;
    }
    avn_status halo_unpack(uint32_t color, uint32_t peer, const void* in, size_t count) override {
        if (color >= halo_slots() || peer >= halo.peers.size()) return AVN_ERR_BAD_ARG;
        const size_t k = (size_t)color * halo.peers.size() + peer, b0 = halo.recv_off[k], b1 = halo.recv_off[k + 1];
        if (count != b1 - b0 || (count && !in)) { error = "halo_unpack: count does not match the plan"; return AVN_ERR_BAD_ARG; }
        const S* r = (const S*)in;
        if (is_joint_slot(color)) {
            for (size_t i = b0; i < b1; ++i, r += 16) {
                SolverBody<S>& sb = bodies[halo.recv[i]].sb;
                sb.delta_position = V3<S>{r[0], r[1], r[2]}; sb.delta_rotation = Q4<S>{r[4], r[5], r[6], r[7]};
                sb.linear_velocity = V3<S>{r[8], r[9], r[10]}; sb.angular_velocity = V3<S>{r[12], r[13], r[14]};
            }
            return AVN_OK;
        }
        for (size_t i = b0; i < b1; ++i, r += 8) {
            SolverBody<S>& sb = bodies[halo.recv[i]].sb;
            sb.linear_velocity = V3<S>{r[0], r[1], r[2]}; sb.angular_velocity = V3<S>{r[4], r[5], r[6]};
        }
        return AVN_OK;
    }
    // ---- islands and sleeping: the checker of avn_islands_get / avn_sleep_update (header).  Deliberately NOT the product's algorithm: a
    // breadth-first flood over adjacency lists in ascending body order (the first body reached is the island's lowest index), and the
    // serial loops of update_sleeping_states / sleep_islands (islands/sleeping.rs:184-280) over bodies and islands.
    std::vector<uint32_t> isl_label;
    std::vector<float> sleep_timer;
    std::vector<uint8_t> isl_rests, isl_wakes;
    uint32_t isl_count = 0, isl_nodes = 0;
    void islands_compute() {
        const size_t n = bodies.size();
        std::vector<std::vector<uint32_t>> adj(n);
        auto node = [&](int32_t b) { return b >= 0 && (size_t)b < n && bodies[b].rb_type != AVN_RB_STATIC && !(bodies[b].body_flags & AVN_BODY_DISABLED); };   // BodyIslandNode, islands/mod.rs:96-140
        for (const auto& m : manifolds) if (node(m.body1) && node(m.body2) && m.body1 != m.body2) { adj[m.body1].push_back((uint32_t)m.body2); adj[m.body2].push_back((uint32_t)m.body1); }
        for (const auto& j : joints) if (node(j.body1) && node(j.body2) && j.body1 != j.body2) { adj[j.body1].push_back((uint32_t)j.body2); adj[j.body2].push_back((uint32_t)j.body1); }
        isl_label.assign(n, 0xFFFFFFFFu);
        isl_count = 0; isl_nodes = 0;
        std::vector<uint32_t> queue;
        for (size_t s = 0; s < n; ++s) {
            if (!node((int32_t)s)) continue;
            ++isl_nodes;
            if (isl_label[s] != 0xFFFFFFFFu) continue;
            ++isl_count;
            isl_label[s] = (uint32_t)s;
            queue.assign(1, (uint32_t)s);
            for (size_t q = 0; q < queue.size(); ++q)
                for (uint32_t o : adj[queue[q]]) if (isl_label[o] == 0xFFFFFFFFu) { isl_label[o] = (uint32_t)s; queue.push_back(o); }
        }
    }
    avn_status islands_get(uint32_t* island_of_body, uint32_t* n_islands) override {
        islands_compute();
        if (island_of_body) std::copy(isl_label.begin(), isl_label.end(), island_of_body);
        if (n_islands) *n_islands = isl_count;
        return AVN_OK;
    }
    avn_status sleep_update(const avn_sleep_params* sp, avn_sleep_stats* out) override {
        if (!sp || sp->struct_size != sizeof(avn_sleep_params)) { error = "sleep_update: bad params"; return AVN_ERR_BAD_ARG; }
        islands_compute();
        const size_t n = bodies.size();
        if (sleep_timer.size() != n) sleep_timer.assign(n, 0.0f);
        const S length_unit_squared = (S)sp->length_unit * (S)sp->length_unit;
        const float delta_secs = sp->delta_secs;
        std::vector<uint8_t> awake(n, 0), is_sleeping(n, 0);   // AwakeIslandBitVec and PhysicsIsland::is_sleeping, indexed by island label
        uint32_t sleeping_bodies = 0;
        for (size_t b = 0; b < n; ++b) {
            if (isl_label[b] == 0xFFFFFFFFu) { sleep_timer[b] = 0.0f; continue; }
            if (bodies[b].body_flags & AVN_BODY_SLEEPING) ++sleeping_bodies;
            // wake_islands_with_sleeping_disabled (sleeping.rs:164-182; the reference chains it after update_sleeping_states, whose query skips these bodies)
            if (sp->body_sleeping_disabled && sp->body_sleeping_disabled[b]) {
                awake[isl_label[b]] = 1; sleep_timer[b] = 0.0f;
                if (bodies[b].body_flags & AVN_BODY_SLEEPING) is_sleeping[isl_label[b]] = 1;
                continue;
            }
            if (bodies[b].body_flags & AVN_BODY_SLEEPING) { is_sleeping[isl_label[b]] = 1; continue; }   // query filter Without<Sleeping>
            const SolverBody<S>& sb = bodies[b].sb;
            const S lin_vel_squared = length_squared(sb.linear_velocity), ang_vel_squared = length_squared(sb.angular_velocity);
            // "Keep signs."
            const float lt = sp->body_linear_threshold ? sp->body_linear_threshold[b] : sp->linear_threshold;
            const float at = sp->body_angular_threshold ? sp->body_angular_threshold[b] : sp->angular_threshold;
            const float lin_threshold_squared = lt * std::fabs(lt);
            const float ang_threshold_squared = at * std::fabs(at);
            if (lin_vel_squared < length_unit_squared * (S)lin_threshold_squared && ang_vel_squared < (S)ang_threshold_squared) sleep_timer[b] += delta_secs;
            else sleep_timer[b] = 0.0f;
            if (sleep_timer[b] < sp->time_to_sleep) awake[isl_label[b]] = 1;
        }
        // sleep_islands (sleeping.rs:256-266): awake bit and sleeping -> wake; no awake bit, not sleeping (and no pending split) -> sleep
        isl_rests.assign(n, 0); isl_wakes.assign(n, 0);
        uint32_t resting_islands = 0, resting_bodies = 0, waking_islands = 0, waking_bodies = 0;
        for (size_t b = 0; b < n; ++b) {
            const uint32_t l = isl_label[b];
            if (l == 0xFFFFFFFFu) continue;
            if (awake[l] && is_sleeping[l]) { isl_wakes[b] = 1; if (l == b) ++waking_islands; if (bodies[b].body_flags & AVN_BODY_SLEEPING) ++waking_bodies; }
            else if (!awake[l] && !is_sleeping[l]) { isl_rests[b] = 1; ++resting_bodies; if (l == b) ++resting_islands; }
        }
        if (out) { out->n_islands = isl_count; out->n_island_bodies = isl_nodes; out->n_sleeping_bodies = sleeping_bodies; out->n_awake_bodies = isl_nodes - sleeping_bodies;
                   out->n_resting_islands = resting_islands; out->n_resting_bodies = resting_bodies; out->n_waking_islands = waking_islands; out->n_waking_bodies = waking_bodies; }
        return AVN_OK;
    }
    avn_status sleep_get(const avn_sleep_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        if (sleep_timer.size() != bodies.size() || isl_rests.size() != bodies.size()) { error = "sleep_get: call avn_sleep_update first"; return AVN_ERR_STATE; }
        if (o->sleep_timer) std::copy(sleep_timer.begin(), sleep_timer.end(), o->sleep_timer);
        if (o->island) std::copy(isl_label.begin(), isl_label.end(), o->island);
        if (o->island_rests) std::copy(isl_rests.begin(), isl_rests.end(), o->island_rests);
        if (o->island_wakes) std::copy(isl_wakes.begin(), isl_wakes.end(), o->island_wakes);
        return AVN_OK;
    }
    avn_status sleep_reset(const uint32_t* ids, size_t n) override {
        if (sleep_timer.size() != bodies.size()) sleep_timer.assign(bodies.size(), 0.0f);
        if (!ids || n == 0) { std::fill(sleep_timer.begin(), sleep_timer.end(), 0.0f); return AVN_OK; }
        for (size_t i = 0; i < n; ++i) if (ids[i] < sleep_timer.size()) sleep_timer[ids[i]] = 0.0f;
        return AVN_OK;
    }
    avn_status timers(avn_timers* t) override { if (!t) return AVN_ERR_BAD_ARG; *t = last_timers; return AVN_OK; }
    avn_status diagnostics(avn_diagnostics* d) override { if (!d) return AVN_ERR_BAD_ARG; *d = diag; return AVN_OK; }
    avn_status profile_system(avn_system sys, uint32_t repeats, double* total_ms, uint32_t* launches) override {
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t r = 0; r < repeats; ++r) { avn_status st = run_system(sys); if (st != AVN_OK) return st; }
        auto t1 = std::chrono::steady_clock::now();
        if (total_ms) *total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (launches) *launches = 0;
        return AVN_OK;
    }
};

// ---- ConstraintGraph (solver/constraint_graph.rs:129-296) ------------------------------------------
struct ConstraintGraph {
    struct Handle { uint64_t handle; uint32_t body1, body2; };
    struct Color { std::vector<bool> body_set; std::vector<Handle> manifold_handles; };
    Color colors[AVN_GRAPH_COLOR_COUNT];
    struct Loc { uint8_t color; uint32_t local_index; };
    std::unordered_map<uint64_t, Loc> where;  // ContactEdge::constraint_handles
    static bool get(const std::vector<bool>& s, uint32_t i) { return i < s.size() && s[i]; }
    static void set_and_grow(std::vector<bool>& s, uint32_t i) { if (i >= s.size()) s.resize((size_t)i + 1, false); s[i] = true; }
    static void unset(std::vector<bool>& s, uint32_t i) { if (i < s.size()) s[i] = false; }
    int push_manifold(uint64_t handle, uint32_t body1, uint32_t body2, bool is_static1, bool is_static2) {
        if (where.count(handle)) return -1;
        int color_index = AVN_COLOR_OVERFLOW_INDEX;
        if (!is_static1 && !is_static2) {
            for (int i = 0; i < AVN_DYNAMIC_COLOR_COUNT; ++i) {
                Color& c = colors[i];
                if (get(c.body_set, body1) || get(c.body_set, body2)) continue;
                set_and_grow(c.body_set, body1); set_and_grow(c.body_set, body2);
                color_index = i; break;
            }
        } else if (!is_static1) {
            for (int i = AVN_COLOR_OVERFLOW_INDEX - 1; i >= 1; --i) {
                Color& c = colors[i];
                if (get(c.body_set, body1)) continue;
                set_and_grow(c.body_set, body1);
                color_index = i; break;
            }
        } else if (!is_static2) {
            for (int i = AVN_COLOR_OVERFLOW_INDEX - 1; i >= 1; --i) {
                Color& c = colors[i];
                if (get(c.body_set, body2)) continue;
                set_and_grow(c.body_set, body2);
                color_index = i; break;
            }
        }
        Color& c = colors[color_index];
        where[handle] = {(uint8_t)color_index, (uint32_t)c.manifold_handles.size()};
        c.manifold_handles.push_back({handle, body1, body2});
        return color_index;
    }
    bool pop_manifold(uint64_t handle) {
        auto it = where.find(handle);
        if (it == where.end()) return false;
        Loc loc = it->second;
        where.erase(it);
        Color& c = colors[loc.color];
        Handle h = c.manifold_handles[loc.local_index];
        if (loc.color != AVN_COLOR_OVERFLOW_INDEX) { unset(c.body_set, h.body1); unset(c.body_set, h.body2); }
        uint32_t moved_index = (uint32_t)c.manifold_handles.size() - 1;
        c.manifold_handles[loc.local_index] = c.manifold_handles[moved_index];  // swap_remove
        c.manifold_handles.pop_back();
        if (moved_index != loc.local_index) where[c.manifold_handles[loc.local_index].handle].local_index = loc.local_index;
        return true;
    }
};

// ---- standalone closed loop (header: avn_pipeline_enable): IdPool (id_pool.rs:31-40), ContactGraph edge bookkeeping
//      (contact_graph.rs:521-566), the status-change loop of NarrowPhase::update (system_param.rs:141-389) ----------------
struct PipelineState {
    std::set<uint32_t> free_ids;  // lowest free id first
    uint32_t next_id = 0;
    struct Pair { uint32_t c1, c2; int32_t b1, b2; uint32_t n_handles; };
    std::map<uint32_t, Pair> pairs;
    std::vector<uint32_t> active;
    ConstraintGraph graph;
    // the ContactGraph's adjacency (StableUnGraph: a node per collider, per-node edge lists linked at the head): what remove_collider walks
    IslandManager::Lists lists;
    std::unordered_map<uint32_t, uint32_t> node_of_collider;
    uint32_t node(uint32_t collider) { auto it = node_of_collider.find(collider); if (it != node_of_collider.end()) return it->second; const uint32_t n = (uint32_t)node_of_collider.size() + nodes_dropped; node_of_collider.emplace(collider, n); return n; }
    uint32_t nodes_dropped = 0;   // (node indices are never reused: a removed collider's node stays behind, empty)
    std::vector<uint32_t> handles, report_handles;   // the solver's list | the ConstraintGraph's lists as avn_pipeline_handles_get reports them (equal unless the loop is sharded)
    uint32_t offsets[AVN_GRAPH_COLOR_COUNT + 1] = {0}, report_offsets[AVN_GRAPH_COLOR_COUNT + 1] = {0};
    avn_pipeline_stats stats;
    bool handles_dirty = true;
    PipelineState() { std::memset(&stats, 0, sizeof stats); }
};
inline PipelineState* pipeline_new() { return new PipelineState(); }
inline void pipeline_delete(PipelineState* p) { delete p; }

template <class S> avn_status World<S>::pipeline_enable(int on) {
    if (on && !have_colliders) { error = "pipeline_enable: upload bodies and colliders first"; return AVN_ERR_STATE; }
    if (on && pipe) return AVN_OK;
    if (pipe) {
        if (slp) sleeping_enable(nullptr);   // the island manager goes with the loop: every island awake again first
        for (auto& kv : pipe->pairs) { uint32_t id = kv.first; contact_pairs_remove(&id, 1); }
        pipeline_delete(pipe); pipe = nullptr;
    }
    active_pairs.clear();
    if (on) pipe = pipeline_new();
    return AVN_OK;
}
template <class S> avn_status World<S>::pipeline_stats_get(avn_pipeline_stats* o) {
    if (!o) return AVN_ERR_BAD_ARG;
    if (!pipe) { std::memset(o, 0, sizeof *o); return AVN_OK; }
    pipe->stats.active_pairs = (uint32_t)pipe->active.size();
    pipe->stats.manifolds = (uint32_t)pipe->handles.size();
    *o = pipe->stats;
    return AVN_OK;
}
template <class S> avn_status World<S>::pipeline_handles_get(uint32_t* off, const uint32_t** ids, size_t* n) {
    if (!off || !ids || !n) return AVN_ERR_BAD_ARG;
    static const uint32_t none = 0;
    if (!pipe) { std::memset(off, 0, sizeof(uint32_t) * (AVN_GRAPH_COLOR_COUNT + 1)); *ids = &none; *n = 0; return AVN_OK; }
    std::memcpy(off, pipe->report_offsets, sizeof pipe->report_offsets);
    *ids = pipe->report_handles.data(); *n = pipe->report_handles.size();
    return AVN_OK;
}
// GraphColor::manifold_handles of all colours, concatenated colour-major -> the solver's handle list (when the lists changed)
template <class S> avn_status World<S>::pipeline_refresh_handles() {
    PipelineState& P = *pipe;
    if (!P.handles_dirty) return AVN_OK;
    size_t n = 0, g = 0;
    P.handles.clear(); P.report_handles.clear();
    for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
        P.offsets[c] = (uint32_t)n; P.report_offsets[c] = (uint32_t)g;
        for (const auto& h : P.graph.colors[c].manifold_handles) {
            P.report_handles.push_back((uint32_t)(h.handle >> 8)); ++g;   // what avn_pipeline_handles_get reports: the ConstraintGraph's lists (the whole world's on every rank of a shard)
            if (dsh_on) {   // the solver's share: the manifolds of the bodies this rank simulates, order kept
                const bool f1 = dsh_foreign(h.body1), f2 = dsh_foreign(h.body2);
                const bool d1 = bodies[h.body1].rb_type != AVN_RB_STATIC, d2 = bodies[h.body2].rb_type != AVN_RB_STATIC;
                if (d1 && d2 && f1 != f2) { error = "sharded closed loop: a manifold joins bodies of two ranks -- their islands have met"; return AVN_ERR_STATE; }
                if ((d1 && f1) || (d2 && f2)) continue;
            }
            P.handles.push_back((uint32_t)(h.handle >> 8)); ++n;
        }
    }
    P.offsets[AVN_GRAPH_COLOR_COUNT] = (uint32_t)n; P.report_offsets[AVN_GRAPH_COLOR_COUNT] = (uint32_t)g;
    dsh_own_manifolds = (uint32_t)n; dsh_global_manifolds = (uint32_t)g;
    avn_status st = manifold_handles_upload(P.offsets, P.handles.data());
    if (st != AVN_OK) return st;
    P.handles_dirty = false;
    return AVN_OK;
}
template <class S> avn_status World<S>::pipeline_step() {
    PipelineState& P = *pipe;
    if (despawn_needs_bodies || despawn_needs_colliders) { error = "avn_step: avn_despawn must be followed by avn_bodies_upload and avn_colliders_upload of what remains"; return AVN_ERR_STATE; }
    diag.broad_phase_ms = 0; diag.narrow_phase_ms = 0;
    timed(diag.broad_phase_ms, [&] { update_aabb(); collect_collision_pairs(); });
    auto np_t0 = std::chrono::steady_clock::now();
    // AVO_PIPE_TIMING=1 (debug aid): where a closed-loop step of this restatement spends its wall clock
    static const bool pipe_timing_on = std::getenv("AVO_PIPE_TIMING") != nullptr;
    double tm_phase[5] = {0, 0, 0, 0, 0};
    auto tm_last = np_t0;
    auto tm_mark = [&](int k) { auto now = std::chrono::steady_clock::now(); tm_phase[k] += std::chrono::duration<double, std::milli>(now - tm_last).count(); tm_last = now; };
    new_pair_ids.clear();
    if (!pairs.empty()) {
        std::vector<uint32_t> ids, c1, c2, fl;
        for (const avn_pair& pr : pairs) {
            uint32_t id;
            if (!P.free_ids.empty()) { id = *P.free_ids.begin(); P.free_ids.erase(P.free_ids.begin()); } else id = P.next_id++;
            P.pairs[id] = {pr.collider1, pr.collider2, pr.body1, pr.body2, 0u};
            P.lists.add_edge(id, P.node(pr.collider1), P.node(pr.collider2));
            P.active.push_back(id);
            ids.push_back(id); c1.push_back(pr.collider1); c2.push_back(pr.collider2); fl.push_back(pr.flags);
        }
        avn_contact_pairs cp{(uint32_t)ids.size(), ids.data(), c1.data(), c2.data(), fl.data()};
        avn_status st = contact_pairs_add(&cp);
        if (st != AVN_OK) return st;
        new_pair_ids = ids;
        P.stats.pairs_added += ids.size();
        if (slp) for (size_t i = 0; i < ids.size(); ++i) { st = slp->isl.pair_add(ids[i], c1[i], c2[i]); if (st != AVN_OK) { error = slp->isl.error; return st; } }
    }
    tm_mark(0);
    avn_status st = active_pairs_set(P.active.data(), P.active.size());
    if (st != AVN_OK) return st;
    narrow_phase();
    tm_mark(1);
    auto push = [&](uint32_t cid, uint32_t flags) {
        PipelineState::Pair& p = P.pairs[cid];
        P.graph.push_manifold(((uint64_t)cid << 8) | p.n_handles, (uint32_t)p.b1, (uint32_t)p.b2, flags & AVN_CP_STATIC1, flags & AVN_CP_STATIC2);
        ++p.n_handles; P.handles_dirty = true; ++P.stats.manifolds_pushed;
    };
    auto pop = [&](uint32_t cid) {
        PipelineState::Pair& p = P.pairs[cid];
        if (!p.n_handles) return;
        --p.n_handles;
        P.graph.pop_manifold(((uint64_t)cid << 8) | p.n_handles);
        P.handles_dirty = true; ++P.stats.manifolds_popped;
    };
    std::vector<uint32_t> removed;
    // AVO_PIPE_STATS=1 (debug aid for the device constraint graph's design): shape of this step's push / pop sequence
    static const bool pipe_stats_on = std::getenv("AVO_PIPE_STATS") != nullptr;
    struct OpRec { uint8_t push; uint8_t color; uint32_t b1, b2, pos, len; bool s1, s2; };
    std::vector<OpRec> oprecs;
    for (const avn_contact_change& c : contact_changes) {
        uint32_t cid = c.contact_id, flags = c.flags;
        bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
        if (slp) { avn_status si = slp->isl.status_change(cid, flags, c.manifold_count); if (si != AVN_OK) { error = slp->isl.error; return si; } }
        if (pipe_stats_on) {
            PipelineState::Pair& p = P.pairs[cid];
            bool is_pop = (flags & AVN_CP_DISJOINT_AABB) ? (generates && p.n_handles) : (flags & AVN_CP_STARTED_TOUCHING) ? false : (flags & AVN_CP_STOPPED_TOUCHING) ? (generates && p.n_handles)
                          : false;
            bool is_push = !(flags & AVN_CP_DISJOINT_AABB) && (((flags & AVN_CP_STARTED_TOUCHING) && generates) || (!(flags & AVN_CP_STARTED_TOUCHING) && !(flags & AVN_CP_STOPPED_TOUCHING) && touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)));
            if (is_pop) { auto it = P.graph.where.find(((uint64_t)cid << 8) | (p.n_handles - 1)); oprecs.push_back({0, it->second.color, (uint32_t)p.b1, (uint32_t)p.b2, it->second.local_index, (uint32_t)P.graph.colors[it->second.color].manifold_handles.size(), (bool)(flags & AVN_CP_STATIC1), (bool)(flags & AVN_CP_STATIC2)}); }
            else if (is_push) oprecs.push_back({1, 255, (uint32_t)p.b1, (uint32_t)p.b2, 0, 0, (bool)(flags & AVN_CP_STATIC1), (bool)(flags & AVN_CP_STATIC2)});
        }
        if (flags & AVN_CP_DISJOINT_AABB) {
            if (generates) while (P.pairs[cid].n_handles) pop(cid);
            removed.push_back(cid);
        } else if (flags & AVN_CP_STARTED_TOUCHING) {
            if (generates) for (uint32_t k = 0; k < c.manifold_count; ++k) push(cid, flags);
        } else if (flags & AVN_CP_STOPPED_TOUCHING) {
            if (generates) while (P.pairs[cid].n_handles) pop(cid);
        } else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) {
            for (uint32_t k = 0; k < c.manifold_count; ++k) push(cid, flags);
        } else if (touching && generates && c.manifold_count_change > 0) {
            for (int32_t k = 0; k < c.manifold_count_change; ++k) push(cid, flags);
        } else if (touching && generates && c.manifold_count_change < 0) {
            for (int32_t k = 0; k < -c.manifold_count_change; ++k) pop(cid);
        }
    }
    P.stats.last_status_changes = (uint32_t)contact_changes.size();
    if (!removed.empty()) {
        st = contact_pairs_remove(removed.data(), removed.size());
        if (st != AVN_OK) return st;
        std::set<uint32_t> gone(removed.begin(), removed.end());
        std::vector<uint32_t> keep;
        for (uint32_t a : P.active) if (!gone.count(a)) keep.push_back(a);
        P.active.swap(keep);
        for (uint32_t cid : removed) { P.pairs.erase(cid); P.free_ids.insert(cid); P.lists.remove_edge(cid); }
        P.stats.pairs_removed += removed.size();
    }
    if (slp) {   // the deferred WakeIslands of the status loop (system_param.rs:391-398), applied before the solver
        slp->isl.flush_wake();
        sleeping_apply(false);
    }
    tm_mark(2);
    st = pipeline_refresh_handles();
    if (st != AVN_OK) return st;
    tm_mark(3);
    P.stats.last_overflow_manifolds = P.offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - P.offsets[AVN_COLOR_OVERFLOW_INDEX];
    diag.narrow_phase_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - np_t0).count();
    if (pipe_stats_on) {
        // dependency-round depth of the op sequence (ops sharing a non-static body are ordered), pops per colour, pops that hit
        // the tail window of their colour's list, level depth of the overflow colour
        std::unordered_map<uint32_t, uint32_t> depth_of_body;
        uint32_t max_depth = 0, n_push = 0, n_pop = 0, pops_c[AVN_GRAPH_COLOR_COUNT] = {0}, win_pops = 0;
        for (const OpRec& o : oprecs) {
            uint32_t d = 0;
            if (!o.s1) d = std::max(d, depth_of_body[o.b1]);
            if (!o.s2) d = std::max(d, depth_of_body[o.b2]);
            ++d;
            if (!o.s1) depth_of_body[o.b1] = d;
            if (!o.s2) depth_of_body[o.b2] = d;
            max_depth = std::max(max_depth, d);
            if (o.push) ++n_push; else { ++n_pop; ++pops_c[o.color]; }
        }
        for (const OpRec& o : oprecs) if (!o.push && o.pos + pops_c[o.color] >= o.len) ++win_pops;
        uint32_t push_depth = 0;
        {
            std::unordered_map<uint32_t, uint32_t> pd;
            for (const OpRec& o : oprecs) {
                if (!o.push) continue;
                uint32_t d = 0;
                if (!o.s1) d = std::max(d, pd[o.b1]);
                if (!o.s2) d = std::max(d, pd[o.b2]);
                ++d;
                if (!o.s1) pd[o.b1] = d;
                if (!o.s2) pd[o.b2] = d;
                push_depth = std::max(push_depth, d);
            }
        }
        std::fprintf(stderr, "[avo pipe] push-only depth %u\n", push_depth);
        uint32_t max_pops = 0; for (uint32_t x : pops_c) max_pops = std::max(max_pops, x);
        std::unordered_map<uint32_t, uint32_t> lvl_of_body; uint32_t levels = 0;
        for (const auto& h : P.graph.colors[AVN_COLOR_OVERFLOW_INDEX].manifold_handles) {
            uint32_t l = 0;
            const bool d1 = h.body1 < bodies.size() && bodies[h.body1].rb_type != AVN_RB_STATIC, d2 = h.body2 < bodies.size() && bodies[h.body2].rb_type != AVN_RB_STATIC;
            if (d1) l = std::max(l, lvl_of_body[h.body1]);
            if (d2) l = std::max(l, lvl_of_body[h.body2]);
            ++l;
            if (d1) lvl_of_body[h.body1] = l;
            if (d2) lvl_of_body[h.body2] = l;
            levels = std::max(levels, l);
        }
        std::fprintf(stderr, "[avo pipe] changes %zu push %u pop %u (overflow pops %u, max/colour %u, tail-window pops %u) op depth %u | manifolds %zu overflow %u levels %u | new pairs %zu removed %zu\n",
                     contact_changes.size(), n_push, n_pop, pops_c[AVN_COLOR_OVERFLOW_INDEX], max_pops, win_pops, max_depth, P.handles.size(), P.stats.last_overflow_manifolds, levels, pairs.size(), removed.size());
    }
    tm_last = std::chrono::steady_clock::now();
    solver();
    tm_mark(4);
    if (pipe_timing_on) std::fprintf(stderr, "[avo timing] broad phase %.0f | pair add %.0f | narrow phase %.0f | status loop + removals %.0f | handle lists %.0f | solver %.0f ms\n", diag.broad_phase_ms, tm_phase[0], tm_phase[1], tm_phase[2], tm_phase[3], tm_phase[4]);
    diag.contact_count = (uint32_t)P.active.size();
    if (slp) { sleeping_systems(); return pipeline_refresh_handles(); }   // (SleepIslands / WakeIslands changed the colour lists: what avn_pipeline_handles_get shows is the state after the step)
    return AVN_OK;
}

// ---- persistent islands + sleeping in the closed loop ------------------------------------------------------------------------------
template <class S> avn_status World<S>::sleeping_enable(const avn_sleep_params* p) {
    if (!p) {   // off: every island awake again
        if (slp) {
            for (size_t b = 0; b < bodies.size(); ++b) if (slp->isl.has_node((uint32_t)b) && slp->isl.body_sleeping[b]) { slp->isl.wake_body((uint32_t)b); sleeping_apply(false); }
            delete slp; slp = nullptr;
        }
        return AVN_OK;
    }
    if (p->struct_size != sizeof(avn_sleep_params)) { error = "sleeping_enable: bad params"; return AVN_ERR_BAD_ARG; }
    if (!pipe) { error = "sleeping_enable: needs the closed loop (avn_pipeline_enable)"; return AVN_ERR_STATE; }
    if (!pipe->pairs.empty()) { error = "sleeping_enable: enable it before the first step of the closed loop"; return AVN_ERR_STATE; }
    delete slp;
    slp = new Sleeping();
    slp->p = *p;
    const size_t n = bodies.size();
    if (p->body_linear_threshold) slp->lin.assign(p->body_linear_threshold, p->body_linear_threshold + n);
    if (p->body_angular_threshold) slp->ang.assign(p->body_angular_threshold, p->body_angular_threshold + n);
    if (p->body_sleeping_disabled) slp->disabled.assign(p->body_sleeping_disabled, p->body_sleeping_disabled + n);
    slp->p.body_linear_threshold = nullptr; slp->p.body_angular_threshold = nullptr; slp->p.body_sleeping_disabled = nullptr;
    slp->timer.assign(n, 0.0f);
    auto node = [&](size_t b) { return bodies[b].rb_type != AVN_RB_STATIC && !(bodies[b].body_flags & AVN_BODY_DISABLED); };   // BodyIslandNode, islands/mod.rs:96-140
    for (size_t b = 0; b < n; ++b) if (node(b)) slp->isl.body_add((uint32_t)b);
    for (const Collider<S>& c : colliders) slp->isl.collider_add(c.entity, node((size_t)c.body) ? (uint32_t)c.body : IslandManager::NONE);
    for (size_t j = 0; j < joints.size(); ++j) slp->isl.joint_add((uint32_t)j, (uint32_t)joints[j].body1, (uint32_t)joints[j].body2);
    for (size_t b = 0; b < n; ++b) if (node(b) && (bodies[b].body_flags & AVN_BODY_SLEEPING)) { slp->isl.sleep_body((uint32_t)b); slp->isl.clear_results(); }   // bodies uploaded asleep
    return AVN_OK;
}
// pops / pushes of SleepIslands / WakeIslands into the ConstraintGraph, pairs into / out of the active set, Sleeping on / off the bodies
template <class S> void World<S>::sleeping_apply(bool count) {
    PipelineState& P = *pipe;
    IslandManager& M = slp->isl;
    for (uint32_t cid : M.popped) {
        PipelineState::Pair& pr = P.pairs[cid];
        if (!pr.n_handles) continue;
        --pr.n_handles;
        P.graph.pop_manifold(((uint64_t)cid << 8) | pr.n_handles);
        P.handles_dirty = true; ++P.stats.manifolds_popped;
    }
    for (uint32_t cid : M.pushed) {
        PipelineState::Pair& pr = P.pairs[cid];
        const uint32_t flags = contact_rows[cid].flags;
        P.graph.push_manifold(((uint64_t)cid << 8) | pr.n_handles, (uint32_t)pr.b1, (uint32_t)pr.b2, flags & AVN_CP_STATIC1, flags & AVN_CP_STATIC2);
        ++pr.n_handles; P.handles_dirty = true; ++P.stats.manifolds_pushed;
    }
    if (!M.pairs_slept.empty()) {   // ContactGraph::sleeping_pairs: out of the narrow phase's iteration
        std::set<uint32_t> gone(M.pairs_slept.begin(), M.pairs_slept.end());
        std::vector<uint32_t> keep;
        for (uint32_t a : P.active) if (!gone.count(a)) keep.push_back(a);
        P.active.swap(keep);
    }
    for (uint32_t cid : M.pairs_woken) P.active.push_back(cid);
    for (uint32_t b : M.bodies_slept) { bodies[b].body_flags |= AVN_BODY_SLEEPING; bodies[b].has_solver_body = false; }
    for (uint32_t b : M.bodies_woken) {
        bodies[b].body_flags &= (uint8_t)~AVN_BODY_SLEEPING;
        bodies[b].has_solver_body = bodies[b].rb_type != AVN_RB_STATIC && bodies[b].active() && !dsh_foreign(b);
        slp->timer[b] = 0.0f;   // sleep_timer.0 = 0.0 (sleeping.rs:492)
    }
    if (count) { slp->last_popped = (uint32_t)M.popped.size(); slp->last_pushed = (uint32_t)M.pushed.size(); }
}
template <class S> void World<S>::sleeping_systems() {
    auto t0 = std::chrono::steady_clock::now();
    Sleeping& Z = *slp;
    const size_t n = bodies.size();
    Z.isl.split_candidate_now();   // split_island, SolverSystems::Finalize (islands/mod.rs:160-178)
    // update_sleeping_states, body side (sleeping.rs:203-223): bodies with a SolverBody, not Sleeping, not SleepingDisabled
    const S length_unit_squared = (S)Z.p.length_unit * (S)Z.p.length_unit;
    std::vector<uint8_t> flags(n, 0);
    Z.n_awake = 0;
    for (size_t b = 0; b < n; ++b) {
        if (!Z.isl.has_node((uint32_t)b)) continue;
        const bool disabled = !Z.disabled.empty() && Z.disabled[b];
        if (disabled) { flags[b] = 2; Z.timer[b] = 0.0f; if (bodies[b].has_solver_body) ++Z.n_awake; continue; }   // wake_islands_with_sleeping_disabled resets the timer
        if (!bodies[b].has_solver_body) continue;   // Sleeping
        ++Z.n_awake;
        flags[b] = 1;
        const SolverBody<S>& sb = bodies[b].sb;
        const S lin_vel_squared = length_squared(sb.linear_velocity), ang_vel_squared = length_squared(sb.angular_velocity);
        const float lt = Z.lin.empty() ? Z.p.linear_threshold : Z.lin[b], at = Z.ang.empty() ? Z.p.angular_threshold : Z.ang[b];
        const float lin_threshold_squared = lt * std::fabs(lt), ang_threshold_squared = at * std::fabs(at);   // "Keep signs."
        if (lin_vel_squared < length_unit_squared * (S)lin_threshold_squared && ang_vel_squared < (S)ang_threshold_squared) Z.timer[b] += Z.p.delta_secs;
        else Z.timer[b] = 0.0f;
    }
    Z.isl.sleeping_systems(Z.timer.data(), flags.data(), (uint32_t)n, Z.p.time_to_sleep);
    Z.last_slept = Z.isl.last_slept; Z.last_woken = Z.isl.last_woken;
    sleeping_apply(true);
    Z.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
template <class S> avn_status World<S>::sleeping_stats_get(avn_sleeping_stats* o) {
    if (!o) return AVN_ERR_BAD_ARG;
    std::memset(o, 0, sizeof *o);
    if (!slp) return AVN_OK;
    slp->isl.stats(&o->islands);
    o->n_awake_bodies = slp->n_awake; o->last_islands_slept = slp->last_slept; o->last_islands_woken = slp->last_woken;
    o->last_manifolds_popped = slp->last_popped; o->last_manifolds_pushed = slp->last_pushed; o->last_host_ms = slp->host_ms;
    return AVN_OK;
}
template <class S> avn_status World<S>::sleeping_state_get(const avn_sleeping_out* o) {
    if (!o) return AVN_ERR_BAD_ARG;
    if (!slp) { error = "sleeping_state_get: sleeping is not enabled"; return AVN_ERR_STATE; }
    const uint32_t n = (uint32_t)bodies.size();
    slp->isl.state(n, o->island, o->next_in_island, nullptr, nullptr);
    for (uint32_t b = 0; b < n; ++b) {
        if (o->sleeping) o->sleeping[b] = (bodies[b].body_flags & AVN_BODY_SLEEPING) ? 1 : 0;
        if (o->sleep_timer) o->sleep_timer[b] = slp->timer[b];
    }
    return AVN_OK;
}
// ---- despawn inside the closed loop (header: avn_despawn) ------------------------------------------------------------------------------
// collision/narrow_phase/mod.rs:399-457 remove_collider (the callback: pops + islands.remove_contact for TOUCHING pairs), :459-560 the observers
// (WakeIslands([the body's island]) queued, then remove_collider per collider); contact_types/contact_graph.rs:641-700 remove_collider_with (every
// edge of the node, callback first, then out of the active / sleeping pairs and the pair set); data_structures/stable_graph.rs:251-283
// remove_node_with (the node's OUTGOING list from its head, then its INCOMING list from its head: newest edge first; the edge id is freed,
// :286-315); dynamics/solver/islands/mod.rs:1336-1400 BodyIslandNode::on_remove.
template <class S> avn_status World<S>::despawn(const avn_despawn_list* d) {
    if (!d || (d->struct_size != sizeof(avn_despawn_list) && d->struct_size != AVN_DESPAWN_LIST_SIZE_R4) || (d->n_colliders && !d->collider_entities) || (d->n_bodies && !d->bodies)) { error = "despawn: bad argument"; return AVN_ERR_BAD_ARG; }
    if (d->struct_size == sizeof(avn_despawn_list) && d->n_joints && !d->joints) { error = "despawn: bad argument"; return AVN_ERR_BAD_ARG; }
    if (!pipe) { error = "despawn: needs the closed loop (avn_pipeline_enable)"; return AVN_ERR_STATE; }
    if (despawn_needs_bodies || despawn_needs_colliders) { error = "despawn: the previous avn_despawn is still waiting for avn_bodies_upload / avn_colliders_upload"; return AVN_ERR_STATE; }
    PipelineState& P = *pipe;
    new_pair_ids.clear();   // (the ABI: after avn_despawn the last step's new-pair ids may name rows that left -- an empty list until the next step; header, avn_pipeline_new_pair_ids_get)
    const size_t n_old = bodies.size();
    std::vector<uint8_t> gone_body(n_old, 0);
    for (uint32_t i = 0; i < d->n_bodies; ++i) {
        const uint32_t b = d->bodies[i];
        if (b >= n_old || gone_body[b]) { error = "despawn: body index out of range or listed twice"; return AVN_ERR_BAD_ARG; }
        gone_body[b] = 1;
    }
    for (uint32_t i = 0; i < d->n_colliders; ++i) if (!collider_slot.count(d->collider_entities[i])) { error = "despawn: unknown collider"; return AVN_ERR_BAD_ARG; }
    const uint32_t n_gone_joints = d->struct_size == sizeof(avn_despawn_list) ? d->n_joints : 0u;
    std::vector<uint8_t> gone_joint(joints.size(), 0);
    for (uint32_t i = 0; i < n_gone_joints; ++i) {
        const uint32_t j = d->joints[i];
        if (j >= gone_joint.size() || gone_joint[j]) { error = "despawn: joint index out of range or listed twice"; return AVN_ERR_BAD_ARG; }
        gone_joint[j] = 1;
    }
    for (size_t j = 0; j < joints.size(); ++j)
        if (!gone_joint[j] && (gone_body[(size_t)joints[j].body1] || gone_body[(size_t)joints[j].body2])) { error = "despawn: a joint names a despawned body: list it in avn_despawn_list::joints"; return AVN_ERR_STATE; }
    std::unordered_set<uint32_t> gone_collider;
    auto remove_collider = [&](uint32_t entity) {
        if (gone_collider.count(entity)) return;   // (ContactGraph::remove_collider_with: the entity has no node any more)
        gone_collider.insert(entity);
        auto nd = P.node_of_collider.find(entity);
        if (nd != P.node_of_collider.end()) {
            for (uint32_t id : P.lists.edges_of(nd->second)) {   // outgoing newest first, then incoming newest first
                PipelineState::Pair& pr = P.pairs[id];
                const bool touching = contact_rows[id].flags & AVN_CP_TOUCHING;
                if (touching) {
                    while (pr.n_handles) { --pr.n_handles; P.graph.pop_manifold(((uint64_t)id << 8) | pr.n_handles); P.handles_dirty = true; ++P.stats.manifolds_popped; }
                }
                if (slp) slp->isl.remove_collider_edge(id);   // (unlinks a touching, linked pair from its island first)
                // out of the active pairs, the pair set, the graph; the id returns to the pool
                contact_pairs_remove(&id, 1);
                P.active.erase(std::remove(P.active.begin(), P.active.end(), id), P.active.end());
                P.lists.remove_edge(id);
                P.pairs.erase(id); P.free_ids.insert(id);
                ++P.stats.pairs_removed;
            }
            P.node_of_collider.erase(nd); ++P.nodes_dropped;
        }
        if (slp) slp->isl.collider_remove(entity);
    };
    auto island_of = [&](int32_t body) -> uint32_t { return slp && body >= 0 && slp->isl.has_node((uint32_t)body) ? slp->isl.body_node[(size_t)body].island_id : IslandManager::NONE; };
    auto wake = [&](uint32_t island) {
        if (!slp || island == IslandManager::NONE) return;
        slp->isl.clear_results();
        slp->isl.wake_islands({island});
        sleeping_apply(false);
    };
    // 0. joints (round 5): remove_joint_from_graph::<Remove, T> per joint, in the order given (joint_graph/plugin.rs:163-194) -- out of its island
    //    (constraints_removed += 1), out of the JointGraph, WakeIslands([island]) when it sleeps; then the joint array closes up
    if (n_gone_joints) {
        for (uint32_t i = 0; i < n_gone_joints; ++i)
            if (slp) {
                slp->isl.clear_results();
                const uint32_t isl = slp->isl.joint_remove(d->joints[i]);
                if (isl != IslandManager::NONE) { slp->isl.wake_islands({isl}); sleeping_apply(false); }
            }
        std::vector<uint32_t> jmap(joints.size(), IslandManager::NONE);
        std::vector<Joint<S>> kept;
        for (size_t j = 0; j < joints.size(); ++j) if (!gone_joint[j]) { jmap[j] = (uint32_t)kept.size(); kept.push_back(joints[j]); }
        joints.swap(kept);
        joint_order.resize(joints.size());
        for (uint32_t i = 0; i < joints.size(); ++i) joint_order[i] = i;
        std::stable_sort(joint_order.begin(), joint_order.end(), [&](uint32_t a, uint32_t b) { return joints[a].type < joints[b].type; });
        if (slp) slp->isl.renumber_joints(jmap);
    }
    // 1. colliders despawned on their own (remove_collider_on::<Remove, ColliderMarker>)
    for (uint32_t i = 0; i < d->n_colliders; ++i) {
        const uint32_t ent = d->collider_entities[i];
        const uint32_t isl = island_of(colliders[collider_slot[ent]].body);
        remove_collider(ent);
        wake(isl);
    }
    // 2. bodies with their colliders (remove_body_on::<Remove, RigidBody>, the colliders' own observers find nothing left), then
    //    BodyIslandNode::on_remove, then the queued WakeIslands
    for (uint32_t i = 0; i < d->n_bodies; ++i) {
        const uint32_t b = d->bodies[i];
        const uint32_t isl = island_of((int32_t)b);
        for (const Collider<S>& c : colliders) if ((uint32_t)c.body == b) remove_collider(c.entity);   // RigidBodyColliders: upload order
        if (slp) slp->isl.body_remove(b);
        wake(isl);
    }
    // 3. stable compaction of the bodies and of everything that names one
    std::vector<uint32_t> new_index(n_old, IslandManager::NONE);
    uint32_t n_new = 0;
    for (size_t b = 0; b < n_old; ++b) if (!gone_body[b]) new_index[b] = n_new++;
    {
        std::vector<Body<S>> nb; std::vector<V3<S>> al, aa;
        nb.reserve(n_new); al.reserve(n_new); aa.reserve(n_new);
        for (size_t b = 0; b < n_old; ++b) if (!gone_body[b]) { nb.push_back(bodies[b]); al.push_back(accel_linear[b]); aa.push_back(accel_angular[b]); }
        bodies.swap(nb); accel_linear.swap(al); accel_angular.swap(aa);
        if (d->n_bodies) { local_acc_linear.clear(); local_acc_angular.clear(); }   // (header: a despawn of bodies drops the local accelerations: the host uploads them again for what remains)
    }
    // colliders: the despawned ones leave; the others keep their relative slot order, intervals are retained in place (broad_phase.rs:230-279)
    {
        std::vector<uint32_t> new_slot(colliders.size(), IslandManager::NONE);
        std::vector<Collider<S>> nc; std::vector<Material> nm;
        for (size_t s_ = 0; s_ < colliders.size(); ++s_) {
            const Collider<S>& c = colliders[s_];
            if (gone_collider.count(c.entity) || gone_body[(size_t)c.body]) continue;
            new_slot[s_] = (uint32_t)nc.size();
            nc.push_back(c); nc.back().body = (int32_t)new_index[(size_t)c.body];
            if (s_ < materials.size()) nm.push_back(materials[s_]);
        }
        std::vector<AabbInterval> kept;
        for (const AabbInterval& iv : intervals) if (new_slot[iv.collider] != IslandManager::NONE) kept.push_back({new_slot[iv.collider], iv.flags});
        colliders.swap(nc); intervals.swap(kept);
        if (!materials.empty()) materials.swap(nm);
        collider_slot.clear();
        for (uint32_t s_ = 0; s_ < colliders.size(); ++s_) collider_slot.emplace(colliders[s_].entity, s_);
    }
    for (Joint<S>& j : joints) { j.body1 = (int32_t)new_index[(size_t)j.body1]; j.body2 = (int32_t)new_index[(size_t)j.body2]; }
    collision_disabled_bodies.clear();
    for (const Joint<S>& j : joints) if (j.collision_disabled) collision_disabled_bodies.insert(pair_key((uint32_t)j.body1, (uint32_t)j.body2));
    for (auto& kv : P.pairs) { kv.second.b1 = (int32_t)new_index[(size_t)kv.second.b1]; kv.second.b2 = (int32_t)new_index[(size_t)kv.second.b2]; }
    for (auto& col : P.graph.colors) {   // GraphColor::body_set is indexed by the body: the bits move with the bodies (a despawned body's bits went with its pops)
        std::vector<bool> ns(n_new, false);
        for (size_t b = 0; b < col.body_set.size() && b < n_old; ++b) if (col.body_set[b] && !gone_body[b]) ns[new_index[b]] = true;
        col.body_set.swap(ns);
        for (auto& h : col.manifold_handles) { h.body1 = new_index[h.body1]; h.body2 = new_index[h.body2]; }
    }
    if (slp) {
        slp->isl.renumber_bodies(new_index, n_new);
        auto compact = [&](auto& v) { if (v.empty()) return; std::remove_reference_t<decltype(v)> o; for (size_t b = 0; b < n_old && b < v.size(); ++b) if (!gone_body[b]) o.push_back(v[b]); v.swap(o); };
        compact(slp->timer); compact(slp->lin); compact(slp->ang); compact(slp->disabled);
    }
    despawn_needs_bodies = d->n_bodies != 0; despawn_needs_colliders = d->n_bodies != 0 || d->n_colliders != 0;
    return pipeline_refresh_handles();
}
template <class S> avn_status World<S>::wake_bodies(const uint32_t* ids, size_t n) {
    if (!slp) { error = "wake_bodies: sleeping is not enabled"; return AVN_ERR_STATE; }
    if (n && !ids) return AVN_ERR_BAD_ARG;
    for (size_t i = 0; i < n; ++i) {
        if (!slp->isl.has_node(ids[i])) { error = "wake_bodies: the body has no island node"; return AVN_ERR_BAD_ARG; }
        slp->isl.wake_body(ids[i]);
        sleeping_apply(false);
    }
    return pipeline_refresh_handles();
}

}  // namespace avo
