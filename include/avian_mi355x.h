/*
 * avian_mi355x.h — C ABI of the MI355X-native physics step for Avian's 3D hot path.
 *
 * This is the drop-in boundary.  The reference (avianphysics/avian, pure Rust)
 * has NO foreign interface for this path: the path is reached through Bevy `Plugin`s
 * (`BroadPhasePlugin`, `IntegratorPlugin`, `SolverPlugin`, `XpbdSolverPlugin`,
 * `SolverBodyPlugin`) whose systems read/write ECS components and resources.  Every entry
 * point below therefore cites the reference *system / resource / component* it replaces; the
 * Rust shim a maintainer would write over this header is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; every pointer is a HOST pointer that is
 *    borrowed for the duration of the call only (the library copies into HBM).
 *  - `scalar_bits` (32|64) selects the reference's `f32`/`f64` cargo feature; every `const void*`
 *    scalar array is `float*` or `double*` accordingly.  Vectors are interleaved xyz (the ECS
 *    table layout of `Vec3` components), quaternions xyzw, symmetric tensors
 *    (m00,m01,m02,m11,m12,m22) (= glam_matrix_extras `SymmetricMat3` field order).
 *  - no panics/exceptions cross the ABI: every call returns `avn_status`; a message is
 *    available from `avn_last_error`.
 *  - one caller thread per world (Bevy's PhysicsSchedule is single-threaded,
 *    reference src/schedule/mod.rs:90).
 *
 * The SAME header is implemented twice: by the product (HIP, prefix `avn_`, libavian_mi355x.so)
 * and by the CPU oracle (test infrastructure, prefix `avo_`, oracle/liboracle.so).  Define
 * AVN_PREFIX_ORACLE before including to get the `avo_` names.
 */
#ifndef AVIAN_MI355X_H
#define AVIAN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef AVN_PREFIX_ORACLE
#define AVN_FN(name) avo_##name
#else
#define AVN_FN(name) avn_##name
#endif

#if defined(__GNUC__)
#define AVN_API __attribute__((visibility("default")))
#else
#define AVN_API
#endif

typedef int32_t avn_status;
enum {
    AVN_OK = 0,
    AVN_ERR_BAD_ARG = 1,   /* null pointer, bad size, bad enum */
    AVN_ERR_HIP = 2,       /* a HIP runtime call failed (message has the hipError name) */
    AVN_ERR_OOM = 3,       /* host or device allocation failed */
    AVN_ERR_CAPACITY = 4,  /* fixed capacity exceeded (e.g. pair buffer) */
    AVN_ERR_NO_DEVICE = 5, /* no gfx950 device visible: the product NEVER falls back to CPU */
    AVN_ERR_STATE = 6      /* call order violated (e.g. step before bodies upload) */
};

typedef struct avn_world avn_world; /* opaque */

/* ---- constants mirrored from the reference ------------------------------------------------ */
/* constraint_graph.rs:39-48 */
#define AVN_GRAPH_COLOR_COUNT 24
#define AVN_COLOR_OVERFLOW_INDEX 23
#define AVN_DYNAMIC_COLOR_COUNT 20
#define AVN_MAX_MANIFOLD_POINTS 4 /* 3D manifolds are pruned to <= 4 points, system_param.rs:761-763 */

/* RigidBody (dynamics/rigid_body/mod.rs) */
enum { AVN_RB_DYNAMIC = 0, AVN_RB_STATIC = 1, AVN_RB_KINEMATIC = 2 };
/* body_flags */
enum { AVN_BODY_SLEEPING = 1, AVN_BODY_DISABLED = 2, AVN_BODY_CUSTOM_VELOCITY_INTEGRATION = 4,
       AVN_BODY_CUSTOM_POSITION_INTEGRATION = 8 };
/* LockedAxes bits (dynamics/rigid_body/locked_axes.rs:36-40): 0b100_000 = translation X ... 0b000_001 = rotation Z */
/* SolverBodyFlags (solver_body/mod.rs:132-157): bits 0-5 locked axes, 6 kinematic, 7 gyroscopic */
enum { AVN_SB_KINEMATIC = 1u << 6, AVN_SB_GYROSCOPIC = 1u << 7 };
/* AabbIntervalFlags (broad_phase.rs:190-201) */
enum { AVN_AABB_IS_INACTIVE = 1, AVN_AABB_CONTACT_EVENTS = 2, AVN_AABB_GENERATE_CONSTRAINTS = 4,
       AVN_AABB_CUSTOM_FILTER = 8, AVN_AABB_MODIFY_CONTACTS = 16 };
/* collider_flags given by the host (sources of the interval flags, broad_phase.rs:214-280) */
enum { AVN_COLLIDER_SENSOR = 1, AVN_COLLIDER_EVENTS = 2, AVN_COLLIDER_FILTER_PAIRS = 4,
       AVN_COLLIDER_MODIFY_CONTACTS = 8, AVN_COLLIDER_SWEPT_CCD = 16 };
enum { AVN_SHAPE_CUBOID = 0, AVN_SHAPE_BALL = 1,
       AVN_SHAPE_HOST = 2 /* any other AnyCollider: its aabb / contact_manifolds are the host's, through avn_host_shapes_set */ };
/* manifold_flags */
enum { AVN_MANIFOLD_GENERATES_CONSTRAINTS = 1 };
/* pair flags returned by the broad phase (ContactEdgeFlags/ContactPairFlags set at broad_phase.rs:443-468) */
enum { AVN_PAIR_CONTACT_EVENTS = 1, AVN_PAIR_MODIFY_CONTACTS = 2, AVN_PAIR_GENERATE_CONSTRAINTS = 4,
       AVN_PAIR_NEEDS_CUSTOM_FILTER = 8 /* host must still run CollisionHooks::filter_pairs */ };

/* ---- configuration = the ECS *resources* the systems read --------------------------------- */
typedef struct avn_config {
    uint32_t struct_size;   /* = sizeof(avn_config), for forward compatibility */
    uint32_t scalar_bits;   /* 32 | 64  (cargo features f32 / f64) */
    int32_t device;         /* HIP device ordinal (ignored by the oracle) */
    uint32_t substeps;      /* SubstepCount, solver/schedule.rs:185-191 (default 6) */
    uint64_t dt_ns;         /* Time<Physics>::delta() as integer nanoseconds (Duration) */
    double gravity[3];      /* Gravity, integrator/mod.rs:156-162 (default 0,-9.81,0) */
    double length_unit;     /* PhysicsLengthUnit, solver/plugin.rs:199-207 */
    /* SolverConfig, solver/plugin.rs:216-302 */
    double contact_damping_ratio;    /* 10 */
    double contact_frequency_factor; /* 1.5 */
    double max_overlap_solve_speed;  /* 4 */
    double warm_start_coefficient;   /* 1 */
    double restitution_threshold;    /* 1 */
    uint32_t restitution_iterations; /* 1 */
    /* NarrowPhaseConfig, narrow_phase/mod.rs:203-255 */
    uint32_t match_contacts;            /* 1: warm starting enabled */
    double default_speculative_margin;  /* Scalar::MAX  (pass a value >= FLT_MAX for "unbounded") */
    double contact_tolerance;           /* 0.005 */
    /* Declared extension (NOT in the reference, SURVEY.md header note 2): outer repeats of the
       biased solve / relax / joint pass inside one substep.  1 = reference behaviour. */
    uint32_t solver_iterations;
    uint32_t use_graph; /* product only: replay the substep loop from a captured hipGraph */
} avn_config;

/* ---- rigid bodies = the per-entity components read by prepare_solver_bodies (a3),
 *      pre_process_velocity_increments (a4), writeback_solver_bodies (a9) ------------------- */
typedef struct avn_bodies {
    uint32_t count;
    const void* position;          /* [3n] Position */
    const void* rotation;          /* [4n] Rotation (xyzw) */
    const void* linear_velocity;   /* [3n] LinearVelocity */
    const void* angular_velocity;  /* [3n] AngularVelocity */
    const void* inv_mass;          /* [n]  ComputedMass::inverse() */
    const void* inv_inertia_local; /* [6n] ComputedAngularInertia::inverse() (local space) */
    const void* center_of_mass;    /* [3n] ComputedCenterOfMass; NULL = zero */
    const void* linear_damping;    /* [n]  LinearDamping;  NULL = 0 */
    const void* angular_damping;   /* [n]  AngularDamping; NULL = 0 */
    const void* gravity_scale;     /* [n]  GravityScale;   NULL = 1 */
    const void* accel_linear;      /* [3n] VelocityIntegrationData.linear_increment as left by ForcePlugin
                                           (an acceleration, integrator/mod.rs:219-221); NULL = 0 */
    const void* accel_angular;     /* [3n] likewise angular; NULL = 0 */
    const void* max_linear_speed;  /* [n]  MaxLinearSpeed;  NULL or <0 = absent */
    const void* max_angular_speed; /* [n]  MaxAngularSpeed; NULL or <0 = absent */
    const uint8_t* rb_type;        /* [n]  AVN_RB_* */
    const uint8_t* locked_axes;    /* [n]  LockedAxes::to_bits(); NULL = 0 */
    const int8_t* dominance;       /* [n]  Dominance; NULL = 0 */
    const uint8_t* body_flags;     /* [n]  AVN_BODY_*; NULL = 0 */
} avn_bodies;

typedef struct avn_bodies_out { /* any pointer may be NULL = not wanted */
    void* position;         /* [3n] */
    void* rotation;         /* [4n] */
    void* linear_velocity;  /* [3n] */
    void* angular_velocity; /* [3n] */
} avn_bodies_out;

/* SolverBody / SolverBodyInertia / VelocityIntegrationData (a1, a2, a4) — inspection only */
typedef struct avn_solver_bodies_out {
    void* linear_velocity;  /* [3n] */
    void* angular_velocity; /* [3n] */
    void* delta_position;   /* [3n] */
    void* delta_rotation;   /* [4n] */
    uint32_t* flags;        /* [n] SolverBodyFlags; bit 31 set = body has NO SolverBody (static/sleeping/disabled) */
    void* inv_mass;         /* [n] */
    void* inv_inertia_world;/* [6n] effective_inv_angular_inertia */
    int16_t* dominance;     /* [n] */
    void* linear_increment; /* [3n] */
    void* angular_increment;/* [3n] */
    void* linear_damping_rhs;  /* [n] */
    void* angular_damping_rhs; /* [n] */
} avn_solver_bodies_out;

/* ---- contact manifolds = what prepare_contact_constraints reads from ContactGraph +
 *      ConstraintGraph (solver/plugin.rs:363-448).  Manifolds are given colour-major:
 *      colour c owns [color_offsets[c], color_offsets[c+1]) in `manifold_handles` order. ------ */
typedef struct avn_manifolds {
    uint32_t count;                 /* M */
    const uint32_t* color_offsets;  /* [AVN_GRAPH_COLOR_COUNT + 1] */
    const int32_t* body1;           /* [M] index into the body table (ContactPair::body1) */
    const int32_t* body2;           /* [M] */
    const void* normal;             /* [3M] ContactManifold::normal */
    const void* friction;           /* [M] */
    const void* restitution;        /* [M] */
    const void* tangent_velocity;   /* [3M]; NULL = zero */
    const uint8_t* point_count;     /* [M] 0..4 */
    const uint8_t* manifold_flags;  /* [M] AVN_MANIFOLD_*; NULL = GENERATES_CONSTRAINTS */
    /* contact points, slot = 4*m + p (ContactPoint, contact_types/mod.rs:603-660) */
    const void* anchor1;            /* [3*4M] relative to centre of mass of body1, world space */
    const void* anchor2;            /* [3*4M] */
    const void* penetration;        /* [4M] */
    const void* normal_speed;       /* [4M] */
    const void* warm_start_normal_impulse;  /* [4M];   NULL = zero */
    const void* warm_start_tangent_impulse; /* [2*4M]; NULL = zero */
} avn_manifolds;

/* store_contact_impulses output (solver/plugin.rs:744-749), slot = 4*m + p */
typedef struct avn_impulses_out {
    void* warm_start_normal_impulse;  /* [4M] */
    void* warm_start_tangent_impulse; /* [2*4M] */
    void* normal_impulse;             /* [4M] (= total_impulse) */
} avn_impulses_out;

/* ContactConstraint / ContactConstraintPoint (contact/mod.rs:32-106) — inspection only */
typedef struct avn_constraints_out {
    uint8_t* point_count;     /* [M] 0 = constraint absent (skipped by prepare) */
    int16_t* relative_dominance; /* [M] */
    void* tangent1;           /* [3M] */
    void* anchor1;            /* [3*4M] (copied through) */
    void* initial_separation; /* [4M] */
    void* normal_impulse;     /* [4M] ContactNormalPart::impulse */
    void* total_impulse;      /* [4M] */
    void* normal_effective_mass; /* [4M] */
    void* tangent_impulse;    /* [2*4M] */
    void* tangent_effective_inverse_mass; /* [3*4M] */
    uint8_t* softness_non_dynamic; /* [M] 1 = non_dynamic coefficients */
} avn_constraints_out;

/* ---- XPBD DistanceJoint (dynamics/joints/distance.rs:26-39 + xpbd/joints/distance.rs) ------ */
typedef struct avn_distance_joints {
    uint32_t count;
    const int32_t* body1;          /* [J] */
    const int32_t* body2;          /* [J] */
    const void* local_anchor1;     /* [3J] JointAnchor::Local */
    const void* local_anchor2;     /* [3J] */
    const void* limit_min;         /* [J] DistanceLimit */
    const void* limit_max;         /* [J] */
    const void* compliance;        /* [J] */
    const void* damping_linear;    /* [J] JointDamping; NULL = no JointDamping component */
    const void* damping_angular;   /* [J] */
    const uint8_t* collision_disabled; /* [J] JointCollisionDisabled; NULL = 0 */
} avn_distance_joints;

typedef struct avn_joints_out {
    void* world_r1;          /* [3J] PointConstraintShared / Distance / Prismatic solver data */
    void* world_r2;          /* [3J] */
    void* center_difference; /* [3J] */
    void* total_lagrange;    /* [3J] XpbdConstraintSolverData::total_position_lagrange */
    void* force;             /* [3J] JointForces::force after writeback_joint_forces */
    void* total_rotation_lagrange; /* [3J] total_rotation_lagrange (align+limit / swing+twist / angle); NULL = skip */
    void* torque;            /* [3J] JointForces::torque; NULL = skip */
} avn_joints_out;

/* ---- all XPBD joint types (dynamics/joints/{fixed,revolute,spherical,prismatic,distance}.rs +
 *      solver/xpbd/joints/<type>.rs).  The type ids are the reference's SOLVE ORDER (xpbd/plugin.rs:77-82): all fixed joints
 *      in array order, then revolute, spherical, prismatic, distance.  Local frames are given resolved (JointFrame with
 *      JointAnchor::Local / JointBasis::Local, i.e. after the reference's joint-frame system). ------------------------- */
enum { AVN_JOINT_FIXED = 0, AVN_JOINT_REVOLUTE = 1, AVN_JOINT_SPHERICAL = 2, AVN_JOINT_PRISMATIC = 3, AVN_JOINT_DISTANCE = 4,
       AVN_JOINT_TYPE_COUNT = 5 };
enum { AVN_JOINT_HAS_LIMIT1 = 1, /* revolute angle_limit / spherical swing_limit / prismatic limits are Some(..) */
       AVN_JOINT_HAS_LIMIT2 = 2  /* spherical twist_limit is Some(..) */ };
typedef struct avn_joints {
    uint32_t count;
    const uint8_t* joint_type;     /* [J] AVN_JOINT_* */
    const int32_t* body1;          /* [J] */
    const int32_t* body2;          /* [J] */
    const void* local_anchor1;     /* [3J] */
    const void* local_anchor2;     /* [3J] */
    const void* local_basis1;      /* [4J] quaternion xyzw; NULL = identity */
    const void* local_basis2;      /* [4J] */
    const void* axis;              /* [3J] hinge_axis (revolute, default Z) / twist_axis (spherical, Y) / slider_axis (prismatic, X) */
    const void* limit_min;         /* [J] DistanceLimit (distance, prismatic) or AngleLimit (revolute angle, spherical swing) */
    const void* limit_max;         /* [J] */
    const void* limit2_min;        /* [J] spherical twist_limit; NULL = 0 */
    const void* limit2_max;        /* [J] */
    const uint8_t* limit_flags;    /* [J] AVN_JOINT_HAS_LIMIT*; NULL = 0 (distance joints always use their limit) */
    const void* compliance;        /* [3J] fixed: (point, angle, -)   revolute: (point, align, limit)   spherical: (point, swing, twist)
                                           prismatic: (align, angle, limit)   distance: (compliance, -, -) */
    const void* damping_linear;    /* [J] JointDamping; NULL = no JointDamping component on any joint */
    const void* damping_angular;   /* [J] */
    const uint8_t* collision_disabled; /* [J] NULL = 0 */
} avn_joints;

/* ---- colliders for the broad phase (a24-a27).  A collider sits on its rigid body entity unless
 *      avn_collider_transforms_upload (below) says it is a child of it. ------------------------ */
typedef struct avn_colliders {
    uint32_t count;                 /* C */
    const uint32_t* entity_index;   /* [C] Entity::index() used for PairKey; must be unique */
    const int32_t* body;            /* [C] ColliderOf.body as body-table index */
    const uint8_t* shape;           /* [C] AVN_SHAPE_* */
    const void* half_extents;       /* [3C] cuboid half extents; ball: radius in x */
    const uint32_t* memberships;    /* [C] CollisionLayers; NULL = default (1) */
    const uint32_t* filters;        /* [C] NULL = default (0xFFFFFFFF) */
    const uint8_t* collider_flags;  /* [C] AVN_COLLIDER_*; NULL = 0 */
    const void* collision_margin;   /* [C] NULL = 0 */
    const void* speculative_margin; /* [C] NULL or <0 = use config default */
} avn_colliders;

/* ---- child colliders (round 6): colliders on CHILD entities of their rigid body (compound bodies) -------------------------------------
 *      Avian keeps a ColliderTransform per collider (collision/collider/collider_transform/mod.rs:20-27: translation, rotation, scale relative to the
 *      rigid body) and, first thing in every step, update_child_collider_position (collider_transform/plugin.rs:62-91) sets the child's
 *          Position = rb_pos + rb_rot * translation,   Rotation = (rb_rot * rotation).normalize().
 *      The device computes exactly that wherever a collider's pose is read -- update_aabb (collider/backend.rs:498-624: for a child the swept box uses the
 *      body's velocity AT THE COLLIDER'S OFFSET from the centre of mass, :569-586), update_contacts (narrow_phase/system_param.rs:540-575: the manifold is
 *      computed at the colliders' poses, anchors are shifted by collider_offset = collider.position - body.position), the host-shape queries -- so a
 *      compound body needs no per-step upload: one entry per collider, after every avn_colliders_upload (which puts all colliders back on their bodies).
 *      `scale` is the host's: half_extents / radius are uploaded scaled (Collider::shape_scaled), `translation` is ColliderTransform::translation as
 *      propagate_collider_transforms leaves it (already multiplied by the parents' scale).  Mass properties of the compound (ComputedMass,
 *      ComputedAngularInertia, ComputedCenterOfMass) are the host's as for any body. */
typedef struct avn_collider_transforms {
    uint32_t count;               /* C of the last avn_colliders_upload */
    const uint8_t* is_child;      /* [C] 1: the collider is a child entity of its body (ColliderTransform applies); 0: it sits on the body's entity */
    const void* translation;      /* [3C] ColliderTransform::translation */
    const void* rotation;         /* [4C] ColliderTransform::rotation (xyzw) */
} avn_collider_transforms;
AVN_API avn_status AVN_FN(collider_transforms_upload)(avn_world* w, const avn_collider_transforms* t);   /* NULL or count 0: no collider is a child */

typedef struct avn_pair {
    uint32_t collider1; /* Entity::index() of the collider earlier in sorted order (broad_phase.rs:443) */
    uint32_t collider2;
    int32_t body1;      /* body-table index */
    int32_t body2;
    uint32_t flags;     /* AVN_PAIR_* */
    uint32_t reserved;
} avn_pair;

/* ---- narrow phase, part 1: contact_query::contact_manifolds
 *      (collision/collider/parry/contact_query.rs:156-261, a public function of the reference) for a BATCH of shape pairs.
 *      Shapes are parry3d Ball / Cuboid; a convex pair yields at most ONE manifold.  Points are returned as the reference
 *      returns them (before the narrow phase prunes them to 4): anchors in world orientation relative to the collider
 *      origins, the world-space midpoint, the penetration (negative = separated but within prediction_distance) and
 *      parry's PackedFeatureIds (collision/contact_types/feature_id.rs).  parry3d / nalgebra are third-party and not
 *      vendored in the reference tree: their part is implemented from the published algorithm (parity unpinned). ------ */
#define AVN_MAX_QUERY_POINTS 16
typedef struct avn_shape_pairs {
    uint32_t count;
    const uint8_t* shape1;        /* [n] AVN_SHAPE_* */
    const void* half_extents1;    /* [3n] cuboid half extents; ball: radius in x */
    const void* position1;        /* [3n] collider Position */
    const void* rotation1;        /* [4n] collider Rotation (xyzw) */
    const uint8_t* shape2;
    const void* half_extents2;
    const void* position2;
    const void* rotation2;
    const void* prediction_distance; /* [n] */
} avn_shape_pairs;
typedef struct avn_query_manifolds_out { /* slot = AVN_MAX_QUERY_POINTS * pair + point; any pointer may be NULL */
    uint8_t* point_count;   /* [n] 0 = no manifold */
    void* normal;           /* [3n] world space, from shape 1 towards shape 2 */
    void* anchor1;          /* [3 * 16n] */
    void* anchor2;          /* [3 * 16n] */
    void* point;            /* [3 * 16n] */
    void* penetration;      /* [16n] */
    uint32_t* feature_id1;  /* [16n] */
    uint32_t* feature_id2;  /* [16n] */
} avn_query_manifolds_out;

/* ---- host shapes (round 6): AnyCollider's two methods as callbacks ------------------------------------------------------------------
 *      The device narrow phase holds parry's Ball and Cuboid.  Every other collider (capsule, cylinder, cone, convex hull, a custom
 *      AnyCollider ...) is uploaded with shape = AVN_SHAPE_HOST and keeps exactly the two pieces of the reference that depend on the shape on
 *      the host -- the extension points Avian itself defines for custom colliders (collision/collider/mod.rs: AnyCollider):
 *        aabb_with_context / swept_aabb_with_context   called by update_aabb (collision/collider/backend.rs:498-624) for the start pose and, when the
 *                                                      speculative margin is positive, the predicted end pose: ONE box per collider comes back, and the
 *                                                      device grows it by contact_tolerance + collision margin as it does for its own shapes;
 *        contact_manifolds_with_context                called by update_contacts (collision/narrow_phase/system_param.rs:700-712) with the colliders'
 *                                                      poses and max_contact_distance: the manifold comes back as contact_query::contact_manifolds
 *                                                      returns it (contact_query.rs:156-261: normal, anchor1 relative to collider 1's position in world
 *                                                      orientation, penetration, feature ids; anchor2 = anchor1 + (position1 - position2) is the device's).
 *      EVERYTHING ELSE of update_contacts stays on the device for those pairs too: layers / AABB test, flags, margins, the speculative filter,
 *      prune_points, match_contacts (the warm-start impulses never leave HBM), normal_speed, the status change.  Per step the bus carries
 *      88 / 168 B (f32 / f64: query out + box back) per host-shaped COLLIDER and 76 / 136 B out + 476 / 808 B back per PAIR that involves one -- nothing for the rest of the
 *      world (VERDICT r5 "HostNarrowPhase without re-sending the world").  Convex shapes: one manifold per pair (manifolds[0]); a composite shape's
 *      further manifolds are out of scope as they are for the device shapes.  The callbacks run on the thread that calls avn_step /
 *      avn_run_system, between two stream synchronisations; `n` queries per call, answered in place.  scalar_bits = 32 / 64 selects the _f32 / _f64
 *      record types.  Works in the device closed loop (avn_pipeline_enable) and in the host-bookkeeping mode (AVN_SYS_UPDATE_AABB / AVN_SYS_NARROW_PHASE). */
typedef struct avn_host_aabb_query_f32 {
    uint32_t collider;          /* Entity::index() */
    uint32_t swept;             /* 0: aabb_with_context(start); 1: swept_aabb_with_context(start, end) */
    float start_position[3], start_rotation[4], end_position[3], end_rotation[4];
} avn_host_aabb_query_f32;
typedef struct avn_host_aabb_query_f64 {
    uint32_t collider;
    uint32_t swept;
    double start_position[3], start_rotation[4], end_position[3], end_rotation[4];
} avn_host_aabb_query_f64;
typedef struct avn_host_aabb_f32 { float min[3], max[3]; } avn_host_aabb_f32;
typedef struct avn_host_aabb_f64 { double min[3], max[3]; } avn_host_aabb_f64;
typedef struct avn_host_manifold_query_f32 {
    uint32_t contact_id, collider1, collider2 /* Entity::index() */, reserved;
    float position1[3], rotation1[4], position2[3], rotation2[4];
    float max_contact_distance;
} avn_host_manifold_query_f32;
typedef struct avn_host_manifold_query_f64 {
    uint32_t contact_id, collider1, collider2, reserved;
    double position1[3], rotation1[4], position2[3], rotation2[4];
    double max_contact_distance;
} avn_host_manifold_query_f64;
typedef struct avn_host_manifold_f32 {
    uint32_t point_count;       /* 0: no manifold; at most AVN_MAX_QUERY_POINTS */
    float normal[3];            /* ContactManifold::normal (world space, unit, from collider 1 towards collider 2) */
    float anchor1[48];          /* [3 * 16] ContactPoint::anchor1 as contact_manifolds returns it */
    float penetration[16];
    uint32_t feature_id1[16], feature_id2[16];   /* PackedFeatureId bits; 0 = unknown */
} avn_host_manifold_f32;
typedef struct avn_host_manifold_f64 {
    uint32_t point_count;
    uint32_t reserved;
    double normal[3];
    double anchor1[48];
    double penetration[16];
    uint32_t feature_id1[16], feature_id2[16];
} avn_host_manifold_f64;
typedef void (*avn_host_aabb_fn)(void* user, uint32_t scalar_bits, uint32_t n, const void* queries /* avn_host_aabb_query_fNN[n] */, void* aabbs_out /* avn_host_aabb_fNN[n] */);
typedef void (*avn_host_manifolds_fn)(void* user, uint32_t scalar_bits, uint32_t n, const void* queries /* avn_host_manifold_query_fNN[n], ascending contact_id */, void* manifolds_out /* avn_host_manifold_fNN[n] */);
/* NULL callbacks unregister.  A world that holds an AVN_SHAPE_HOST collider and no callbacks fails avn_step / the two systems with AVN_ERR_STATE. */
AVN_API avn_status AVN_FN(host_shapes_set)(avn_world* w, avn_host_aabb_fn aabb, avn_host_manifolds_fn manifolds, void* user);
typedef struct avn_host_shape_stats { uint32_t host_colliders, last_aabb_queries, last_manifold_queries, last_manifolds_with_points; uint64_t bytes_to_host, bytes_from_host; double last_callback_ms; } avn_host_shape_stats;
AVN_API avn_status AVN_FN(host_shape_stats_get)(avn_world* w, avn_host_shape_stats* out);

/* ---- collision hooks (round 6): CollisionHooks::filter_pairs / modify_contacts as callbacks ---------------------------------------------
 *      The reference's two user hooks (collision/hooks.rs:137-231) run only for colliders that carry ActiveCollisionHooks
 *      (AVN_COLLIDER_FILTER_PAIRS / AVN_COLLIDER_MODIFY_CONTACTS in avn_colliders::collider_flags).  Until round 6 a world with such a collider had to leave the
 *      closed loop (HostNarrowPhase mode: every manifold re-sent every step).  With the callbacks registered it stays; per step the bus carries 12 B per
 *      pair the filter is asked about and 232 / 392 B (f32 / f64) each way per TOUCHING pair whose contacts a hook may modify -- nothing for the rest of the world.
 *        filter_pairs     broad_phase.rs:431-439: asked for every NEW candidate pair (AABBs intersect, layers interact, not yet in the ContactGraph, no
 *                         joint disables it) one of whose colliders has FILTER_PAIRS, in the sweep's emission order, before the pair gets its ContactId.  A
 *                         rejected pair does not enter the graph, so it is asked about again every step while the AABBs overlap -- as in the reference.
 *                         should_collide[] is preset to 1.
 *        modify_contacts  narrow_phase/system_param.rs:770-778: called for every pair flagged MODIFY_CONTACTS (either collider had the flag when the pair was
 *                         created) that has a manifold after the speculative filter and prune_points, BEFORE match_contacts and the status change:
 *                         the record holds the ContactPair as the hook sees it (one manifold: convex pairs) and is modified IN PLACE;
 *                         touching = the hook's return value (0: manifolds.clear(), the pair stops / does not start touching).  Everything after the hook --
 *                         match_contacts against the previous step's points (warm-start impulses never leave HBM), manifold_count_change, the status
 *                         change -- runs on the device from the returned record.  The callback gets its records in ascending contact_id.
 *      NULL for either hook = the trait's default (true / unmodified).  The callbacks run on the thread that calls avn_step / avn_run_system, between two
 *      stream synchronisations.  Works in the device closed loop, in the host-bookkeeping loop and in AVN_SYS_BROAD_PHASE / AVN_SYS_NARROW_PHASE
 *      (with a filter registered, avn_pairs_get returns the pairs that passed; AVN_PAIR_NEEDS_CUSTOM_FILTER stays set on them as information). */
typedef struct avn_hook_pair { uint32_t index /* place in the step's emission order */, collider1, collider2 /* Entity::index() */; } avn_hook_pair;
typedef struct avn_hook_contact_f32 {
    uint32_t contact_id, collider1, collider2; /* Entity::index() */
    uint32_t body1, body2;      /* body-table indices (ContactPair::body1 / body2) */
    uint32_t flags;             /* AVN_CP_* as update_contacts has set them when the hook runs: STATIC1/2, GENERATE_CONSTRAINTS current, TOUCHING still the previous step's */
    uint32_t touching;          /* in: 1.  out: the hook's return value */
    uint32_t manifold_count;    /* in: 1.  out: 0 if the hook emptied ContactPair::manifolds, else 1 */
    uint32_t point_count;       /* in: 1..4.  out: 0..4 (points [0, point_count) are kept, in this order) */
    uint32_t reserved;
    float normal[3];            /* ContactManifold::normal */
    float friction, restitution;
    float tangent_velocity[3];  /* in: 0 (system_param.rs:722-729) */
    float anchor1[12], anchor2[12];   /* [3 * 4] relative to the bodies' centres of mass, world orientation */
    float penetration[4], normal_speed[4];
    uint32_t feature_id1[4], feature_id2[4];
} avn_hook_contact_f32;
typedef struct avn_hook_contact_f64 {
    uint32_t contact_id, collider1, collider2, body1, body2, flags, touching, manifold_count, point_count, reserved;
    double normal[3];
    double friction, restitution;
    double tangent_velocity[3];
    double anchor1[12], anchor2[12];
    double penetration[4], normal_speed[4];
    uint32_t feature_id1[4], feature_id2[4];
} avn_hook_contact_f64;
typedef void (*avn_filter_pairs_fn)(void* user, uint32_t n, const avn_hook_pair* pairs /* emission order */, uint8_t* should_collide /* [n], preset to 1 */);
typedef void (*avn_modify_contacts_fn)(void* user, uint32_t scalar_bits, uint32_t n, void* contacts /* avn_hook_contact_fNN[n], ascending contact_id, in place */);
AVN_API avn_status AVN_FN(collision_hooks_set)(avn_world* w, avn_filter_pairs_fn filter, avn_modify_contacts_fn modify, void* user);
typedef struct avn_collision_hook_stats { uint32_t last_filter_queries, last_filter_rejected, last_modify_queries, last_modify_rejected; uint64_t bytes_to_host, bytes_from_host; double last_callback_ms; } avn_collision_hook_stats;
AVN_API avn_status AVN_FN(collision_hook_stats_get)(avn_world* w, avn_collision_hook_stats* out);

/* ---- narrow phase, part 2: the ContactGraph side kept on device ----------------------------
 *      NarrowPhase::update_contacts (collision/narrow_phase/system_param.rs:437-830) runs as AVN_SYS_NARROW_PHASE over a
 *      device-resident table of contact pairs indexed by the reference's ContactId (contact_graph.rs, id_pool.rs); the
 *      manifolds it produces stay in HBM and feed prepare_contact_constraints through a colour-major list of handles
 *      (= GraphColor::manifold_handles, constraint_graph.rs:66-80), so that per step only STATUS CHANGES cross the bus.
 *      The host keeps what the reference keeps in host structures: ContactId allocation, the ConstraintGraph, events.
 *      Convex pairs only (Ball / Cuboid): one manifold per pair, manifold index 0.  CollisionHooks::modify_contacts is called through
 *      avn_collision_hooks_set's callback when one is registered; without one, pairs flagged AVN_PAIR_MODIFY_CONTACTS are processed unmodified. */
/* ContactPairFlags (contact_types/mod.rs) as stored per table row; bits 8.. are this step's status outputs */
enum { AVN_CP_TOUCHING = 1, AVN_CP_GENERATE_CONSTRAINTS = 2, AVN_CP_STATIC1 = 4, AVN_CP_STATIC2 = 8, AVN_CP_MODIFY_CONTACTS = 16,
       AVN_CP_CONTACT_EVENTS = 32,
       AVN_CP_DISJOINT_AABB = 1 << 8, AVN_CP_STARTED_TOUCHING = 1 << 9, AVN_CP_STOPPED_TOUCHING = 1 << 10,
       AVN_CP_STARTED_GENERATING_CONSTRAINTS = 1 << 11 };
/* CoefficientCombine (dynamics/rigid_body/physics_material.rs:13-24) */
enum { AVN_COMBINE_AVERAGE = 1, AVN_COMBINE_GEOMETRIC_MEAN = 2, AVN_COMBINE_MIN = 3, AVN_COMBINE_MULTIPLY = 4, AVN_COMBINE_MAX = 5 };
/* Friction / Restitution per collider, RESOLVED by the host (collider component, else the body's, else the Default*
 * resource: system_param.rs:596-620), in collider upload order */
typedef struct avn_collider_materials {
    uint32_t count;                     /* must equal the collider count */
    const void* friction;               /* [C] dynamic_coefficient; NULL = 0.5 */
    const void* restitution;            /* [C] coefficient; NULL = 0.0 */
    const uint8_t* friction_combine;    /* [C] AVN_COMBINE_*; NULL = AVERAGE */
    const uint8_t* restitution_combine; /* [C] NULL = AVERAGE */
} avn_collider_materials;
/* ContactGraph::add_edge_and_key_with (contact_types/contact_graph.rs:521-566): new rows, not touching, no manifolds */
typedef struct avn_contact_pairs {
    uint32_t count;
    const uint32_t* contact_id;  /* [n] ContactId chosen by the host (lowest free, id_pool.rs:31-40) */
    const uint32_t* collider1;   /* [n] Entity::index() */
    const uint32_t* collider2;
    const uint32_t* pair_flags;  /* [n] AVN_PAIR_* as returned by the broad phase */
} avn_contact_pairs;
/* one entry per contact pair whose ContactStatusBits bit is set after AVN_SYS_NARROW_PHASE, ascending contact_id
 * (the order system_param.rs:141-145 processes them in) */
typedef struct avn_contact_change {
    uint32_t contact_id;
    uint32_t flags;                 /* AVN_CP_* */
    int32_t manifold_count_change;  /* ContactPair::manifold_count_change */
    uint32_t manifold_count;        /* manifolds the pair has now (0 | 1) */
} avn_contact_change;
/* inspection of table rows (ContactPair::manifolds[0] + flags), slot = 4 * i + p */
typedef struct avn_contacts_out {
    uint32_t* flags;          /* [n] AVN_CP_* */
    uint8_t* point_count;     /* [n] */
    void* normal;             /* [3n] */
    void* friction;           /* [n] */
    void* restitution;        /* [n] */
    void* anchor1;            /* [3*4n] relative to the centre of mass of body1 */
    void* anchor2;            /* [3*4n] */
    void* penetration;        /* [4n] */
    void* normal_speed;       /* [4n] */
    void* warm_start_normal_impulse;   /* [4n] */
    void* warm_start_tangent_impulse;  /* [2*4n] */
    void* normal_impulse;     /* [4n] */
    uint32_t* feature_id1;    /* [4n] */
    uint32_t* feature_id2;    /* [4n] */
} avn_contacts_out;
/* the same row, host -> device (avn_contacts_upload); every field is required */
typedef struct avn_contacts_in {
    const uint32_t* flags;          /* [n] AVN_CP_* */
    const uint8_t* point_count;     /* [n] 0..4 */
    const void* normal;             /* [3n] */
    const void* friction;           /* [n] */
    const void* restitution;        /* [n] */
    const void* anchor1;            /* [3*4n] */
    const void* anchor2;            /* [3*4n] */
    const void* penetration;        /* [4n] */
    const void* normal_speed;       /* [4n] */
    const void* warm_start_normal_impulse;   /* [4n] */
    const void* warm_start_tangent_impulse;  /* [2*4n] */
    const void* normal_impulse;     /* [4n] */
    const uint32_t* feature_id1;    /* [4n] */
    const uint32_t* feature_id2;    /* [4n] */
} avn_contacts_in;

/* ---- systems (one id per reference system on the path; for schedule-faithful drivers and
 *      per-kernel parity tests) ---------------------------------------------------------------- */
typedef enum avn_system {
    AVN_SYS_UPDATE_AABB = 0,                /* collider/backend.rs:498-624 */
    AVN_SYS_COLLECT_COLLISION_PAIRS = 1,    /* broad_phase.rs:214-474 (update intervals + SAP) */
    AVN_SYS_PREPARE_SOLVER_BODIES = 2,      /* solver_body/plugin.rs:173-251 */
    AVN_SYS_PREPARE_JOINTS = 3,             /* xpbd/plugin.rs:125-142 */
    AVN_SYS_PREPARE_CONTACT_CONSTRAINTS = 4,/* solver/plugin.rs:326-448 (softness + generate) */
    AVN_SYS_PRE_PROCESS_VELOCITY_INCREMENTS = 5, /* integrator/mod.rs:260-313 */
    AVN_SYS_INTEGRATE_VELOCITIES = 6,       /* integrator/mod.rs:343-500 (+ clamp_velocities) */
    AVN_SYS_WARM_START = 7,                 /* solver/plugin.rs:453-515 */
    AVN_SYS_SOLVE_CONTACTS_BIAS = 8,        /* solver/plugin.rs:531-619 USE_BIAS=true */
    AVN_SYS_INTEGRATE_POSITIONS = 9,        /* integrator/mod.rs:503-535 + solver_body/plugin.rs:287-295 */
    AVN_SYS_SOLVE_CONTACTS_RELAX = 10,      /* USE_BIAS=false */
    AVN_SYS_XPBD_SOLVE = 11,                /* xpbd/plugin.rs:58-86 (snapshot + solve_xpbd_joint) */
    AVN_SYS_XPBD_VELOCITY_PROJECTION = 12,  /* xpbd/plugin.rs:192-240 */
    AVN_SYS_JOINT_DAMPING = 13,             /* solver/plugin.rs:759-806 */
    AVN_SYS_CLEAR_VELOCITY_INCREMENTS = 14, /* integrator/mod.rs:316-328 */
    AVN_SYS_SOLVE_RESTITUTION = 15,         /* solver/plugin.rs:630-718 */
    AVN_SYS_WRITEBACK_SOLVER_BODIES = 16,   /* solver_body/plugin.rs:255-284 (+ writeback_joint_forces) */
    AVN_SYS_STORE_CONTACT_IMPULSES = 17,    /* solver/plugin.rs:722-755 */
    AVN_SYS_SUBSTEP = 18,                   /* one run of SubstepSchedule (systems 6..13 in order) */
    AVN_SYS_SOLVER = 19,                    /* PhysicsStepSystems::Solver: 2..5, S x SUBSTEP, 14..17 */
    AVN_SYS_NARROW_PHASE = 20,              /* NarrowPhase::update_contacts, narrow_phase/system_param.rs:437-830 */
    AVN_SYS_COUNT_
} avn_system;

/* per-system device timers (mirrors SolverDiagnostics / CollisionDiagnostics,
 * solver/diagnostics.rs:12-38, collision/diagnostics.rs:13-19); milliseconds of the last avn_step */
typedef struct avn_timers {
    double broad_phase_ms;
    double prepare_ms;      /* prepare bodies + joints + constraints + increments */
    double substeps_ms;     /* the whole substep loop */
    double finalize_ms;     /* clear + restitution + writeback + store */
    double step_ms;         /* the whole step on the main stream.  In avn_step with host-uploaded manifolds the broad phase
                               runs on a second stream next to the solver (it only reads what the solver rewrites at the very
                               end): broad_phase_ms is then its own duration and NOT a term of step_ms */
    uint32_t contact_constraint_count;
    uint32_t pair_count;
    uint32_t kernel_launches; /* launches issued (or replayed) in the last step */
    uint32_t bias_pass_launches; /* colour launches of ONE biased-solve pass (solve_contacts<true>) */
    double bias_pass_ms;      /* device time of that pass, MEAN over the last step's substeps, each bracketed by events on the world's
                                 stream: the in-step duration of the dominant kernel for the roofline.  Only with use_graph = 0
                                 (0 otherwise: events captured into a hipGraph cannot be read back) */
    uint32_t island_blocks;   /* workgroups of the island-block substep kernel in the last step; 0 = the substeps ran as
                                 device-wide colour launches (big islands, joints, f64, or AVN_ISLAND_BLOCKS=0) */
    uint32_t side_island_bodies; /* bodies of islands that hold joints and no contact manifold: their substep loop ran on a second stream
                                    next to the other islands' contact passes (island-level concurrency; 0 = one stream) */
} avn_timers;

/* SolverDiagnostics (dynamics/solver/diagnostics.rs:13-37) and CollisionDiagnostics (collision/diagnostics.rs:13-19): the same fields in
 * the same meaning, as milliseconds of the LAST avn_step -- what a replacement plugin writes into those two resources.  The systems of the
 * SubstepSchedule are accumulated over the step's substeps, as the reference accumulates them (solver/plugin.rs:459,481 ...).  Device times
 * come from events on the world's stream; per-substep systems need direct launches (events captured into a hipGraph cannot be read back):
 * with avn_config.use_graph = 1 the five substep fields are 0, `substeps_ms` carries the loop's total and `per_system_valid` is 0.
 * integrate_velocities is fused into the warm-start launch (k_body_warm_start): its time is part of warm_start_ms and the field is 0. */
typedef struct avn_diagnostics {
    double prepare_constraints_ms;         /* SolverSystems::PrepareSolverBodies .. PrepareContactConstraints (+ PrepareJoints) */
    double update_velocity_increments_ms;  /* pre_process_velocity_increments */
    double integrate_velocities_ms;
    double warm_start_ms;
    double solve_constraints_ms;           /* solve_contacts<true> */
    double integrate_positions_ms;
    double relax_velocities_ms;            /* solve_contacts<false> (+ the XPBD systems that follow it in the substep) */
    double apply_restitution_ms;           /* clear_velocity_increments + solve_restitution */
    double finalize_ms;                    /* writeback_solver_bodies + writeback_joint_forces */
    double store_impulses_ms;
    double swept_ccd_ms;                   /* always 0: SweptCcd is outside the path */
    double substeps_ms;                    /* the whole substep loop (always filled) */
    double broad_phase_ms;                 /* CollisionDiagnostics::broad_phase: update_aabb + collect_collision_pairs */
    double narrow_phase_ms;                /* CollisionDiagnostics::narrow_phase: update_contacts + the status-change processing (closed loop only) */
    uint32_t contact_constraint_count;
    uint32_t contact_count;                /* contact pairs in the ContactGraph (closed loop), else broad-phase pairs of the step */
    uint32_t per_system_valid;
    uint32_t reserved0;
} avn_diagnostics;

/* ---- entry points --------------------------------------------------------------------------- */
AVN_API avn_status AVN_FN(world_create)(const avn_config* cfg, avn_world** out);
AVN_API void AVN_FN(world_destroy)(avn_world* w);
AVN_API const char* AVN_FN(last_error)(const avn_world* w); /* w may be NULL: last create error */
AVN_API avn_status AVN_FN(config_set)(avn_world* w, const avn_config* cfg); /* resources changed */

/* replaces the reads of prepare_solver_bodies / pre_process_velocity_increments */
AVN_API avn_status AVN_FN(bodies_upload)(avn_world* w, const avn_bodies* b);
/* replaces the writes of writeback_solver_bodies */
AVN_API avn_status AVN_FN(bodies_download)(avn_world* w, const avn_bodies_out* out);
AVN_API avn_status AVN_FN(solver_bodies_download)(avn_world* w, const avn_solver_bodies_out* out);
/* AccumulatedLocalAcceleration (dynamics/rigid_body/forces/mod.rs:661-673: what ConstantLocalForce / ConstantLocalTorque /
 * ConstantLocalLinearAcceleration / ConstantLocalAngularAcceleration and Forces::apply_local_* accumulate before the step,
 * forces/plugin.rs:145-203), consumed by apply_local_acceleration -- a SubstepSchedule system in front of integrate_velocities
 * (forces/plugin.rs:34-38,62-65,207-241).  Replaces that system: in EVERY substep, for every body that has a SolverBody and no
 * CustomVelocityIntegration (kinematic bodies included: the reference's query has no body-type filter),
 *     rotation = SolverBody::delta_rotation * Rotation;  v += locked(rotation * linear) * h;  omega += locked(rotation * angular) * h
 * with h = Time<Substeps>::delta_secs_f64() as Scalar and `locked` = LockedAxes::apply_to_vec for BOTH vectors (the TRANSLATION locks
 * also mask the angular acceleration: forces/plugin.rs:227-230 with rigid_body/locked_axes.rs:230-243 -- restated as written).
 *   linear, angular: [3n] scalars of the world's type, n = the number of bodies of the last avn_bodies_upload; either may be NULL = zero.
 *   count 0 (or both NULL): no body has a local acceleration -- the system costs nothing (every RigidBody carries the component in the
 *   reference, all zero by default: adding rotation * 0 changes at most the sign of a zero velocity component).
 * The values stay in effect until the next call, an avn_bodies_upload with another body count, or avn_despawn (all three drop them).  The
 * reference clears the component after every step (clear_accumulated_local_acceleration, forces/plugin.rs:68-71,243-251) and accumulates it
 * again before the next: a host mirrors that by uploading what it accumulated before each step (count 0 once nothing is left).
 * Sleeping: a sleeping body has no SolverBody and is skipped, as in the reference; Forces::apply_local_* WAKE the body they push (try_wake_up,
 * forces/query_data.rs:344-349) -- with avn_sleeping_enable that is the host's avn_wake_bodies call, like every other host-side change of a sleeping body. */
AVN_API avn_status AVN_FN(local_accelerations_upload)(avn_world* w, uint32_t count, const void* linear, const void* angular);

/* replaces the ContactGraph/ConstraintGraph reads of prepare_contact_constraints */
AVN_API avn_status AVN_FN(manifolds_upload)(avn_world* w, const avn_manifolds* m);
/* replaces the ContactGraph writes of store_contact_impulses */
AVN_API avn_status AVN_FN(impulses_download)(avn_world* w, const avn_impulses_out* out);
AVN_API avn_status AVN_FN(constraints_download)(avn_world* w, const avn_constraints_out* out);

/* replaces the DistanceJoint queries of prepare_xpbd_joint / solve_xpbd_joint / joint_damping */
AVN_API avn_status AVN_FN(distance_joints_upload)(avn_world* w, const avn_distance_joints* j);
/* replaces the joint queries of prepare_xpbd_joint<T> / solve_xpbd_joint<T> / joint_damping<T> / writeback_joint_forces<T>
 * for every T; supersedes distance_joints_upload (which is the special case joint_type = DISTANCE) */
AVN_API avn_status AVN_FN(joints_upload)(avn_world* w, const avn_joints* j);
AVN_API avn_status AVN_FN(joints_download)(avn_world* w, const avn_joints_out* out);

/* replaces add_new_aabb_intervals (broad_phase.rs:296-315): APPENDS colliders not yet known
 * (by entity_index) at the end of the interval list and refreshes the data of known ones;
 * colliders absent from the call are dropped in place (retain_mut, :230-279). */
AVN_API avn_status AVN_FN(colliders_upload)(avn_world* w, const avn_colliders* c);
/* existing PairKeys of ContactGraph::pair_set (broad_phase.rs:417-420); body pairs whose joints
 * have collision_disabled come from the uploaded joints (:423-428). */
AVN_API avn_status AVN_FN(existing_pairs_upload)(avn_world* w, const uint64_t* pair_keys, size_t n);
/* result of the last COLLECT_COLLISION_PAIRS; buffer owned by the world, valid until the next call */
AVN_API avn_status AVN_FN(pairs_get)(avn_world* w, const avn_pair** out, size_t* n_out);
/* ColliderAabb min/max [3C each] in collider upload order, and the current interval order
 * (entity_index per interval) — inspection */
AVN_API avn_status AVN_FN(aabbs_download)(avn_world* w, void* aabb_min, void* aabb_max,
                                          uint32_t* interval_entities, size_t* n_intervals);

AVN_API avn_status AVN_FN(run_system)(avn_world* w, avn_system sys);
/* one PhysicsSchedule pass over the path: (UPDATE_AABB + COLLECT_COLLISION_PAIRS if colliders were
 * uploaded) then AVN_SYS_SOLVER.  Asynchronous on the world's stream in the product. */
AVN_API avn_status AVN_FN(step)(avn_world* w);
AVN_API avn_status AVN_FN(synchronize)(avn_world* w);
AVN_API avn_status AVN_FN(timers_get)(avn_world* w, avn_timers* out);
/* Measurement hook (bench.py roofline): run `sys` `repeats` times back to back on the world's stream, bracketed by
 * events ON THAT STREAM; returns the total elapsed milliseconds and the number of kernel launches issued.
 * (The oracle times the same calls with a host clock.)  The reference's equivalent is the per-system
 * `Instant::now()/elapsed()` accumulation into SolverDiagnostics (solver/plugin.rs:459,481). */
AVN_API avn_status AVN_FN(diagnostics_get)(avn_world* w, avn_diagnostics* out);
AVN_API avn_status AVN_FN(profile_system)(avn_world* w, avn_system sys, uint32_t repeats, double* total_ms,
                                           uint32_t* kernel_launches);

/* batch form of contact_query::contact_manifolds (see avn_shape_pairs); scalar type = the world's */
AVN_API avn_status AVN_FN(contact_manifolds)(avn_world* w, const avn_shape_pairs* pairs, const avn_query_manifolds_out* out);

AVN_API avn_status AVN_FN(collider_materials_upload)(avn_world* w, const avn_collider_materials* m);
AVN_API avn_status AVN_FN(contact_pairs_add)(avn_world* w, const avn_contact_pairs* p);
/* ContactGraph::remove_edge_by_id: the rows are freed (ids may be reused by a later _add) and the pairs' keys leave the
 * broad phase's pair set */
AVN_API avn_status AVN_FN(contact_pairs_remove)(avn_world* w, const uint32_t* contact_id, size_t n);
/* ContactGraph::active_pairs (the pairs update_contacts iterates) */
AVN_API avn_status AVN_FN(active_pairs_set)(avn_world* w, const uint32_t* contact_id, size_t n);
/* status changes of the last AVN_SYS_NARROW_PHASE; buffer owned by the world, valid until the next call */
AVN_API avn_status AVN_FN(contact_changes_get)(avn_world* w, const avn_contact_change** out, size_t* n_out);
/* GraphColor::manifold_handles of all colours (colour-major, manifold index 0 implied): from now on
 * prepare_contact_constraints reads its manifolds from the contact table through these handles, and
 * store_contact_impulses writes the impulses back to it.  Replaces avn_manifolds_upload for worlds that run the device
 * narrow phase; n = color_offsets[24]. */
AVN_API avn_status AVN_FN(manifold_handles_upload)(avn_world* w, const uint32_t* color_offsets, const uint32_t* contact_id);
AVN_API avn_status AVN_FN(contacts_download)(avn_world* w, const uint32_t* contact_id, size_t n, const avn_contacts_out* out);
/* the inverse: overwrite rows that exist (avn_contact_pairs_add) with what NarrowPhase::update reads of the previous step -- the
 * ContactPair flags (collision/contact_types/mod.rs:56-110; whether the pair was touching decides the started / stopped events,
 * narrow_phase/system_param.rs:560-584) and manifolds[0] with its points, whose feature ids carry the warm-start impulses over in
 * match_contacts (contact_types/mod.rs:568-600).  For rows that move between worlds when a closed-loop world is re-partitioned or
 * rebuilt: a world that receives bodies, colliders in interval order, avn_existing_pairs_upload, avn_contact_pairs_add and these
 * rows continues bit-identically (tests/test_pipeline_cpu.py, tests/test_gpu_pipeline.py).  The collider slots of the row and its
 * manifold_count_change stay as they are; points beyond point_count are not read. */
AVN_API avn_status AVN_FN(contacts_upload)(avn_world* w, const uint32_t* contact_id, size_t n, const avn_contacts_in* in);

/* ---- standalone closed loop -----------------------------------------------------------------------------------------
 * For drivers WITHOUT Avian's host structures (benches, demos, tests): the library keeps them itself — IdPool
 * (data_structures/id_pool.rs:31-40), the ContactGraph edge list / active pairs (contact_graph.rs:521-566), the status-change
 * processing of NarrowPhase::update (narrow_phase/system_param.rs:141-389) and the ConstraintGraph
 * (solver/constraint_graph.rs:163-296) — and avn_step becomes the whole PhysicsSchedule pass over the path:
 *   UPDATE_AABB -> COLLECT_COLLISION_PAIRS -> (new rows) -> NARROW_PHASE -> (status changes -> push / pop -> handles) -> SOLVER.
 * An Avian integration does NOT use this: it forwards its own structures' changes through the calls above.
 * avn_pipeline_enable(w, 1): ALL of that bookkeeping runs on the device (k_graph.hip: ids in emission order, the status-change loop in
 *   ascending ContactId, the greedy colouring and the push / swap_remove order of every colour list replayed exactly, the overflow
 *   colour solved in list order by a dataflow pass); per step the host reads three small counter blocks.
 * avn_pipeline_enable(w, 2) (or AVN_PIPELINE_HOST=1): the same loop with host-side structures (round-1 path, kept for A/B runs).
 * avn_pipeline_enable(w, 0): off. */
typedef struct avn_pipeline_stats {
    uint64_t pairs_added, pairs_removed, manifolds_pushed, manifolds_popped;  /* since avn_pipeline_enable */
    uint32_t active_pairs, manifolds;      /* now */
    uint32_t last_status_changes;          /* in the last step */
    uint32_t last_overflow_manifolds;      /* manifolds in colour 23 in the last step */
    double last_host_ms;                   /* host bookkeeping time of the last step (status processing + uploads) */
} avn_pipeline_stats;
AVN_API avn_status AVN_FN(pipeline_enable)(avn_world* w, int on);
AVN_API avn_status AVN_FN(pipeline_stats_get)(avn_world* w, avn_pipeline_stats* out);
/* the colour lists the pipeline holds: offsets[25] and the contact ids (buffer owned by the world) */
AVN_API avn_status AVN_FN(pipeline_handles_get)(avn_world* w, uint32_t* color_offsets, const uint32_t** contact_id, size_t* n_out);
/* The ContactIds the loop gave the LAST step's new pairs (IdPool::alloc_id in emission order), entry i for pair i of avn_pairs_get: what a host that mirrors
 * the ContactGraph keys its own edge list by (Avian's CollisionStart / CollisionEnd / CollidingEntities are rebuilt from these and avn_contact_changes_get:
 * collision/narrow_phase/system_param.rs:141-389).  In the closed loop avn_contact_changes_get returns the status changes of the last step's narrow phase,
 * ascending ContactId; both buffers are owned by the world and valid until the next call on it (call them right after avn_step, before avn_despawn).  The id list is
 * EMPTY after avn_pipeline_enable, after avn_despawn (it may name rows that just left) and after a step that found no new pair: it only ever describes a completed step. */
AVN_API avn_status AVN_FN(pipeline_new_pair_ids_get)(avn_world* w, const uint32_t** contact_ids, size_t* n);

/* ---- level-2 sharding: ONE contact island split over several worlds (x-slabs), SURVEY.md section 8(e) ---------------------------------
 * Every world holds the manifolds it OWNS (those whose body1 -- body2 when body1 is static -- lies in its slab), the bodies they touch and
 * the static bodies.  A body present in several worlds ("shared") is integrated identically by each of them; what differs is who solves
 * which manifold.  With ONE global colouring (the reference's, over the whole island: avn_manifolds.color_offsets of every world uses the
 * global colour of each manifold) every world runs the colour launches of the single-world step on its subset, and after each colour the
 * world whose manifold moved a shared body hands that body's (linear, angular) velocity to the other holders: the single world's state is
 * restored before the next colour, so the split run is BIT-IDENTICAL to the unsplit one (a body is in at most one manifold per colour).
 * Round 6: overflow-colour manifolds on shared bodies travel level by level (avn_halo_overflow_levels_upload) and joints on shared bodies as owned components with a
 * joint slot (avn_halo_joint_slot_set): exchange SLOTS = 23 colours, then the overflow levels (or the one overflow colour), then -- when planned -- the joint slot.
 *
 *   avn_halo_plan_upload   per (colour, peer): local body indices to send and to receive, in the same order on both sides
 *   avn_run_color_pass     one colour of one contact pass (AVN_SYS_WARM_START / SOLVE_CONTACTS_BIAS / SOLVE_CONTACTS_RELAX / SOLVE_RESTITUTION)
 *   avn_halo_pack/unpack   host transport of one (colour, peer) list: 8 scalars per body (linear.xyz, w lane, angular.xyz, w lane)
 *   avn_comm_unique_id / avn_comm_init   library transport: RCCL; avn_step then exchanges by grouped ncclSend / ncclRecv on the world's
 *                          stream after every colour and no host code runs inside the step */
typedef struct avn_halo_plan {
    uint32_t n_peers;
    const int32_t* peer_rank;      /* [n_peers] */
    const uint32_t* send_offsets;  /* [24 * n_peers + 1]: list of (colour c, peer p) = send_bodies[send_offsets[c * n_peers + p] .. [.. + 1]) */
    const int32_t* send_bodies;    /* local body indices */
    const uint32_t* recv_offsets;  /* [24 * n_peers + 1] */
    const int32_t* recv_bodies;
} avn_halo_plan;
AVN_API avn_status AVN_FN(halo_plan_upload)(avn_world* w, const avn_halo_plan* plan);
/* Round 6: the overflow colour ACROSS worlds.  The reference solves it serially in list order (solver/plugin.rs:461-467); only the relative order of manifolds that share
 * a body matters, so the planner cuts the GLOBAL list into levels (level(m) = 1 + the highest level of an earlier overflow manifold on one of m's non-static bodies) and
 * every level becomes an exchange SLOT behind the colours: slot c < 23 = colour c, slot 23 + l = overflow level l.  avn_halo_overflow_levels_upload tells a world the
 * number of levels and the level of each of ITS overflow manifolds (local order); call it BEFORE avn_halo_plan_upload, whose offset arrays then hold
 * (23 + n_levels) * n_peers + 1 entries.  Never called (or n_levels <= 1): 24 slots, slot 23 = the world's whole overflow colour, as before.  In avn_run_color_pass /
 * avn_halo_pack / avn_halo_unpack `color` is the slot. */
AVN_API avn_status AVN_FN(halo_overflow_levels_upload)(avn_world* w, uint32_t n_levels, const uint32_t* level_of_local_overflow_manifold, size_t count);
/* Round 6: joints ACROSS worlds.  The reference walks the joints of a type serially (xpbd/plugin.rs:145-189; joint_damping, solver/plugin.rs:756-830), so the planner
 * (avn_level2_plan_create_joints) makes a joint COMPONENT the unit of ownership -- joints linked through non-static bodies and, with JointDamping, through the type's
 * DUMMY pair that stands in for bodies without a SolverBody (plugin.rs:766-767); owner = the slab of the component's lowest non-static body, which holds all its bodies --
 * and adds ONE exchange slot behind the colours and overflow levels: after the joint systems of a substep (XPBD solve, velocity projection, joint damping) the owner
 * sends the SolverBody records of the component's SHARED bodies to their other holders.  avn_halo_joint_slot_set(w, joint_slot, global_joints), BEFORE
 * avn_halo_plan_upload: joint_slot != 0 -> the plan's offset arrays hold one more slot (index 23 + n_levels), whose records in avn_halo_pack / _unpack are 16 scalars per
 * body (delta_position.xyz 0 | delta_rotation.xyzw | linear_velocity.xyz 0 | angular_velocity.xyz 0) instead of 8; global_joints != 0 -> the unsplit world holds joints,
 * so this world runs the XPBD snapshot / velocity projection over its bodies even when it owns no joint (the unsplit world does: xpbd/plugin.rs:61-76,192-240). */
AVN_API avn_status AVN_FN(halo_joint_slot_set)(avn_world* w, uint32_t joint_slot, uint32_t global_joints);
AVN_API avn_status AVN_FN(run_color_pass)(avn_world* w, avn_system pass, uint32_t color);
AVN_API avn_status AVN_FN(halo_pack)(avn_world* w, uint32_t color, uint32_t peer, void* out /* [8 * count] scalars */, size_t* count);
AVN_API avn_status AVN_FN(halo_unpack)(avn_world* w, uint32_t color, uint32_t peer, const void* in, size_t count);
/* ---- host planners of the x-slab sharded broad phase and of the re-partition (integer work, no device) -------------
 * avn_slab_select: slabs are balanced contiguous ranges of the colliders sorted by AABB min.x, cut ON key values (colliders with equal
 * min.x never straddle a cut); rank r's sub-world = the colliders it owns + the later ones its owned intervals can reach (min.x <= the
 * largest max.x it owns), listed in the single world's PERSISTENT interval order (prev_order: the order before this frame's sort, NULL =
 * upload order; colliders not in it are appended, broad_phase.rs:296-315).  next_order = that order after this frame's stable sort
 * (-0.0 == +0.0, non-finite keys dropped, :230-279, :479-487): next frame's prev_order, identical on every rank. */
typedef struct avn_slab_in {
    uint32_t n_colliders;
    const double* aabb_min_x;    /* [n] */
    const double* aabb_max_x;    /* [n] */
    const uint32_t* prev_order;  /* [n_prev] collider indices, or NULL */
    uint32_t n_prev;
    uint32_t n_ranks, rank;
} avn_slab_in;
AVN_API avn_status AVN_FN(slab_select)(const avn_slab_in* in, uint32_t* local /* [n] */, uint8_t* owned /* [n] */, uint32_t* n_local,
                                       uint32_t* next_order /* [n], may be NULL */, uint32_t* n_next);
/* avn_interval_orders_merge: the global persistent interval order out of the ranks' orders when islands migrate (shard.repartition): k-way
 * merge on (the min.x each list is sorted by, entity index) that never reorders one list's own entries; an entity present in several lists
 * (static bodies) is kept once; a NaN key (never swept) sorts first. */
AVN_API avn_status AVN_FN(interval_orders_merge)(uint32_t n_lists, const uint32_t* const* entities, const double* const* keys, const uint32_t* lengths,
                                                 uint32_t* out /* [sum of lengths] */, uint32_t* n_out);

/* The level-2 planner (host integer work, no device): which world owns which manifold, which bodies each world holds, and the per-colour
 * send / receive lists -- everything avn_halo_plan_upload and the per-rank uploads need, so that a host in any language shards without
 * re-implementing it.  Slabs are cut at the quantiles of the non-static bodies' x; a manifold belongs to the slab of its body1 (body2 when
 * body1 is static); a body is SHARED when more than one world holds it; lists are in ascending global body index (the same order on both
 * sides of every pair of ranks).  When an overflow-colour manifold touches a shared body the overflow colour is cut into levels (avn_halo_overflow_levels_upload) and
 * halo.send_offsets / recv_offsets hold (23 + n_levels) * n_peers + 1 entries -- avn_level2_plan_rank_overflow returns n_levels (1: the plain 24 slots) and the levels of
 * the rank's overflow manifolds.  Joints: avn_level2_plan_create_joints (a joint component is owned by one world, its shared bodies travel in the joint slot). */
typedef struct avn_level2_in {
    uint32_t n_bodies;
    const uint8_t* rb_type;        /* [n_bodies] AVN_RB_* */
    const double* center_x;        /* [n_bodies] */
    uint32_t n_manifolds;
    const int32_t* body1;          /* [n_manifolds] the GLOBAL colour-major manifold set */
    const int32_t* body2;
    const uint32_t* color_offsets; /* [25] of that set */
    uint32_t n_ranks;
} avn_level2_in;
typedef struct avn_level2_rank {   /* pointers are owned by the plan */
    uint32_t n_bodies;  const int32_t* bodies;            /* global index of every local body, ascending (local index = position) */
    uint32_t n_manifolds; const uint32_t* manifolds;      /* global manifold indices, ascending (= colour-major) */
    const uint32_t* color_offsets;                        /* [25] of the local set */
    avn_halo_plan halo;                                   /* LOCAL body indices */
} avn_level2_rank;
typedef struct avn_level2_plan avn_level2_plan;
AVN_API avn_status AVN_FN(level2_plan_create)(const avn_level2_in* in, avn_level2_plan** out);
AVN_API void AVN_FN(level2_plan_destroy)(avn_level2_plan* plan);
AVN_API avn_status AVN_FN(level2_plan_rank)(const avn_level2_plan* plan, uint32_t rank, avn_level2_rank* out);
/* the same plan for a world with joints (see avn_halo_joint_slot_set): joints of a rank = the global joint array restricted to the components it owns, in array order */
typedef struct avn_level2_joints {
    uint32_t n_joints;
    const int32_t* body1;        /* [J] global body indices */
    const int32_t* body2;        /* [J] */
    const uint8_t* joint_type;   /* [J] AVN_JOINT_* */
    uint32_t damped;             /* the joints carry JointDamping (avn_joints.damping_linear != NULL) */
} avn_level2_joints;
AVN_API avn_status AVN_FN(level2_plan_create_joints)(const avn_level2_in* in, const avn_level2_joints* joints, avn_level2_plan** out);
AVN_API avn_status AVN_FN(level2_plan_rank_joints)(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_joints, const uint32_t** joints /* global joint indices, ascending */,
                                                   uint32_t* joint_slot /* the halo arrays carry the joint slot */, uint32_t* global_joints);
AVN_API avn_status AVN_FN(level2_plan_rank_overflow)(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_levels, const uint32_t** level_of_local_overflow_manifold /* [rank's overflow manifolds] */);
#define AVN_COMM_ID_BYTES 128
AVN_API avn_status AVN_FN(comm_unique_id)(uint8_t* out /* [AVN_COMM_ID_BYTES] */);
AVN_API avn_status AVN_FN(comm_init)(avn_world* w, const uint8_t* unique_id, int n_ranks, int rank);
/* Level-1 sharding's per-step exchange, issued BY THE LIBRARY: this world's dynamic bounds (avn_dynamic_bounds: union of the ColliderAabbs of the
 * colliders on non-static bodies, after AVN_SYS_UPDATE_AABB / avn_step) are reduced on the device and all-gathered over the communicator of
 * avn_comm_init (ncclAllGather of 48 bytes per rank on the world's stream; without a communicator the world is its own only rank).
 * bounds [n_ranks][6] = (min.xyz, max.xyz) per rank; overlaps [cap][2]: the rank pairs (i < j) whose bounds intersect -- islands of different
 * ranks coming into AABB contact, the trigger of the re-partition (avian_amd/shard.py: repartition; broad_phase.rs:373-474 decides pairs on
 * the same closed-interval test).  *n_overlaps = how many exist (may exceed cap). */
AVN_API avn_status AVN_FN(bounds_exchange)(avn_world* w, double* bounds, uint32_t cap_ranks, uint32_t* n_ranks, uint32_t* overlaps, uint32_t cap_overlaps, uint32_t* n_overlaps);

/* PairKey::new (data_structures/pair_key.rs:14-21) — exported so hosts build identical keys */
AVN_API uint64_t AVN_FN(pair_key)(uint32_t id1, uint32_t id2);

/* Host-side ConstraintGraph (constraint_graph.rs:163-296) for standalone drivers that do not
 * have Avian's own graph: persistent greedy colouring over manifolds keyed by a caller handle. */
typedef struct avn_constraint_graph avn_constraint_graph;
AVN_API avn_status AVN_FN(constraint_graph_create)(uint32_t body_capacity, avn_constraint_graph** out);
AVN_API void AVN_FN(constraint_graph_destroy)(avn_constraint_graph* g);
/* push_manifold: returns the colour chosen; `handle` is an opaque caller id (e.g. contact_id<<2|manifold_index) */
AVN_API int32_t AVN_FN(constraint_graph_push)(avn_constraint_graph* g, uint64_t handle, uint32_t body1,
                                              uint32_t body2, int is_static1, int is_static2);
/* push_manifold for `n` manifolds in array order (identical to n calls of _push); colors_out may be NULL */
AVN_API avn_status AVN_FN(constraint_graph_push_batch)(avn_constraint_graph* g, size_t n, const uint64_t* handles,
                                                       const uint32_t* body1, const uint32_t* body2,
                                                       const uint8_t* is_static1, const uint8_t* is_static2, int8_t* colors_out);
/* pop_manifold (swap-remove) */
AVN_API avn_status AVN_FN(constraint_graph_pop)(avn_constraint_graph* g, uint64_t handle);
/* colour-major handle list: offsets[25], handles[count] in manifold_handles order */
AVN_API avn_status AVN_FN(constraint_graph_lists)(const avn_constraint_graph* g, uint32_t* offsets,
                                                  uint64_t* handles, size_t capacity, size_t* count);

/* ---- multi-GPU sharding: interaction islands -------------------------------------------------
 * Islands as in the reference's IslandPlugin (dynamics/solver/islands/mod.rs:1-10): bodies connected by constraint
 * edges form an island; static bodies never merge islands (merge_islands early-returns for them, :822-834).  The
 * edge set given here is the caller's choice; for sharding it is (broad-phase pairs U manifolds U joints), a superset
 * of the reference's touching-contact edges, so that everything one island can interact with this step lives on one
 * rank.  Islands are then assigned to `n_ranks` as contiguous slabs along x (the SAP axis) of balanced weight
 * (weight = bodies + edges, the reference's island statistics, islands/mod.rs:213-232).  Pure host integer work. */
typedef struct avn_islands_in {
    uint32_t n_bodies;
    const uint8_t* rb_type;     /* [n] AVN_RB_*; static bodies get island -1 / rank -1 (replicated where needed) */
    const double* center_x;     /* [n] x of the body position (slab ordering) */
    uint32_t n_edges;
    const int32_t* edge_body1;  /* [E] */
    const int32_t* edge_body2;  /* [E] */
    uint32_t n_ranks;
} avn_islands_in;
/* island_of_body[n]: island id (islands numbered by their smallest body index, ascending) or -1;
 * rank_of_body[n]: owning rank or -1; *n_islands: number of islands */
AVN_API avn_status AVN_FN(islands_partition)(const avn_islands_in* in, int32_t* island_of_body, int32_t* rank_of_body,
                                              uint32_t* n_islands);

/* ---- islands and sleeping on the device ---------------------------------------------------------------------
 * avn_islands_get: the simulation islands of the world's CURRENT constraint graph -- connected components over the contact manifolds the
 * solver holds (closed loop: the touching pairs that generate constraints; otherwise the uploaded manifolds) and the joints, where only
 * non-static bodies connect (islands/mod.rs:1-10, :814-990 merge_islands: a static body never merges islands; kinematic ones do).  Label =
 * the LOWEST body index of the island (0xFFFFFFFF for static bodies): the partition the reference's persistent union-find holds once its
 * pending splits are done (split_island, :995+, is deferred there: ONE island per step, and only when a body wants to sleep).
 * Computed by a lock-free union-find on the device (k_islands.hip); feeds the island-block builder and the multi-GPU partitioner.
 *
 * avn_sleep_update: update_sleeping_states (islands/sleeping.rs:184-241) for one step, on the device, in the reference's arithmetic:
 *   per awake body        v2 = |SolverBody.linear_velocity|^2, w2 = |SolverBody.angular_velocity|^2      (Scalar; bodies uploaded with
 *                         AVN_BODY_SLEEPING are skipped like the reference's `Without<Sleeping>`, their timer stays)
 *                         rests = v2 < length_unit^2 * (lin * |lin|) && w2 < ang * |ang|                 (thresholds f32, "keep signs")
 *                         SleepTimer (f32) += delta_secs (f32) if rests else = 0
 *   per island            kept awake if any awake body has SleepTimer < time_to_sleep; an awake island nobody keeps awake RESTS
 *                         (sleep_islands, :243-280, would put it to sleep); an island with sleeping bodies that IS kept awake WAKES.
 * What the call reports is the DECISION.  Putting an island to sleep / waking it (SleepIslands / WakeIslands, :300-520: Sleeping
 * components, the ContactGraph's sleeping set, pop / push of the constraint handles) is the host's side of the boundary and is NOT done
 * here; neither is the one-step delay a pending split adds in the reference.  Call it after avn_step (the Sleeping set runs after the
 * Solver set, schedule/mod.rs). */
typedef struct avn_sleep_params {
    uint32_t struct_size;
    float time_to_sleep;          /* TimeToSleep, default 0.5 s */
    float linear_threshold;       /* SleepThreshold.linear, default 0.15 */
    float angular_threshold;      /* SleepThreshold.angular, default 0.15 */
    float delta_secs;             /* Time::delta_secs() of the step (f32) */
    double length_unit;           /* PhysicsLengthUnit, default 1 */
    /* optional per-body components, host arrays of n_bodies entries (NULL = the world-level value above for every body) */
    const float* body_linear_threshold;    /* SleepThreshold.linear of each body */
    const float* body_angular_threshold;   /* SleepThreshold.angular */
    const uint8_t* body_sleeping_disabled; /* 1 = SleepingDisabled: skipped by update_sleeping_states (`Without<SleepingDisabled>`), timer reset to 0
                                              and its island kept awake (wake_islands_with_sleeping_disabled, sleeping.rs:164-182) */
} avn_sleep_params;
typedef struct avn_sleep_stats {
    uint32_t n_islands;                   /* islands of the current constraint graph */
    uint32_t n_island_bodies;             /* bodies with a BodyIslandNode: non-static, not AVN_BODY_DISABLED */
    uint32_t n_sleeping_bodies;           /* of those, uploaded with AVN_BODY_SLEEPING (the host has put their island to sleep) */
    uint32_t n_awake_bodies;              /* n_island_bodies - n_sleeping_bodies: the solver's N of the step just taken */
    uint32_t n_resting_islands;           /* awake islands no body keeps awake: sleep_islands would put them to sleep (sleeping.rs:262-266) */
    uint32_t n_resting_bodies;            /* bodies in those islands */
    uint32_t n_waking_islands;            /* islands with a sleeping body AND a body that keeps them awake (a contact merged an awake island
                                             into a sleeping one): sleep_islands would wake them (sleeping.rs:258-261) */
    uint32_t n_waking_bodies;             /* sleeping bodies in those islands */
} avn_sleep_stats;
typedef struct avn_sleep_out {
    float* sleep_timer;        /* [n_bodies] SleepTimer after the update (0 for bodies without an island node); may be NULL */
    uint32_t* island;          /* [n_bodies] island label (lowest body index), 0xFFFFFFFF for bodies without an island node; may be NULL */
    uint8_t* island_rests;     /* [n_bodies] 1 = the body's island is awake and rests (would be put to sleep); may be NULL */
    uint8_t* island_wakes;     /* [n_bodies] 1 = the body's island holds sleeping bodies and is kept awake (would be woken); may be NULL */
} avn_sleep_out;
AVN_API avn_status AVN_FN(islands_get)(avn_world* w, uint32_t* island_of_body /* [n_bodies] */, uint32_t* n_islands);
AVN_API avn_status AVN_FN(sleep_update)(avn_world* w, const avn_sleep_params* p, avn_sleep_stats* stats);
AVN_API avn_status AVN_FN(sleep_get)(avn_world* w, const avn_sleep_out* out);
/* SleepTimer = 0 for the listed bodies (n = 0: for all): what waking an island does to its bodies (sleeping.rs:492) */
AVN_API avn_status AVN_FN(sleep_reset)(avn_world* w, const uint32_t* bodies, size_t n);


/* ------------------------------------------------------------------------------------------------------------------------------------
 * Persistent simulation islands and the ACTUATION of sleeping (second half).
 *
 * Reference: dynamics/solver/islands/mod.rs (PhysicsIslands: create / remove, add_contact :513-582, remove_contact :594-660, add_joint,
 * merge_islands :814-990, split_island :995-1280, BodyIslandNode hooks :1330-1414) and islands/sleeping.rs (update_sleeping_states
 * :184-241, sleep_islands :243-280, SleepIslands :355-420, WakeIslands :470-540), on top of collision/contact_types/contact_graph.rs
 * (edge lists of a collider in petgraph order: outgoing edges newest first, then incoming edges newest first; sleep_entity_with /
 * wake_entity_with :705-838) and data_structures/stable_graph.rs.
 *
 * Two layers:
 *  (1) avn_island_manager -- the reference's PhysicsIslands plus the island-relevant part of ContactGraph / JointGraph as a HOST structure
 *      (it is host state of the ECS in the reference too).  Everything whose ORDER the reference defines through linked lists is kept in
 *      that order: an island's body list (merges append the smaller island; a split rebuilds it in depth-first visit order), the slab ids
 *      of islands (vacant keys are reused last-freed-first), a collider's edge list.  A Bevy integration can keep Avian's own resources
 *      instead and only use layer (2); the standalone closed loop owns one manager per world.
 *  (2) avn_sleeping_enable -- the closed loop (avn_pipeline_enable) drives its manager itself: island bookkeeping inside the status loop,
 *      WakeIslands after the narrow phase, split_island in Finalize, the Sleeping set at the end of avn_step, and the actuation on the
 *      device: a sleeping island's touching constraint-generating pairs leave the colour lists (pop_manifold in body-list x edge-list
 *      order -- that order decides where swap_remove moves the other handles), its pairs stop being updated by the narrow phase, its
 *      bodies lose their SolverBody and their intervals turn inactive; waking pushes the manifolds back in the same order (which decides
 *      their colours) and resets the bodies' SleepTimers. */
typedef struct avn_island_manager avn_island_manager;
AVN_API avn_island_manager* AVN_FN(islands_create)(void);
AVN_API void AVN_FN(islands_destroy)(avn_island_manager* m);
/* BodyIslandNode::on_add: a new island holding `body` (bodies with a SolverBody: dynamic / kinematic, enabled); call in spawn order */
AVN_API avn_status AVN_FN(islands_body_add)(avn_island_manager* m, uint32_t body);
/* RigidBodyColliders of `body` gains `collider` (Entity::index() of the collider); colliders of bodies WITHOUT a node (static) are announced
 * with body = 0xFFFFFFFF */
AVN_API avn_status AVN_FN(islands_collider_add)(avn_island_manager* m, uint32_t collider, uint32_t body);
/* JointGraph::add_joint + PhysicsIslands::add_joint: merges the two bodies' islands; call in joint spawn order */
AVN_API avn_status AVN_FN(islands_joint_add)(avn_island_manager* m, uint32_t joint, uint32_t body1, uint32_t body2);
/* ContactGraph::add_edge_and_key_with: the pair enters both colliders' edge lists at their heads */
AVN_API avn_status AVN_FN(islands_pair_add)(avn_island_manager* m, uint32_t contact_id, uint32_t collider1, uint32_t collider2);
/* one iteration of the status loop of NarrowPhase::update (system_param.rs:141-389) for `contact_id`: `flags` are the pair's AVN_CP_* bits
 * as avn_contact_change reports them (DISJOINT_AABB removes the pair; STARTED_TOUCHING / STOPPED_TOUCHING / STARTED_GENERATING_CONSTRAINTS
 * link / unlink the contact and queue its island for waking when it sleeps).  Call in ascending ContactId like the reference. */
AVN_API avn_status AVN_FN(islands_status_change)(avn_island_manager* m, uint32_t contact_id, uint32_t flags, uint32_t manifold_count);
/* WakeIslands(sorted, deduplicated islands queued by the status loop), then returns how many manifolds it pushed back: their contact ids in
 * push order and the bodies whose Sleeping component it removed, readable until the next call through avn_islands_last_*.  */
AVN_API avn_status AVN_FN(islands_flush_wake)(avn_island_manager* m);
/* split_island(split_candidate) -- SolverSystems::Finalize */
AVN_API avn_status AVN_FN(islands_split_candidate)(avn_island_manager* m);
/* The same split with the contact neighbours handed in by the caller (round 6) -- for a host that holds the contact graph's adjacency elsewhere; the closed
 * loop builds it on the device.  CSR over bodies: adj[off[b] .. off[b + 1]) = the OTHER body of every contact edge of body b that (a) holds constraint
 * handles and (b) whose other body owns an island node, in the order split_island's depth-first walk meets them (islands/mod.rs:1112-1148): the body's
 * colliders in RigidBodyColliders order; per collider the outgoing edges newest first, then the incoming edges newest first
 * (data_structures/stable_graph.rs:640-675).  off has n_bodies + 1 entries.  Joints are walked from the manager's own JointGraph.  The result equals
 * avn_islands_split_candidate's; the oracle's implementation checks the CSR rows of the candidate's bodies against its own edge lists first and returns
 * AVN_ERR_STATE when they disagree.
 * `labels` (may be NULL) [n_bodies]: a component id per body (a body index of the component) over exactly those edges plus the joints, 0xFFFFFFFF for a body
 * without a node (the closed loop labels them on the device).  With labels the split's bookkeeping is known without the walk -- the pieces are the labels, a piece
 * takes its slab key when the old body list first names one of its bodies (the walk starts its pieces in that order), sizes are counts, constraints_removed and the
 * timers restart at 0 -- and only the ORDER inside the pieces' body lists is outstanding: the walk then runs on a worker thread, `off` / `adj` must stay valid and
 * unchanged until avn_islands_split_join (or the next call that needs such an order: the Sleeping set putting one of the pieces to sleep, a merge of a piece INTO
 * another island, the next split, despawn, avn_islands_state -- they join themselves), and the call returns at once.  The result is the same either way. */
AVN_API avn_status AVN_FN(islands_split_candidate_adjacency)(avn_island_manager* m, const uint32_t* off, const uint32_t* adj, uint32_t n_bodies, const uint32_t* labels);
AVN_API avn_status AVN_FN(islands_split_join)(avn_island_manager* m);
/* The Sleeping set after the solver.  `sleep_timer` [n_bodies]: the SleepTimers AFTER update_sleeping_states' increment / reset (the caller
 * owns that arithmetic: it reads SolverBody velocities); `flags` [n_bodies]: bit 0 = the body took part in update_sleeping_states (has a
 * SolverBody, not Sleeping, not SleepingDisabled), bit 1 = SleepingDisabled (wake_islands_with_sleeping_disabled).  Runs the island side of
 * update_sleeping_states (awake bits, split candidate), wake_islands_with_sleeping_disabled, sleep_islands, then SleepIslands and
 * WakeIslands; results through avn_islands_last_*. */
AVN_API avn_status AVN_FN(islands_sleeping_systems)(avn_island_manager* m, const float* sleep_timer, const uint8_t* flags, uint32_t n_bodies, float time_to_sleep);
/* WakeBody / SleepBody commands (user-driven; sleeping.rs:283-352, 438-452) */
AVN_API avn_status AVN_FN(islands_wake_body)(avn_island_manager* m, uint32_t body);
AVN_API avn_status AVN_FN(islands_sleep_body)(avn_island_manager* m, uint32_t body);
/* what the last flush_wake / sleeping_systems / wake_body / sleep_body did, in the reference's order: contact ids whose manifolds were
 * popped (SleepIslands) and pushed (WakeIslands), contact ids moved to the sleeping / the active pair set, bodies put to sleep / woken.
 * Pointers stay valid until the next call on the manager. */
typedef struct avn_islands_result {   /* FIXED layout (round 4's: seven pointer / count pairs).  Round 5 had put a `struct_size` member in FRONT of them, which moved
                                         every field of a round-4 host by 8 bytes: taken out again in round 6.  If this result ever has to grow, it grows
                                         through a new entry point, not through this struct. */
    const uint32_t* popped;  size_t n_popped;
    const uint32_t* pushed;  size_t n_pushed;
    const uint32_t* pairs_slept; size_t n_pairs_slept;    /* ContactEdgeFlags::SLEEPING set (every touching pair, constraint-generating or not) */
    const uint32_t* pairs_woken; size_t n_pairs_woken;
    const uint32_t* bodies_slept; size_t n_bodies_slept;
    const uint32_t* bodies_woken; size_t n_bodies_woken;  /* their SleepTimer is reset to 0 */
    const uint32_t* pairs_removed; size_t n_pairs_removed; /* avn_islands_collider_remove: the collider's edges in removal order (they left the ContactGraph) */
} avn_islands_result;
AVN_API avn_status AVN_FN(islands_last_result)(avn_island_manager* m, avn_islands_result* out);
/* Despawn (round 4; what avn_despawn drives inside the closed loop, for a host that keeps its own loop):
 *  collider_remove: remove_collider (collision/narrow_phase/mod.rs:399-457) -- the collider's edges in the ContactGraph's edge-list order (outgoing
 *    newest first, then incoming newest first): result.popped = the contact ids whose constraint handles the host must pop (one entry per handle, in
 *    order; TOUCHING pairs only), result.pairs_removed = every edge, in order; touching pairs linked into an island are unlinked
 *    (constraints_removed += 1); the collider leaves its body's RigidBodyColliders.
 *  body_remove: BodyIslandNode::on_remove (islands/mod.rs:1336-1400) followed by the queued WakeIslands([the island the body was in]) -- result as for
 *    avn_islands_wake_body.  Remove the body's colliders first.
 *  renumber_bodies: new_index[old] = new body index | 0xFFFFFFFF (removed): the host compacted its body arrays. */
AVN_API avn_status AVN_FN(islands_collider_remove)(avn_island_manager* m, uint32_t collider);
AVN_API avn_status AVN_FN(islands_body_remove)(avn_island_manager* m, uint32_t body);
AVN_API avn_status AVN_FN(islands_renumber_bodies)(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old);
/* A joint leaves (round 5): what `remove_joint_from_graph` does when a joint entity loses its joint component (dynamics/solver/joint_graph/plugin.rs:163-194) --
 * PhysicsIslands::remove_joint (islands/mod.rs:749-812: unlinked from its island, constraints_removed += 1), JointGraph::remove_joint, and the WakeIslands([island])
 * it queues when that island sleeps; result as for avn_islands_wake_body.  renumber_joints: new_index[old] = new joint index | 0xFFFFFFFF (removed): joint ids are
 * the host's array indices, and the host compacted its joint array. */
AVN_API avn_status AVN_FN(islands_joint_remove)(avn_island_manager* m, uint32_t joint);
AVN_API avn_status AVN_FN(islands_renumber_joints)(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old);
typedef struct avn_islands_stats {
    uint32_t n_islands, n_sleeping_islands, n_bodies, n_sleeping_bodies;
    uint32_t merges, splits;           /* totals since creation */
    uint32_t split_candidate;          /* IslandId or 0xFFFFFFFF */
    uint32_t sleeping_pairs;           /* edges with ContactEdgeFlags::SLEEPING */
} avn_islands_stats;
AVN_API avn_status AVN_FN(islands_stats_get)(avn_island_manager* m, avn_islands_stats* out);
/* per body [n_bodies]: IslandId (slab key; 0xFFFFFFFF = no node), the NEXT body of the island's list (0xFFFFFFFF = tail), 1 = the body's
 * island sleeps, constraints_removed of the body's island.  Any pointer may be NULL. */
AVN_API avn_status AVN_FN(islands_state)(avn_island_manager* m, uint32_t n_bodies, uint32_t* island_of_body, uint32_t* next_in_island, uint8_t* island_sleeping,
                                         uint32_t* constraints_removed_of_body_island);

/* (2) the closed loop with persistent islands and sleeping.  `p` as for avn_sleep_update (thresholds, time_to_sleep, delta_secs, optional
 * per-body arrays -- copied); p = NULL switches it off (every island is woken first).  Needs avn_pipeline_enable; islands start as one per
 * body that owns a SolverBody, merged by the uploaded joints in upload order.  From then on avn_step keeps N = the awake bodies. */
AVN_API avn_status AVN_FN(sleeping_enable)(avn_world* w, const avn_sleep_params* p);
typedef struct avn_sleeping_stats {
    avn_islands_stats islands;
    uint32_t n_awake_bodies;           /* bodies that owned a SolverBody in the step just taken */
    uint32_t last_islands_slept, last_islands_woken, last_manifolds_popped, last_manifolds_pushed;
    double last_host_ms;               /* host time of the island bookkeeping of the last step */
} avn_sleeping_stats;
AVN_API avn_status AVN_FN(sleeping_stats_get)(avn_world* w, avn_sleeping_stats* out);
typedef struct avn_sleeping_out {
    uint32_t* island;          /* [n_bodies] IslandId, 0xFFFFFFFF = no node */
    uint32_t* next_in_island;  /* [n_bodies] */
    uint8_t* sleeping;         /* [n_bodies] 1 = the body has the Sleeping component */
    float* sleep_timer;        /* [n_bodies] */
} avn_sleeping_out;
AVN_API avn_status AVN_FN(sleeping_state_get)(avn_world* w, const avn_sleeping_out* out);
/* WakeBody for the listed bodies (a host that moved or kicked a sleeping body: wake_on_changed, sleeping.rs:556-604) */
AVN_API avn_status AVN_FN(wake_bodies)(avn_world* w, const uint32_t* bodies, size_t n);

/* Despawn INSIDE the closed loop (avn_pipeline_enable; round 4 -- rounds 1-3 refused it and the host had to restart the loop, losing warm starts
 * and the ContactId history).  What the reference does when entities with a RigidBody / Collider are despawned between two steps
 * (collision/narrow_phase/mod.rs:399-457 remove_collider, :459-560 remove_body_on / remove_collider_on; contact_types/contact_graph.rs:641-700
 * ContactGraph::remove_collider_with; data_structures/stable_graph.rs:251-315 remove_node_with / remove_edge; dynamics/solver/islands/mod.rs:
 * 1336-1400 BodyIslandNode::on_remove), in the order given:
 *   1. every collider of `collider_entities` (a collider despawned on its own), then every body of `bodies` with all of its colliders (upload
 *      order = RigidBodyColliders order): the collider's contact edges are walked in the ContactGraph's edge-list order -- outgoing edges
 *      (the collider is collider1) newest first, then incoming edges newest first --, a TOUCHING pair's constraint handles are popped from the
 *      ConstraintGraph in that order (swap_remove: the order decides where the other handles of the colour lists end up) and its contact is
 *      unlinked from its island (constraints_removed += 1); every edge leaves ContactGraph::pair_set and its ContactId returns to the IdPool
 *      (lowest free id first at the next allocation);
 *   2. with avn_sleeping_enable on: the body leaves its island's body list (an island left empty is removed), then the queued
 *      WakeIslands([the island the body was in]) runs;
 *   3. the remaining bodies are RENUMBERED by stable compaction (body i becomes i - #removed below i) in everything the library holds: contact
 *      rows, colour masks, joints, the island manager, sleep timers.  A joint that names a removed body is refused (AVN_ERR_STATE: upload the
 *      joints without it first).
 * The rigid-body and collider COMPONENTS are the host's: after avn_despawn it must call avn_bodies_upload (the remaining bodies, new numbering)
 * and avn_colliders_upload (the remaining colliders in their previous relative order; `body` in the new numbering) before the next avn_step --
 * the interval of a removed collider is dropped in place (update_aabb_intervals' retain_mut, collision/broad_phase.rs:230-279), the others keep
 * their order.  Until both have happened avn_step returns AVN_ERR_STATE. */
typedef struct avn_despawn_list {
    uint32_t struct_size;
    uint32_t n_colliders;
    const uint32_t* collider_entities;   /* Entity::index() of colliders despawned WITHOUT their body, in despawn order */
    uint32_t n_bodies;
    const uint32_t* bodies;              /* body indices (current numbering) despawned WITH all their colliders, in despawn order; no duplicates */
    /* round 5 (a caller built against the round-4 header passes the smaller struct_size and no joints): joints that leave, as indices into the LAST
     * avn_joints_upload, in despawn order.  They leave FIRST: remove_joint_from_graph for each (dynamics/solver/joint_graph/plugin.rs:163-194:
     * PhysicsIslands::remove_joint -> constraints_removed += 1, JointGraph::remove_joint, WakeIslands for a sleeping island).  Every joint that names a
     * despawned body must be listed (the reference would keep such a joint, dangling, and skip it in every system; the library holds no dangling joint).
     * The library's joint set shrinks to the remaining joints in their previous order; the host uploads exactly that set (avn_joints_upload) before the
     * next avn_step, bodies in the new numbering. */
    uint32_t n_joints;
    const uint32_t* joints;
} avn_despawn_list;
#define AVN_DESPAWN_LIST_SIZE_R4 (offsetof(avn_despawn_list, n_joints))
AVN_API avn_status AVN_FN(despawn)(avn_world* w, const avn_despawn_list* d);

/* ---- the DEVICE closed loop sharded by islands over several worlds / ranks (round 6) ------------------------------------------------------
 * avn_shard_* above is the host model: replicated integer bookkeeping in host C++, each rank's physics through the host pipeline mode.  This is the same idea with
 * the bookkeeping where avn_pipeline_enable(1) keeps it -- on the device -- and one exchange per step:
 *   * every rank holds EVERY body and collider (global indices) and runs the whole front of the step on them: AABBs, sweep-and-prune, IdPool, narrow phase, the status
 *     loop, greedy colouring, swap_remove replay.  These are pure functions of the bodies' components, which are equal on every rank, so pair sequences, ContactIds,
 *     colours and list positions ARE the single world's on every rank without a word exchanged (the reference's IdPool hands out the lowest free id and
 *     ConstraintGraph::pop_manifold moves the LAST handle of a list into the hole -- data_structures/id_pool.rs:31-40, constraint_graph.rs:245-296 -- so they cannot
 *     be sharded without changing results);
 *   * a body is SIMULATED by exactly one rank (`body_owner`): only there it owns a SolverBody, and that rank's solver takes its share of every GraphColor's
 *     manifold_handles -- the global list restricted to the manifolds of its own bodies, relative order kept (what the overflow colour's serial solve needs);
 *   * after the solver every rank's own bodies' Position / Rotation / LinearVelocity / AngularVelocity go to all the others: ONE all-gather per step, issued by the
 *     library on the world's stream through ncclAllGather when avn_comm_init has run (inside avn_step), or moved by the host between steps with
 *     avn_dshard_bodies_pack / _unpack (several worlds in one process, gloo).  The host still reads only the step's counters.
 * Islands of different ranks must not touch: a manifold between bodies of two owners fails the step (AVN_ERR_STATE) -- avn_bounds_exchange sees it coming, the level-1
 * re-partition handles it.  Not combined with avn_sleeping_enable, avn_despawn or the island blocks (the solver runs its colour launches); results are bit-identical
 * to the single world (tests/test_gpu_dshard.py, tests/test_dshard_cpu.py). */
typedef struct avn_dshard_config {
    size_t struct_size;          /* sizeof(avn_dshard_config) */
    uint32_t n_ranks, rank;
    const int32_t* body_owner;   /* [n_bodies] the rank that simulates the body; -1 = nobody (static bodies: replicated) */
} avn_dshard_config;
/* after avn_pipeline_enable(1), before the first step (or between steps with an unchanged body count); cfg = NULL switches it off */
AVN_API avn_status AVN_FN(dshard_enable)(avn_world* w, const avn_dshard_config* cfg);
/* host-mediated exchange: this rank's bodies (ascending index) as 4 records of 4 scalars each -- (Position, inv mass) (Rotation) (LinearVelocity, gravity scale)
 * (AngularVelocity, linear damping) -- in the world's scalar type; *bytes = n_own x 16 scalars.  unpack writes the records of rank `from_rank`'s bodies. */
AVN_API avn_status AVN_FN(dshard_bodies_pack)(avn_world* w, void* out, size_t cap_bytes, size_t* bytes);
AVN_API avn_status AVN_FN(dshard_bodies_unpack)(avn_world* w, uint32_t from_rank, const void* in, size_t bytes);
typedef struct avn_dshard_stats {
    uint32_t n_ranks, rank, own_bodies, own_manifolds;   /* this rank's share of the step just taken */
    uint32_t global_manifolds;                          /* constraint handles of the replicated ConstraintGraph */
    uint32_t exchanges;                                 /* all-gathers the library issued itself (avn_comm_init) since avn_dshard_enable */
    uint64_t bytes_sent_per_step;                       /* own bodies x 16 scalars */
} avn_dshard_stats;
AVN_API avn_status AVN_FN(dshard_stats_get)(avn_world* w, avn_dshard_stats* out);

/* ---- the closed loop sharded by ISLANDS: replicated integer bookkeeping (round 5; host C++, no device needed) ---------------------------------------------
 * Islands spread over ranks, results equal to the single world's bit for bit.  The reference's IdPool hands out the LOWEST free ContactId in the broad
 * phase's GLOBAL emission order (data_structures/id_pool.rs:31-40, collision/broad_phase.rs:387-388), NarrowPhase::update walks status changes in
 * ascending id (collision/narrow_phase/system_param.rs:141-145) and pop_manifold's swap_remove moves a colour's LAST handle into the hole
 * (dynamics/solver/constraint_graph.rs:245-296): one island's ids, colours and list positions depend on what other islands did.  So every rank
 * REPLAYS everything that is integer from flat arrays the ranks all-gather, and runs only its own islands' physics through the low-level calls above:
 *
 *   per step, rank r:   avn_run_system(UPDATE_AABB), (COLLECT_COLLISION_PAIRS); avn_aabbs_download, avn_pairs_get
 *     all-gather        (global collider slot, min.x) of the colliders whose body r owns [12 B each], avn_shard_pair of r's new pairs [24 B each]
 *     avn_shard_phase2  -> the global interval order (a stable sort of last frame's order: broad_phase.rs:373-387), ids in the global emission order
 *                       -> avn_shard_new_local_pairs -> avn_contact_pairs_add; avn_shard_active -> avn_active_pairs_set; avn_run_system(NARROW_PHASE)
 *     all-gather        avn_contact_changes_get [16 B per change]
 *     avn_shard_phase3  -> every rank's changes on the replicated ConstraintGraph in ascending id; avn_shard_removed_local -> avn_contact_pairs_remove;
 *                          avn_shard_handles(local) -> avn_manifold_handles_upload (the global colour lists restricted to r's pairs: relative order
 *                          kept, which is all the overflow colour's serial solve needs); avn_run_system(SOLVER)
 *
 * Bodies of different ranks must not come into AABB contact (avn_bounds_exchange detects it, the level-1 re-partition handles it).  Body indices in
 * avn_shard_pair are GLOBAL (the single world's); collider entities are global by nature. */
typedef struct avn_shard avn_shard;
typedef struct avn_shard_pair { uint32_t collider1, collider2; int32_t body1, body2; uint32_t flags; uint32_t owner; } avn_shard_pair;   /* owner = the rank that emitted it */
typedef struct avn_shard_stats { uint32_t pairs_added, pairs_removed, pushes, pops, next_id, n_free, last_status_changes, reserved; } avn_shard_stats;
AVN_API avn_status AVN_FN(shard_create)(uint32_t n_colliders, const uint32_t* collider_entities /* every collider of the single world, upload order */, uint32_t rank, avn_shard** out);
AVN_API void AVN_FN(shard_destroy)(avn_shard* s);
AVN_API const char* AVN_FN(shard_last_error)(const avn_shard* s);
AVN_API avn_status AVN_FN(shard_phase2)(avn_shard* s, const uint32_t* key_collider /* global slots */, const double* key_min_x, size_t n_keys, const avn_shard_pair* pairs, size_t n_pairs);
AVN_API avn_status AVN_FN(shard_new_local_pairs)(avn_shard* s, const uint32_t** contact_id, const uint32_t** collider1, const uint32_t** collider2, const uint32_t** flags, size_t* n);
AVN_API avn_status AVN_FN(shard_active)(avn_shard* s, const uint32_t** contact_id, size_t* n);
AVN_API avn_status AVN_FN(shard_phase3)(avn_shard* s, const avn_contact_change* changes /* all ranks', any order */, size_t n);
AVN_API avn_status AVN_FN(shard_removed_local)(avn_shard* s, const uint32_t** contact_id, size_t* n);
AVN_API avn_status AVN_FN(shard_handles)(avn_shard* s, int global /* 0: this rank's restriction, 1: the single world's lists */, uint32_t* color_offsets /* [AVN_GRAPH_COLOR_COUNT + 1] */, const uint32_t** contact_id, size_t* n);
AVN_API avn_status AVN_FN(shard_stats_get)(avn_shard* s, avn_shard_stats* out);

/* Union of the ColliderAabbs (after AVN_SYS_UPDATE_AABB) of all colliders on NON-static bodies of this world, as
 * doubles: the per-rank bound exchanged between ranks to detect islands of different ranks coming into AABB contact.
 * Empty worlds return min = +inf, max = -inf. */
AVN_API avn_status AVN_FN(dynamic_bounds)(avn_world* w, double aabb_min[3], double aabb_max[3]);

#ifdef __cplusplus
}
#endif
#endif /* AVIAN_MI355X_H */
