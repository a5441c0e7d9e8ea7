"""ctypes binding of the C ABI in ``include/avian_mi355x.h``.

The header can be implemented by more than one library (the HIP product uses the prefix ``avn_``; the test
suite loads a second, CPU implementation under another prefix); :class:`Library` is parameterised by
path + prefix so that tests can drive both through identical calls.  Nothing in this package knows where
any other implementation lives.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

GRAPH_COLOR_COUNT = 24
COLOR_OVERFLOW_INDEX = 23
DYNAMIC_COLOR_COUNT = 20
MAX_MANIFOLD_POINTS = 4

RB_DYNAMIC, RB_STATIC, RB_KINEMATIC = 0, 1, 2
BODY_SLEEPING, BODY_DISABLED, BODY_CUSTOM_VEL, BODY_CUSTOM_POS = 1, 2, 4, 8
SB_KINEMATIC, SB_GYROSCOPIC = 1 << 6, 1 << 7
SHAPE_CUBOID, SHAPE_BALL = 0, 1
COLLIDER_SENSOR, COLLIDER_EVENTS, COLLIDER_FILTER_PAIRS, COLLIDER_MODIFY_CONTACTS, COLLIDER_SWEPT_CCD = 1, 2, 4, 8, 16
MANIFOLD_GENERATES_CONSTRAINTS = 1
PAIR_CONTACT_EVENTS, PAIR_MODIFY_CONTACTS, PAIR_GENERATE_CONSTRAINTS, PAIR_NEEDS_CUSTOM_FILTER = 1, 2, 4, 8

STATUS_NAMES = {0: "AVN_OK", 1: "AVN_ERR_BAD_ARG", 2: "AVN_ERR_HIP", 3: "AVN_ERR_OOM", 4: "AVN_ERR_CAPACITY",
                5: "AVN_ERR_NO_DEVICE", 6: "AVN_ERR_STATE"}

# avn_system ids (same order as the header)
SYSTEMS = [
    "UPDATE_AABB", "COLLECT_COLLISION_PAIRS", "PREPARE_SOLVER_BODIES", "PREPARE_JOINTS",
    "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS", "INTEGRATE_VELOCITIES", "WARM_START",
    "SOLVE_CONTACTS_BIAS", "INTEGRATE_POSITIONS", "SOLVE_CONTACTS_RELAX", "XPBD_SOLVE",
    "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING", "CLEAR_VELOCITY_INCREMENTS", "SOLVE_RESTITUTION",
    "WRITEBACK_SOLVER_BODIES", "STORE_CONTACT_IMPULSES", "SUBSTEP", "SOLVER", "NARROW_PHASE",
]
SYS = {name: i for i, name in enumerate(SYSTEMS)}

vp = C.c_void_p


class AvnError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class avn_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("scalar_bits", C.c_uint32), ("device", C.c_int32), ("substeps", C.c_uint32),
        ("dt_ns", C.c_uint64), ("gravity", C.c_double * 3), ("length_unit", C.c_double),
        ("contact_damping_ratio", C.c_double), ("contact_frequency_factor", C.c_double),
        ("max_overlap_solve_speed", C.c_double), ("warm_start_coefficient", C.c_double),
        ("restitution_threshold", C.c_double), ("restitution_iterations", C.c_uint32),
        ("match_contacts", C.c_uint32), ("default_speculative_margin", C.c_double), ("contact_tolerance", C.c_double),
        ("solver_iterations", C.c_uint32), ("use_graph", C.c_uint32),
    ]


class avn_bodies(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "position", "rotation", "linear_velocity", "angular_velocity", "inv_mass", "inv_inertia_local",
        "center_of_mass", "linear_damping", "angular_damping", "gravity_scale", "accel_linear", "accel_angular",
        "max_linear_speed", "max_angular_speed", "rb_type", "locked_axes", "dominance", "body_flags")]


class avn_bodies_out(C.Structure):
    _fields_ = [(n, vp) for n in ("position", "rotation", "linear_velocity", "angular_velocity")]


class avn_solver_bodies_out(C.Structure):
    _fields_ = [(n, vp) for n in (
        "linear_velocity", "angular_velocity", "delta_position", "delta_rotation", "flags", "inv_mass",
        "inv_inertia_world", "dominance", "linear_increment", "angular_increment", "linear_damping_rhs",
        "angular_damping_rhs")]


class avn_manifolds(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "color_offsets", "body1", "body2", "normal", "friction", "restitution", "tangent_velocity", "point_count",
        "manifold_flags", "anchor1", "anchor2", "penetration", "normal_speed", "warm_start_normal_impulse",
        "warm_start_tangent_impulse")]


class avn_impulses_out(C.Structure):
    _fields_ = [(n, vp) for n in ("warm_start_normal_impulse", "warm_start_tangent_impulse", "normal_impulse")]


class avn_constraints_out(C.Structure):
    _fields_ = [(n, vp) for n in (
        "point_count", "relative_dominance", "tangent1", "anchor1", "initial_separation", "normal_impulse",
        "total_impulse", "normal_effective_mass", "tangent_impulse", "tangent_effective_inverse_mass",
        "softness_non_dynamic")]


class avn_distance_joints(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "body1", "body2", "local_anchor1", "local_anchor2", "limit_min", "limit_max", "compliance",
        "damping_linear", "damping_angular", "collision_disabled")]


class avn_joints_out(C.Structure):
    _fields_ = [(n, vp) for n in ("world_r1", "world_r2", "center_difference", "total_lagrange", "force",
                                  "total_rotation_lagrange", "torque")]


JOINT_FIXED, JOINT_REVOLUTE, JOINT_SPHERICAL, JOINT_PRISMATIC, JOINT_DISTANCE = 0, 1, 2, 3, 4
JOINT_TYPE_COUNT = 5
JOINT_HAS_LIMIT1, JOINT_HAS_LIMIT2 = 1, 2


class avn_joints(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "joint_type", "body1", "body2", "local_anchor1", "local_anchor2", "local_basis1", "local_basis2", "axis",
        "limit_min", "limit_max", "limit2_min", "limit2_max", "limit_flags", "compliance", "damping_linear",
        "damping_angular", "collision_disabled")]


class avn_colliders(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "entity_index", "body", "shape", "half_extents", "memberships", "filters", "collider_flags",
        "collision_margin", "speculative_margin")]


MAX_QUERY_POINTS = 16


class avn_shape_pairs(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in (
        "shape1", "half_extents1", "position1", "rotation1", "shape2", "half_extents2", "position2", "rotation2",
        "prediction_distance")]


class avn_query_manifolds_out(C.Structure):
    _fields_ = [(n, vp) for n in ("point_count", "normal", "anchor1", "anchor2", "point", "penetration", "feature_id1", "feature_id2")]


CP_TOUCHING, CP_GENERATE_CONSTRAINTS, CP_STATIC1, CP_STATIC2, CP_MODIFY_CONTACTS, CP_CONTACT_EVENTS = 1, 2, 4, 8, 16, 32
CP_DISJOINT_AABB, CP_STARTED_TOUCHING, CP_STOPPED_TOUCHING, CP_STARTED_GENERATING_CONSTRAINTS = 1 << 8, 1 << 9, 1 << 10, 1 << 11
COMBINE_AVERAGE, COMBINE_GEOMETRIC_MEAN, COMBINE_MIN, COMBINE_MULTIPLY, COMBINE_MAX = 1, 2, 3, 4, 5


class avn_collider_materials(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in ("friction", "restitution", "friction_combine", "restitution_combine")]


class avn_contact_pairs(C.Structure):
    _fields_ = [("count", C.c_uint32)] + [(n, vp) for n in ("contact_id", "collider1", "collider2", "pair_flags")]


class avn_contacts_out(C.Structure):
    _fields_ = [(n, vp) for n in ("flags", "point_count", "normal", "friction", "restitution", "anchor1", "anchor2", "penetration", "normal_speed",
                                  "warm_start_normal_impulse", "warm_start_tangent_impulse", "normal_impulse", "feature_id1", "feature_id2")]


class avn_contacts_in(C.Structure):
    _fields_ = avn_contacts_out._fields_   # the same pointers, read instead of written


class avn_pipeline_stats(C.Structure):
    _fields_ = [("pairs_added", C.c_uint64), ("pairs_removed", C.c_uint64), ("manifolds_pushed", C.c_uint64), ("manifolds_popped", C.c_uint64),
                ("active_pairs", C.c_uint32), ("manifolds", C.c_uint32), ("last_status_changes", C.c_uint32), ("last_overflow_manifolds", C.c_uint32),
                ("last_host_ms", C.c_double)]


CHANGE_DTYPE = np.dtype([("contact_id", "<u4"), ("flags", "<u4"), ("manifold_count_change", "<i4"), ("manifold_count", "<u4")])


class avn_pair(C.Structure):
    _fields_ = [("collider1", C.c_uint32), ("collider2", C.c_uint32), ("body1", C.c_int32), ("body2", C.c_int32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class avn_islands_in(C.Structure):
    _fields_ = [("n_bodies", C.c_uint32), ("rb_type", vp), ("center_x", vp), ("n_edges", C.c_uint32),
                ("edge_body1", vp), ("edge_body2", vp), ("n_ranks", C.c_uint32)]


class avn_timers(C.Structure):
    _fields_ = [("broad_phase_ms", C.c_double), ("prepare_ms", C.c_double), ("substeps_ms", C.c_double),
                ("finalize_ms", C.c_double), ("step_ms", C.c_double), ("contact_constraint_count", C.c_uint32),
                ("pair_count", C.c_uint32), ("kernel_launches", C.c_uint32), ("bias_pass_launches", C.c_uint32), ("bias_pass_ms", C.c_double),
                ("island_blocks", C.c_uint32), ("side_island_bodies", C.c_uint32)]


class avn_diagnostics(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("prepare_constraints_ms", "update_velocity_increments_ms", "integrate_velocities_ms", "warm_start_ms",
                                          "solve_constraints_ms", "integrate_positions_ms", "relax_velocities_ms", "apply_restitution_ms", "finalize_ms",
                                          "store_impulses_ms", "swept_ccd_ms", "substeps_ms", "broad_phase_ms", "narrow_phase_ms")] + \
               [(n, C.c_uint32) for n in ("contact_constraint_count", "contact_count", "per_system_valid", "reserved0")]


class avn_slab_in(C.Structure):
    _fields_ = [("n_colliders", C.c_uint32), ("aabb_min_x", vp), ("aabb_max_x", vp), ("prev_order", vp), ("n_prev", C.c_uint32),
                ("n_ranks", C.c_uint32), ("rank", C.c_uint32)]


class avn_level2_in(C.Structure):
    _fields_ = [("n_bodies", C.c_uint32), ("rb_type", vp), ("center_x", vp), ("n_manifolds", C.c_uint32), ("body1", vp), ("body2", vp),
                ("color_offsets", vp), ("n_ranks", C.c_uint32)]


class avn_sleep_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("time_to_sleep", C.c_float), ("linear_threshold", C.c_float), ("angular_threshold", C.c_float),
                ("delta_secs", C.c_float), ("length_unit", C.c_double), ("body_linear_threshold", vp), ("body_angular_threshold", vp),
                ("body_sleeping_disabled", vp)]


class avn_sleep_stats(C.Structure):
    _fields_ = [("n_islands", C.c_uint32), ("n_island_bodies", C.c_uint32), ("n_sleeping_bodies", C.c_uint32), ("n_awake_bodies", C.c_uint32),
                ("n_resting_islands", C.c_uint32), ("n_resting_bodies", C.c_uint32), ("n_waking_islands", C.c_uint32), ("n_waking_bodies", C.c_uint32)]


class avn_islands_stats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_islands", "n_sleeping_islands", "n_bodies", "n_sleeping_bodies", "merges", "splits", "split_candidate", "sleeping_pairs")]


class avn_despawn_list(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_colliders", C.c_uint32), ("collider_entities", vp), ("n_bodies", C.c_uint32), ("bodies", vp),
                ("n_joints", C.c_uint32), ("joints", vp)]


class avn_dshard_config(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("body_owner", vp)]


class avn_dshard_stats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_ranks", "rank", "own_bodies", "own_manifolds", "global_manifolds", "exchanges")] + [("bytes_sent_per_step", C.c_uint64)]


class avn_sleeping_stats(C.Structure):
    _fields_ = [("islands", avn_islands_stats)] + [(n, C.c_uint32) for n in ("n_awake_bodies", "last_islands_slept", "last_islands_woken", "last_manifolds_popped", "last_manifolds_pushed")] + [("last_host_ms", C.c_double)]


class avn_sleeping_out(C.Structure):
    _fields_ = [("island", vp), ("next_in_island", vp), ("sleeping", vp), ("sleep_timer", vp)]


class avn_sleep_out(C.Structure):
    _fields_ = [("sleep_timer", vp), ("island", vp), ("island_rests", vp), ("island_wakes", vp)]


class avn_level2_joints(C.Structure):
    _fields_ = [("n_joints", C.c_uint32), ("body1", C.POINTER(C.c_int32)), ("body2", C.POINTER(C.c_int32)), ("joint_type", C.POINTER(C.c_uint8)), ("damped", C.c_uint32)]


class avn_halo_plan(C.Structure):
    _fields_ = [("n_peers", C.c_uint32), ("peer_rank", vp), ("send_offsets", vp), ("send_bodies", vp), ("recv_offsets", vp), ("recv_bodies", vp)]


class avn_level2_rank(C.Structure):
    _fields_ = [("n_bodies", C.c_uint32), ("bodies", C.POINTER(C.c_int32)), ("n_manifolds", C.c_uint32), ("manifolds", C.POINTER(C.c_uint32)),
                ("color_offsets", C.POINTER(C.c_uint32)), ("halo", avn_halo_plan)]


PAIR_DTYPE = np.dtype([("collider1", "<u4"), ("collider2", "<u4"), ("body1", "<i4"), ("body2", "<i4"),
                       ("flags", "<u4"), ("reserved", "<u4")])

# ---- host shapes (include/avian_mi355x.h "host shapes"): the two AnyCollider callbacks -----------------------------------------------------
SHAPE_CUBOID, SHAPE_BALL, SHAPE_HOST = 0, 1, 2
HOST_SHAPE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)   # avn_host_aabb_fn == avn_host_manifolds_fn in shape


class avn_host_shape_stats(C.Structure):
    _fields_ = [("host_colliders", C.c_uint32), ("last_aabb_queries", C.c_uint32), ("last_manifold_queries", C.c_uint32), ("last_manifolds_with_points", C.c_uint32),
                ("bytes_to_host", C.c_uint64), ("bytes_from_host", C.c_uint64), ("last_callback_ms", C.c_double)]


class avn_collider_transforms(C.Structure):
    _fields_ = [("count", C.c_uint32), ("is_child", C.c_void_p), ("translation", C.c_void_p), ("rotation", C.c_void_p)]


# ---- collision hooks (include/avian_mi355x.h "collision hooks"): CollisionHooks::filter_pairs / modify_contacts as callbacks -----------------
HOOK_FILTER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p)                 # avn_filter_pairs_fn
HOOK_MODIFY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)                 # avn_modify_contacts_fn
HOOK_PAIR_DTYPE = np.dtype([("index", "<u4"), ("collider1", "<u4"), ("collider2", "<u4")])          # avn_hook_pair


class avn_collision_hook_stats(C.Structure):
    _fields_ = [("last_filter_queries", C.c_uint32), ("last_filter_rejected", C.c_uint32), ("last_modify_queries", C.c_uint32), ("last_modify_rejected", C.c_uint32),
                ("bytes_to_host", C.c_uint64), ("bytes_from_host", C.c_uint64), ("last_callback_ms", C.c_double)]


def hook_contact_dtype(bits: int):
    """numpy view of avn_hook_contact_fNN."""
    S = "<f4" if bits == 32 else "<f8"
    return np.dtype([("contact_id", "<u4"), ("collider1", "<u4"), ("collider2", "<u4"), ("body1", "<u4"), ("body2", "<u4"), ("flags", "<u4"), ("touching", "<u4"),
                     ("manifold_count", "<u4"), ("point_count", "<u4"), ("reserved", "<u4"), ("normal", S, 3), ("friction", S), ("restitution", S), ("tangent_velocity", S, 3),
                     ("anchor1", S, (4, 3)), ("anchor2", S, (4, 3)), ("penetration", S, 4), ("normal_speed", S, 4), ("feature_id1", "<u4", 4), ("feature_id2", "<u4", 4)])


def host_shape_dtypes(bits: int):
    """numpy views of avn_host_aabb_query_fNN, avn_host_aabb_fNN, avn_host_manifold_query_fNN, avn_host_manifold_fNN."""
    S = "<f4" if bits == 32 else "<f8"
    aq = np.dtype([("collider", "<u4"), ("swept", "<u4"), ("start_position", S, 3), ("start_rotation", S, 4), ("end_position", S, 3), ("end_rotation", S, 4)])
    ab = np.dtype([("min", S, 3), ("max", S, 3)])
    mq = np.dtype([("contact_id", "<u4"), ("collider1", "<u4"), ("collider2", "<u4"), ("reserved", "<u4"), ("position1", S, 3), ("rotation1", S, 4), ("position2", S, 3),
                   ("rotation2", S, 4), ("max_contact_distance", S)])
    mm = np.dtype([("point_count", "<u4")] + ([("reserved", "<u4")] if bits == 64 else []) +
                  [("normal", S, 3), ("anchor1", S, (MAX_QUERY_POINTS, 3)), ("penetration", S, MAX_QUERY_POINTS), ("feature_id1", "<u4", MAX_QUERY_POINTS), ("feature_id2", "<u4", MAX_QUERY_POINTS)])
    assert (aq.itemsize, ab.itemsize, mq.itemsize, mm.itemsize) == ((64, 24, 76, 400) if bits == 32 else (120, 48, 136, 672))
    return aq, ab, mq, mm


# every symbol include/avian_mi355x.h declares (without prefix)
ABI_SYMBOLS = [
    "world_create", "world_destroy", "last_error", "config_set", "bodies_upload", "bodies_download",
    "solver_bodies_download", "manifolds_upload", "impulses_download", "constraints_download",
    "distance_joints_upload", "joints_download", "colliders_upload", "existing_pairs_upload", "pairs_get",
    "aabbs_download", "run_system", "step", "synchronize", "timers_get", "diagnostics_get", "halo_plan_upload", "run_color_pass", "halo_pack", "halo_unpack", "comm_unique_id", "comm_init", "islands_get", "sleep_update", "sleep_get", "sleep_reset", "level2_plan_create", "level2_plan_destroy", "level2_plan_rank", "level2_plan_rank_overflow", "halo_overflow_levels_upload", "host_shapes_set", "host_shape_stats_get", "collision_hooks_set", "collision_hook_stats_get", "collider_transforms_upload", "local_accelerations_upload", "halo_joint_slot_set", "level2_plan_create_joints", "level2_plan_rank_joints", "slab_select", "interval_orders_merge", "profile_system", "pair_key", "constraint_graph_create",
    "constraint_graph_destroy", "constraint_graph_push", "constraint_graph_pop", "constraint_graph_lists",
    "islands_partition", "dynamic_bounds", "constraint_graph_push_batch", "joints_upload", "contact_manifolds",
    "collider_materials_upload", "contact_pairs_add", "contact_pairs_remove", "active_pairs_set", "contact_changes_get", "manifold_handles_upload",
    "contacts_download", "contacts_upload", "pipeline_enable", "pipeline_stats_get", "pipeline_handles_get", "pipeline_new_pair_ids_get",
    "islands_create", "islands_destroy", "islands_body_add", "islands_collider_add", "islands_joint_add", "islands_pair_add", "islands_status_change",
    "islands_flush_wake", "islands_split_candidate", "islands_split_candidate_adjacency", "islands_split_join", "islands_sleeping_systems", "islands_wake_body", "islands_sleep_body", "islands_last_result",
    "islands_collider_remove", "islands_body_remove", "islands_renumber_bodies", "islands_joint_remove", "islands_renumber_joints",
    "shard_create", "shard_destroy", "shard_last_error", "shard_phase2", "shard_new_local_pairs", "shard_active", "shard_phase3", "shard_removed_local", "shard_handles", "shard_stats_get",
    "islands_stats_get", "islands_state", "sleeping_enable", "sleeping_stats_get", "sleeping_state_get", "wake_bodies", "bounds_exchange", "despawn",
    "dshard_enable", "dshard_bodies_pack", "dshard_bodies_unpack", "dshard_stats_get",
]


class Library:
    """A loaded implementation of the ABI (``prefix`` = ``avn_`` for the product)."""

    def __init__(self, path: str, prefix: str = "avn_"):
        self.path = path
        self.prefix = prefix
        self.dll = C.CDLL(path)
        missing = [s for s in ABI_SYMBOLS if not hasattr(self.dll, prefix + s)]
        if missing and not os.environ.get("AVN_AB_OLDER_LIBRARY"):
            raise ImportError(f"{path}: missing ABI symbols {missing}")
        # (AVN_AB_OLDER_LIBRARY=1 with AVN_LIB_PATH: an A/B run against an OLDER build of the library, whose ABI lacks the newest entry points)
        class _Absent:
            argtypes = None; restype = None
        f = (lambda name: self.fn(name) if hasattr(self.dll, prefix + name) else _Absent()) if missing else self.fn
        f("world_create").argtypes = [C.POINTER(avn_config), C.POINTER(vp)]
        f("world_destroy").argtypes = [vp]
        f("world_destroy").restype = None
        f("last_error").argtypes = [vp]
        f("last_error").restype = C.c_char_p
        for name in ("config_set", "bodies_upload", "bodies_download", "solver_bodies_download", "manifolds_upload",
                     "impulses_download", "constraints_download", "distance_joints_upload", "joints_upload", "joints_download",
                     "colliders_upload", "timers_get", "diagnostics_get"):
            f(name).argtypes = [vp, vp]
        f("existing_pairs_upload").argtypes = [vp, vp, C.c_size_t]
        f("pairs_get").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("aabbs_download").argtypes = [vp, vp, vp, vp, C.POINTER(C.c_size_t)]
        f("run_system").argtypes = [vp, C.c_int]
        f("halo_plan_upload").argtypes = [vp, vp]
        f("run_color_pass").argtypes = [vp, C.c_int, C.c_uint32]
        f("halo_pack").argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_size_t)]
        f("halo_unpack").argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_size_t]
        f("slab_select").argtypes = [C.POINTER(avn_slab_in), vp, vp, C.POINTER(C.c_uint32), vp, C.POINTER(C.c_uint32)]
        f("interval_orders_merge").argtypes = [C.c_uint32, vp, vp, vp, vp, C.POINTER(C.c_uint32)]
        f("level2_plan_create").argtypes = [C.POINTER(avn_level2_in), C.POINTER(vp)]
        f("level2_plan_destroy").argtypes = [vp]
        f("level2_plan_destroy").restype = None
        f("level2_plan_rank").argtypes = [vp, C.c_uint32, C.POINTER(avn_level2_rank)]
        f("level2_plan_rank_overflow").argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32))]
        f("halo_overflow_levels_upload").argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        f("halo_joint_slot_set").argtypes = [vp, C.c_uint32, C.c_uint32]
        f("level2_plan_create_joints").argtypes = [C.POINTER(avn_level2_in), C.POINTER(avn_level2_joints), C.POINTER(vp)]
        f("level2_plan_rank_joints").argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        f("host_shapes_set").argtypes = [vp, HOST_SHAPE_FN, HOST_SHAPE_FN, vp]
        f("host_shape_stats_get").argtypes = [vp, C.POINTER(avn_host_shape_stats)]
        f("collision_hooks_set").argtypes = [vp, HOOK_FILTER_FN, HOOK_MODIFY_FN, vp]
        f("collider_transforms_upload").argtypes = [vp, C.POINTER(avn_collider_transforms)]
        f("local_accelerations_upload").argtypes = [vp, C.c_uint32, vp, vp]
        f("collision_hook_stats_get").argtypes = [vp, C.POINTER(avn_collision_hook_stats)]
        f("islands_get").argtypes = [vp, vp, C.POINTER(C.c_uint32)]
        f("sleep_update").argtypes = [vp, C.POINTER(avn_sleep_params), C.POINTER(avn_sleep_stats)]
        f("sleep_get").argtypes = [vp, C.POINTER(avn_sleep_out)]
        f("sleep_reset").argtypes = [vp, vp, C.c_size_t]
        f("sleeping_enable").argtypes = [vp, C.POINTER(avn_sleep_params)]
        f("sleeping_stats_get").argtypes = [vp, C.POINTER(avn_sleeping_stats)]
        f("sleeping_state_get").argtypes = [vp, C.POINTER(avn_sleeping_out)]
        f("wake_bodies").argtypes = [vp, vp, C.c_size_t]
        f("despawn").argtypes = [vp, C.POINTER(avn_despawn_list)]
        f("dshard_enable").argtypes = [vp, vp]
        f("dshard_bodies_pack").argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        f("dshard_bodies_unpack").argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        f("dshard_stats_get").argtypes = [vp, C.POINTER(avn_dshard_stats)]
        for n_ in ("dshard_enable", "dshard_bodies_pack", "dshard_bodies_unpack", "dshard_stats_get"): f(n_).restype = C.c_int
        f("bounds_exchange").argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32), vp, C.c_uint32, C.POINTER(C.c_uint32)]
        f("comm_unique_id").argtypes = [vp]
        f("comm_init").argtypes = [vp, vp, C.c_int, C.c_int]
        f("profile_system").argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        f("step").argtypes = [vp]
        f("synchronize").argtypes = [vp]
        f("pair_key").argtypes = [C.c_uint32, C.c_uint32]
        f("pair_key").restype = C.c_uint64
        f("constraint_graph_create").argtypes = [C.c_uint32, C.POINTER(vp)]
        f("constraint_graph_destroy").argtypes = [vp]
        f("constraint_graph_destroy").restype = None
        f("constraint_graph_push").argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        f("constraint_graph_push").restype = C.c_int32
        f("constraint_graph_push_batch").argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
        f("constraint_graph_pop").argtypes = [vp, C.c_uint64]
        f("constraint_graph_lists").argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        f("islands_partition").argtypes = [C.POINTER(avn_islands_in), vp, vp, C.POINTER(C.c_uint32)]
        f("dynamic_bounds").argtypes = [vp, vp, vp]
        f("contact_manifolds").argtypes = [vp, vp, vp]
        f("collider_materials_upload").argtypes = [vp, vp]
        f("contact_pairs_add").argtypes = [vp, vp]
        f("contact_pairs_remove").argtypes = [vp, vp, C.c_size_t]
        f("active_pairs_set").argtypes = [vp, vp, C.c_size_t]
        f("contact_changes_get").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("manifold_handles_upload").argtypes = [vp, vp, vp]
        f("contacts_download").argtypes = [vp, vp, C.c_size_t, vp]
        f("contacts_upload").argtypes = [vp, vp, C.c_size_t, vp]
        f("pipeline_enable").argtypes = [vp, C.c_int]
        f("pipeline_stats_get").argtypes = [vp, vp]
        f("pipeline_handles_get").argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("pipeline_new_pair_ids_get").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]

    def fn(self, name: str):
        return getattr(self.dll, self.prefix + name)

    def _declare_islands(self):
        if getattr(self, "_islands_declared", False):
            return
        f = self.fn
        f("islands_create").restype = vp; f("islands_create").argtypes = []
        f("islands_destroy").restype = None; f("islands_destroy").argtypes = [vp]
        for name, args in (("islands_body_add", [vp, C.c_uint32]), ("islands_collider_add", [vp, C.c_uint32, C.c_uint32]),
                           ("islands_joint_add", [vp, C.c_uint32, C.c_uint32, C.c_uint32]), ("islands_pair_add", [vp, C.c_uint32, C.c_uint32, C.c_uint32]),
                           ("islands_status_change", [vp, C.c_uint32, C.c_uint32, C.c_uint32]), ("islands_flush_wake", [vp]), ("islands_split_candidate", [vp]), ("islands_split_candidate_adjacency", [vp, vp, vp, C.c_uint32, vp]), ("islands_split_join", [vp]),
                           ("islands_sleeping_systems", [vp, vp, vp, C.c_uint32, C.c_float]), ("islands_wake_body", [vp, C.c_uint32]), ("islands_sleep_body", [vp, C.c_uint32]),
                           ("islands_last_result", [vp, vp]), ("islands_stats_get", [vp, vp]), ("islands_state", [vp, C.c_uint32, vp, vp, vp, vp]),
                           ("islands_collider_remove", [vp, C.c_uint32]), ("islands_body_remove", [vp, C.c_uint32]), ("islands_renumber_bodies", [vp, vp, C.c_uint32]),
                           ("islands_joint_remove", [vp, C.c_uint32]), ("islands_renumber_joints", [vp, vp, C.c_uint32])):
            f(name).restype = C.c_int; f(name).argtypes = args
        self._islands_declared = True

    def pair_key(self, a: int, b: int) -> int:
        return int(self.fn("pair_key")(a, b))

    def slab_select(self, aabb_min_x, aabb_max_x, prev_order, n_ranks: int, rank: int, want_next: bool = True):
        """``avn_slab_select``: (local collider indices in the persistent order, owned mask, next frame's prev_order or None)."""
        mn = np.ascontiguousarray(aabb_min_x, np.float64); mx = np.ascontiguousarray(aabb_max_x, np.float64)
        po = None if prev_order is None else np.ascontiguousarray(prev_order, np.uint32)
        n = len(mn)
        inp = avn_slab_in(n, _ptr(mn), _ptr(mx), _ptr(po), 0 if po is None else len(po), int(n_ranks), int(rank))
        local = np.zeros(max(n, 1), np.uint32); owned = np.zeros(max(n, 1), np.uint8); nxt = np.zeros(max(n, 1), np.uint32)
        nl, nn = C.c_uint32(), C.c_uint32()
        st = self.fn("slab_select")(C.byref(inp), _ptr(local), _ptr(owned), C.byref(nl), _ptr(nxt) if want_next else None, C.byref(nn))
        if st != 0:
            raise AvnError(st, "slab_select: bad arguments")
        return local[: nl.value].astype(np.int64), owned[: nl.value].astype(bool), (nxt[: nn.value].astype(np.int64) if want_next else None)

    def interval_orders_merge(self, entity_lists, key_lists):
        """``avn_interval_orders_merge``: the merged global interval order (entity indices)."""
        ents = [np.ascontiguousarray(e, np.uint32) for e in entity_lists]; keys = [np.ascontiguousarray(k, np.float64) for k in key_lists]
        L = len(ents)
        ep = (vp * L)(*[e.ctypes.data_as(vp) for e in ents]); kp = (vp * L)(*[k.ctypes.data_as(vp) for k in keys])
        lens = np.array([len(e) for e in ents], np.uint32)
        out = np.zeros(max(int(lens.sum()), 1), np.uint32)
        n = C.c_uint32()
        st = self.fn("interval_orders_merge")(L, ep, kp, _ptr(lens), _ptr(out), C.byref(n))
        if st != 0:
            raise AvnError(st, "interval_orders_merge: bad arguments")
        return out[: n.value].astype(np.int64)

    def level2_plan(self, rb_type, center_x, body1, body2, color_offsets, n_ranks: int, joints=None):
        """``avn_level2_plan_*``: per rank a dict(bodies, manifolds, color_offsets, peers, send_offsets, send_bodies, recv_offsets, recv_bodies, n_overflow_levels,
        overflow_level).  Exchange slots: 23 colours + n_overflow_levels (1 unless an overflow-colour manifold touches a shared body)."""
        rb = np.ascontiguousarray(rb_type, np.uint8); cx = np.ascontiguousarray(center_x, np.float64)
        b1 = np.ascontiguousarray(body1, np.int32); b2 = np.ascontiguousarray(body2, np.int32); co = np.ascontiguousarray(color_offsets, np.uint32)
        inp = avn_level2_in(len(rb), _ptr(rb), _ptr(cx), len(b1), _ptr(b1), _ptr(b2), _ptr(co), int(n_ranks))
        h = vp()
        if joints is not None:   # (joint body1, joint body2, joint_type, damped): avn_level2_plan_create_joints
            jb1 = np.ascontiguousarray(joints[0], np.int32); jb2 = np.ascontiguousarray(joints[1], np.int32); jt = np.ascontiguousarray(joints[2], np.uint8)
            jn = avn_level2_joints(len(jb1), C.cast(_ptr(jb1), C.POINTER(C.c_int32)), C.cast(_ptr(jb2), C.POINTER(C.c_int32)), C.cast(_ptr(jt), C.POINTER(C.c_uint8)), 1 if joints[3] else 0)
            st = self.fn("level2_plan_create_joints")(C.byref(inp), C.byref(jn), C.byref(h))
        else:
            st = self.fn("level2_plan_create")(C.byref(inp), C.byref(h))
        if st != 0:
            raise AvnError(st, "level2_plan: refused (bad indices)")
        try:
            out = []
            for r in range(n_ranks):
                k = avn_level2_rank()
                st = self.fn("level2_plan_rank")(h, r, C.byref(k))
                if st != 0:
                    raise AvnError(st, "level2_plan_rank")
                arr = lambda p, n, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t).copy() if n else np.zeros(0, t)
                npeers = k.halo.n_peers
                nlev = C.c_uint32(); lev_p = C.POINTER(C.c_uint32)()
                st = self.fn("level2_plan_rank_overflow")(h, r, C.byref(nlev), C.byref(lev_p))
                if st != 0:
                    raise AvnError(st, "level2_plan_rank_overflow")
                nj = C.c_uint32(); jp = C.POINTER(C.c_uint32)(); jslot = C.c_uint32(); gj = C.c_uint32()
                st = self.fn("level2_plan_rank_joints")(h, r, C.byref(nj), C.byref(jp), C.byref(jslot), C.byref(gj))
                if st != 0:
                    raise AvnError(st, "level2_plan_rank_joints")
                nl = (23 + nlev.value + jslot.value) * npeers + 1 if npeers else 1
                n_ovf = int(k.color_offsets[24] - k.color_offsets[23])
                so = arr(C.cast(k.halo.send_offsets, C.POINTER(C.c_uint32)), nl, np.uint32); ro = arr(C.cast(k.halo.recv_offsets, C.POINTER(C.c_uint32)), nl, np.uint32)
                out.append(dict(bodies=arr(k.bodies, k.n_bodies, np.int64), manifolds=arr(k.manifolds, k.n_manifolds, np.int64), color_offsets=arr(k.color_offsets, 25, np.uint32),
                                peers=arr(C.cast(k.halo.peer_rank, C.POINTER(C.c_int32)), npeers, np.int32), send_offsets=so,
                                send_bodies=arr(C.cast(k.halo.send_bodies, C.POINTER(C.c_int32)), int(so[-1]), np.int32), recv_offsets=ro,
                                recv_bodies=arr(C.cast(k.halo.recv_bodies, C.POINTER(C.c_int32)), int(ro[-1]), np.int32),
                                n_overflow_levels=int(nlev.value), overflow_level=arr(lev_p, n_ovf, np.uint32),
                                joints=arr(jp, nj.value, np.int64) if joints is not None else None, joint_slot=bool(jslot.value), global_joints=bool(gj.value)))
            return out
        finally:
            self.fn("level2_plan_destroy")(h)

    def comm_unique_id(self) -> bytes:
        """``avn_comm_unique_id``: the RCCL rendezvous token rank 0 creates and hands to the other ranks (any side channel)."""
        buf = (C.c_uint8 * 128)()
        st = self.fn("comm_unique_id")(buf)
        if st != 0:
            raise AvnError(st, (self.fn("last_error")(None) or b"comm_unique_id failed").decode())
        return bytes(buf)

    def islands_partition(self, rb_type, center_x, edge_body1, edge_body2, n_ranks: int):
        """``avn_islands_partition``: returns (island_of_body, rank_of_body, n_islands)."""
        rb = np.ascontiguousarray(rb_type, np.uint8); cx = np.ascontiguousarray(center_x, np.float64)
        e1 = np.ascontiguousarray(edge_body1, np.int32); e2 = np.ascontiguousarray(edge_body2, np.int32)
        n = len(rb)
        isl = np.empty(n, np.int32); rk = np.empty(n, np.int32); cnt = C.c_uint32(0)
        a = avn_islands_in(n, _ptr(rb), _ptr(cx), len(e1), _ptr(e1), _ptr(e2), int(n_ranks))
        st = self.fn("islands_partition")(C.byref(a), _ptr(isl), _ptr(rk), C.byref(cnt))
        if st != 0:
            raise AvnError(st, "islands_partition")
        return isl, rk, int(cnt.value)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(vp)


def default_config(scalar_bits: int = 32, substeps: int = 6, dt: float = 1.0 / 60.0, **kw) -> avn_config:
    """``SolverConfig::default`` / ``NarrowPhaseConfig::default`` / ``Gravity::default`` of the reference."""
    cfg = avn_config()
    cfg.struct_size = C.sizeof(avn_config)
    cfg.scalar_bits = scalar_bits
    cfg.device = 0
    cfg.substeps = substeps
    # Duration::from_secs_f64 rounds to the nearest nanosecond
    cfg.dt_ns = int(round(dt * 1e9))
    cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = 0.0, -9.81, 0.0
    cfg.length_unit = 1.0
    cfg.contact_damping_ratio = 10.0
    cfg.contact_frequency_factor = 1.5
    cfg.max_overlap_solve_speed = 4.0
    cfg.warm_start_coefficient = 1.0
    cfg.restitution_threshold = 1.0
    cfg.restitution_iterations = 1
    cfg.match_contacts = 1
    cfg.default_speculative_margin = float(np.finfo(np.float64).max)
    cfg.contact_tolerance = 0.005
    cfg.solver_iterations = 1
    cfg.use_graph = 1
    for k, v in kw.items():
        if k == "gravity":
            for i in range(3):
                cfg.gravity[i] = float(v[i])
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


class World:
    """One physics world behind the ABI.  Arrays are numpy, converted to the world's scalar type."""

    def __init__(self, lib: Library, cfg: avn_config):
        self.lib = lib
        self.cfg = cfg
        self.dtype = np.float32 if cfg.scalar_bits == 32 else np.float64
        self.n_bodies = 0
        self.n_manifolds = 0
        self.n_joints = 0
        self.n_colliders = 0
        h = vp()
        st = lib.fn("world_create")(C.byref(cfg), C.byref(h))
        if st != 0:
            msg = lib.fn("last_error")(None)
            raise AvnError(st, (msg or b"").decode())
        self.handle = h

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, st: int):
        if st != 0:
            msg = self.lib.fn("last_error")(self.handle)
            raise AvnError(st, (msg or b"").decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fn("world_destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _s(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a

    @staticmethod
    def _i(a, dt):
        return None if a is None else np.ascontiguousarray(a, dtype=dt)

    def config_set(self, cfg: avn_config):
        self.cfg = cfg
        self._check(self.lib.fn("config_set")(self.handle, C.byref(cfg)))

    # -- bodies ---------------------------------------------------------------------------------
    def bodies_upload(self, position, rotation, linear_velocity, angular_velocity, inv_mass, inv_inertia_local,
                      rb_type, center_of_mass=None, linear_damping=None, angular_damping=None, gravity_scale=None,
                      accel_linear=None, accel_angular=None, max_linear_speed=None, max_angular_speed=None,
                      locked_axes=None, dominance=None, body_flags=None):
        n = len(np.asarray(inv_mass).reshape(-1))
        keep = [self._s(position, (n, 3)), self._s(rotation, (n, 4)), self._s(linear_velocity, (n, 3)),
                self._s(angular_velocity, (n, 3)), self._s(inv_mass, (n,)), self._s(inv_inertia_local, (n, 6)),
                self._s(center_of_mass), self._s(linear_damping), self._s(angular_damping), self._s(gravity_scale),
                self._s(accel_linear), self._s(accel_angular), self._s(max_linear_speed), self._s(max_angular_speed),
                self._i(rb_type, np.uint8), self._i(locked_axes, np.uint8), self._i(dominance, np.int8),
                self._i(body_flags, np.uint8)]
        b = avn_bodies(n, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("bodies_upload")(self.handle, C.byref(b)))
        self.n_bodies = n

    def bodies_download(self, out=None):
        """``out``: a dict of preallocated arrays of the world's scalar type (e.g. page-locked staging) to download into."""
        n, dt = self.n_bodies, self.dtype
        if out is None:
            out = {"position": np.empty((n, 3), dt), "rotation": np.empty((n, 4), dt),
                   "linear_velocity": np.empty((n, 3), dt), "angular_velocity": np.empty((n, 3), dt)}
        o = avn_bodies_out(*[_ptr(out[k]) for k in ("position", "rotation", "linear_velocity", "angular_velocity")])
        self._check(self.lib.fn("bodies_download")(self.handle, C.byref(o)))
        return out

    def solver_bodies_download(self):
        n, dt = self.n_bodies, self.dtype
        out = {"linear_velocity": np.empty((n, 3), dt), "angular_velocity": np.empty((n, 3), dt),
               "delta_position": np.empty((n, 3), dt), "delta_rotation": np.empty((n, 4), dt),
               "flags": np.empty(n, np.uint32), "inv_mass": np.empty(n, dt), "inv_inertia_world": np.empty((n, 6), dt),
               "dominance": np.empty(n, np.int16), "linear_increment": np.empty((n, 3), dt),
               "angular_increment": np.empty((n, 3), dt), "linear_damping_rhs": np.empty(n, dt),
               "angular_damping_rhs": np.empty(n, dt)}
        o = avn_solver_bodies_out(*[_ptr(out[k]) for k, _ in avn_solver_bodies_out._fields_])
        self._check(self.lib.fn("solver_bodies_download")(self.handle, C.byref(o)))
        return out

    # -- manifolds --------------------------------------------------------------------------------
    def manifolds_upload(self, color_offsets, body1, body2, normal, friction, restitution, point_count, anchor1,
                         anchor2, penetration, normal_speed, tangent_velocity=None, manifold_flags=None,
                         warm_start_normal_impulse=None, warm_start_tangent_impulse=None):
        m = len(np.asarray(body1).reshape(-1))
        keep = [self._i(color_offsets, np.uint32), self._i(body1, np.int32), self._i(body2, np.int32),
                self._s(normal, (m, 3)), self._s(friction, (m,)), self._s(restitution, (m,)),
                self._s(tangent_velocity), self._i(point_count, np.uint8), self._i(manifold_flags, np.uint8),
                self._s(anchor1, (m, 4, 3)), self._s(anchor2, (m, 4, 3)), self._s(penetration, (m, 4)),
                self._s(normal_speed, (m, 4)), self._s(warm_start_normal_impulse), self._s(warm_start_tangent_impulse)]
        assert keep[0].shape == (GRAPH_COLOR_COUNT + 1,)
        s = avn_manifolds(m, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("manifolds_upload")(self.handle, C.byref(s)))
        self.n_manifolds = m

    def impulses_download(self, out=None):
        m, dt = self.n_manifolds, self.dtype
        if out is None:
            out = {"warm_start_normal_impulse": np.zeros((m, 4), dt), "warm_start_tangent_impulse": np.zeros((m, 4, 2), dt),
                   "normal_impulse": np.zeros((m, 4), dt)}
        o = avn_impulses_out(*[_ptr(out[k]) for k, _ in avn_impulses_out._fields_])
        self._check(self.lib.fn("impulses_download")(self.handle, C.byref(o)))
        return out

    def constraints_download(self):
        m, dt = self.n_manifolds, self.dtype
        out = {"point_count": np.zeros(m, np.uint8), "relative_dominance": np.zeros(m, np.int16),
               "tangent1": np.zeros((m, 3), dt), "anchor1": np.zeros((m, 4, 3), dt),
               "initial_separation": np.zeros((m, 4), dt), "normal_impulse": np.zeros((m, 4), dt),
               "total_impulse": np.zeros((m, 4), dt), "normal_effective_mass": np.zeros((m, 4), dt),
               "tangent_impulse": np.zeros((m, 4, 2), dt), "tangent_effective_inverse_mass": np.zeros((m, 4, 3), dt),
               "softness_non_dynamic": np.zeros(m, np.uint8)}
        o = avn_constraints_out(*[_ptr(out[k]) for k, _ in avn_constraints_out._fields_])
        self._check(self.lib.fn("constraints_download")(self.handle, C.byref(o)))
        return out

    # -- joints -------------------------------------------------------------------------------------
    def distance_joints_upload(self, body1, body2, local_anchor1, local_anchor2, limit_min, limit_max, compliance,
                               damping_linear=None, damping_angular=None, collision_disabled=None):
        j = len(np.asarray(body1).reshape(-1))
        keep = [self._i(body1, np.int32), self._i(body2, np.int32), self._s(local_anchor1, (j, 3)),
                self._s(local_anchor2, (j, 3)), self._s(limit_min, (j,)), self._s(limit_max, (j,)),
                self._s(compliance, (j,)), self._s(damping_linear), self._s(damping_angular),
                self._i(collision_disabled, np.uint8)]
        s = avn_distance_joints(j, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("distance_joints_upload")(self.handle, C.byref(s)))
        self.n_joints = j

    def joints_upload(self, joint_type, body1, body2, local_anchor1, local_anchor2, compliance, local_basis1=None,
                      local_basis2=None, axis=None, limit_min=None, limit_max=None, limit2_min=None, limit2_max=None,
                      limit_flags=None, damping_linear=None, damping_angular=None, collision_disabled=None):
        """``avn_joints_upload``: every XPBD joint type (``compliance`` is [J, 3], see the header)."""
        j = len(np.asarray(body1).reshape(-1))
        keep = [self._i(joint_type, np.uint8), self._i(body1, np.int32), self._i(body2, np.int32),
                self._s(local_anchor1, (j, 3)), self._s(local_anchor2, (j, 3)), self._s(local_basis1), self._s(local_basis2),
                self._s(axis), self._s(limit_min), self._s(limit_max), self._s(limit2_min), self._s(limit2_max),
                self._i(limit_flags, np.uint8), self._s(compliance, (j, 3)), self._s(damping_linear), self._s(damping_angular),
                self._i(collision_disabled, np.uint8)]
        s = avn_joints(j, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("joints_upload")(self.handle, C.byref(s)))
        self.n_joints = j

    def joints_download(self):
        j, dt = self.n_joints, self.dtype
        out = {k: np.zeros((j, 3), dt) for k, _ in avn_joints_out._fields_}
        o = avn_joints_out(*[_ptr(out[k]) for k, _ in avn_joints_out._fields_])
        self._check(self.lib.fn("joints_download")(self.handle, C.byref(o)))
        return out

    # -- broad phase -----------------------------------------------------------------------------------
    def colliders_upload(self, entity_index, body, shape, half_extents, memberships=None, filters=None,
                         collider_flags=None, collision_margin=None, speculative_margin=None):
        c = len(np.asarray(entity_index).reshape(-1))
        keep = [self._i(entity_index, np.uint32), self._i(body, np.int32), self._i(shape, np.uint8),
                self._s(half_extents, (c, 3)), self._i(memberships, np.uint32), self._i(filters, np.uint32),
                self._i(collider_flags, np.uint8), self._s(collision_margin), self._s(speculative_margin)]
        s = avn_colliders(c, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("colliders_upload")(self.handle, C.byref(s)))
        self.n_colliders = c

    def existing_pairs_upload(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._check(self.lib.fn("existing_pairs_upload")(self.handle, _ptr(keys), keys.size))

    def collider_transforms_upload(self, is_child=None, translation=None, rotation=None):
        """``avn_collider_transforms_upload``: ColliderTransform of the colliders that are CHILD entities of their rigid body (after every colliders_upload).
        No arguments: no collider is a child."""
        if is_child is None:
            self._check(self.lib.fn("collider_transforms_upload")(self.handle, None))
            return
        c = self.n_colliders
        keep = [self._i(is_child, np.uint8), self._s(translation, (c, 3)), self._s(rotation, (c, 4))]
        s = avn_collider_transforms(c, *[_ptr(a) for a in keep])
        self._check(self.lib.fn("collider_transforms_upload")(self.handle, C.byref(s)))

    def local_accelerations_upload(self, linear=None, angular=None):
        """``avn_local_accelerations_upload``: AccumulatedLocalAcceleration per body (what ConstantLocalForce & co. and ``Forces::apply_local_*`` accumulated),
        applied by every substep in front of integrate_velocities.  No arguments: no body has one."""
        if linear is None and angular is None:
            self._check(self.lib.fn("local_accelerations_upload")(self.handle, 0, None, None))
            return
        n = len(linear) if linear is not None else len(angular)
        keep = [None if linear is None else self._s(linear, (n, 3)), None if angular is None else self._s(angular, (n, 3))]
        self._check(self.lib.fn("local_accelerations_upload")(self.handle, n, _ptr(keep[0]), _ptr(keep[1])))

    def pairs_get(self) -> np.ndarray:
        p, n = vp(), C.c_size_t()
        self._check(self.lib.fn("pairs_get")(self.handle, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PAIR_DTYPE)
        buf = (C.c_char * (n.value * PAIR_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=PAIR_DTYPE).copy()

    def aabbs_download(self):
        c, dt = self.n_colliders, self.dtype
        mn, mx = np.zeros((c, 3), dt), np.zeros((c, 3), dt)
        ents = np.zeros(c, np.uint32)
        n = C.c_size_t()
        self._check(self.lib.fn("aabbs_download")(self.handle, _ptr(mn), _ptr(mx), _ptr(ents), C.byref(n)))
        return mn, mx, ents[: n.value]

    # -- narrow phase, part 2: the ContactGraph side -----------------------------------------------------------------
    def collider_materials_upload(self, friction=None, restitution=None, friction_combine=None, restitution_combine=None):
        dt, c = self.dtype, self.n_colliders
        a = lambda x, t: None if x is None else np.ascontiguousarray(np.broadcast_to(np.asarray(x, t), (c,)))
        fr, re, fc, rc = a(friction, dt), a(restitution, dt), a(friction_combine, np.uint8), a(restitution_combine, np.uint8)
        m = avn_collider_materials(c, _ptr(fr), _ptr(re), _ptr(fc), _ptr(rc))
        self._check(self.lib.fn("collider_materials_upload")(self.handle, C.byref(m)))

    def contact_pairs_add(self, contact_id, collider1, collider2, pair_flags):
        ids, c1, c2, fl = (np.ascontiguousarray(x, np.uint32) for x in (contact_id, collider1, collider2, pair_flags))
        p = avn_contact_pairs(len(ids), _ptr(ids), _ptr(c1), _ptr(c2), _ptr(fl))
        self._check(self.lib.fn("contact_pairs_add")(self.handle, C.byref(p)))

    def contact_pairs_remove(self, contact_id):
        ids = np.ascontiguousarray(contact_id, np.uint32)
        self._check(self.lib.fn("contact_pairs_remove")(self.handle, _ptr(ids), ids.size))

    def active_pairs_set(self, contact_id):
        ids = np.ascontiguousarray(contact_id, np.uint32)
        self._check(self.lib.fn("active_pairs_set")(self.handle, _ptr(ids), ids.size))

    def contact_changes_get(self) -> np.ndarray:
        p, n = vp(), C.c_size_t()
        self._check(self.lib.fn("contact_changes_get")(self.handle, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, CHANGE_DTYPE)
        buf = (C.c_char * (n.value * CHANGE_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=CHANGE_DTYPE).copy()

    def manifold_handles_upload(self, color_offsets, contact_id):
        off = np.ascontiguousarray(color_offsets, np.uint32); ids = np.ascontiguousarray(contact_id, np.uint32)
        assert off.size == GRAPH_COLOR_COUNT + 1 and int(off[-1]) == ids.size
        self._check(self.lib.fn("manifold_handles_upload")(self.handle, _ptr(off), _ptr(ids)))
        self.n_manifolds = ids.size

    def contacts_download(self, contact_id):
        ids = np.ascontiguousarray(contact_id, np.uint32)
        n, dt = ids.size, self.dtype
        out = dict(flags=np.zeros(n, np.uint32), point_count=np.zeros(n, np.uint8), normal=np.zeros((n, 3), dt), friction=np.zeros(n, dt),
                   restitution=np.zeros(n, dt), anchor1=np.zeros((n, 4, 3), dt), anchor2=np.zeros((n, 4, 3), dt), penetration=np.zeros((n, 4), dt),
                   normal_speed=np.zeros((n, 4), dt), warm_start_normal_impulse=np.zeros((n, 4), dt), warm_start_tangent_impulse=np.zeros((n, 4, 2), dt),
                   normal_impulse=np.zeros((n, 4), dt), feature_id1=np.zeros((n, 4), np.uint32), feature_id2=np.zeros((n, 4), np.uint32))
        o = avn_contacts_out(*[_ptr(out[k]) for k in ("flags", "point_count", "normal", "friction", "restitution", "anchor1", "anchor2", "penetration",
                                                      "normal_speed", "warm_start_normal_impulse", "warm_start_tangent_impulse", "normal_impulse",
                                                      "feature_id1", "feature_id2")])
        self._check(self.lib.fn("contacts_download")(self.handle, _ptr(ids), n, C.byref(o)))
        return out

    _CONTACT_ROW_FIELDS = (("flags", np.uint32, ()), ("point_count", np.uint8, ()), ("normal", None, (3,)), ("friction", None, ()), ("restitution", None, ()),
                           ("anchor1", None, (4, 3)), ("anchor2", None, (4, 3)), ("penetration", None, (4,)), ("normal_speed", None, (4,)),
                           ("warm_start_normal_impulse", None, (4,)), ("warm_start_tangent_impulse", None, (4, 2)), ("normal_impulse", None, (4,)),
                           ("feature_id1", np.uint32, (4,)), ("feature_id2", np.uint32, (4,)))

    def contacts_upload(self, contact_id, rows):
        """``avn_contacts_upload``: rows in the layout ``contacts_download`` returns (a dict of arrays), for contact ids that exist."""
        ids = np.ascontiguousarray(contact_id, np.uint32)
        n = ids.size
        keep = []
        for name, dt, shape in self._CONTACT_ROW_FIELDS:
            a = np.ascontiguousarray(rows[name], dt or self.dtype)
            assert a.shape == (n,) + shape, f"contacts_upload: {name} has shape {a.shape}, expected {(n,) + shape}"
            keep.append(a)
        o = avn_contacts_in(*[_ptr(a) for a in keep])
        self._check(self.lib.fn("contacts_upload")(self.handle, _ptr(ids), n, C.byref(o)))

    # -- standalone closed loop (the library keeps IdPool / ContactGraph bookkeeping / ConstraintGraph itself) ------------
    def pipeline_enable(self, on: bool = True, host_bookkeeping: bool = False):
        """avn_pipeline_enable: 1 = the closed loop with the ContactGraph / ConstraintGraph bookkeeping on the device (k_graph.hip),
        2 = the same loop with the host-side structures (kept for A/B runs), 0 = off."""
        self._check(self.lib.fn("pipeline_enable")(self.handle, (2 if host_bookkeeping else 1) if on else 0))

    def pipeline_stats(self) -> avn_pipeline_stats:
        st = avn_pipeline_stats()
        self._check(self.lib.fn("pipeline_stats_get")(self.handle, C.byref(st)))
        return st

    def pipeline_new_pair_ids(self) -> np.ndarray:
        """ContactIds of the last closed-loop step's new pairs, entry i for pair i of ``pairs_get`` (``avn_pipeline_new_pair_ids_get``)."""
        p, n = vp(), C.c_size_t()
        self._check(self.lib.fn("pipeline_new_pair_ids_get")(self.handle, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (n.value,)).copy() if n.value else np.zeros(0, np.uint32)

    def pipeline_handles(self):
        off = np.zeros(GRAPH_COLOR_COUNT + 1, np.uint32)
        p, n = vp(), C.c_size_t()
        self._check(self.lib.fn("pipeline_handles_get")(self.handle, _ptr(off), C.byref(p), C.byref(n)))
        if n.value == 0:
            return off, np.zeros(0, np.uint32)
        buf = (C.c_uint32 * n.value).from_address(p.value)
        return off, np.frombuffer(buf, dtype=np.uint32).copy()

    def contact_manifolds(self, shape1, half_extents1, position1, rotation1, shape2, half_extents2, position2, rotation2,
                          prediction_distance):
        """Batch ``contact_query::contact_manifolds``: dict of point_count [n], normal [n,3], anchor1/anchor2/point
        [n,16,3], penetration [n,16], feature_id1/2 [n,16]."""
        dt = self.dtype
        n = len(shape1)
        a = lambda x, w: np.ascontiguousarray(np.asarray(x, dt).reshape(n, w))
        s1 = np.ascontiguousarray(shape1, np.uint8); s2 = np.ascontiguousarray(shape2, np.uint8)
        h1, p1, r1 = a(half_extents1, 3), a(position1, 3), a(rotation1, 4)
        h2, p2, r2 = a(half_extents2, 3), a(position2, 3), a(rotation2, 4)
        pd = np.ascontiguousarray(np.broadcast_to(np.asarray(prediction_distance, dt), (n,)))
        out = dict(point_count=np.zeros(n, np.uint8), normal=np.zeros((n, 3), dt), anchor1=np.zeros((n, MAX_QUERY_POINTS, 3), dt),
                   anchor2=np.zeros((n, MAX_QUERY_POINTS, 3), dt), point=np.zeros((n, MAX_QUERY_POINTS, 3), dt),
                   penetration=np.zeros((n, MAX_QUERY_POINTS), dt), feature_id1=np.zeros((n, MAX_QUERY_POINTS), np.uint32),
                   feature_id2=np.zeros((n, MAX_QUERY_POINTS), np.uint32))
        pin = avn_shape_pairs(n, _ptr(s1), _ptr(h1), _ptr(p1), _ptr(r1), _ptr(s2), _ptr(h2), _ptr(p2), _ptr(r2), _ptr(pd))
        pout = avn_query_manifolds_out(*[_ptr(out[k]) for k in ("point_count", "normal", "anchor1", "anchor2", "point", "penetration",
                                                               "feature_id1", "feature_id2")])
        self._check(self.lib.fn("contact_manifolds")(self.handle, C.byref(pin), C.byref(pout)))
        return out

    # -- running -----------------------------------------------------------------------------------------
    def run_system(self, name: str):
        self._check(self.lib.fn("run_system")(self.handle, SYS[name]))

    def profile_system(self, name: str, repeats: int):
        """(total_ms, kernel_launches) of `repeats` back-to-back runs of a system, timed on the world's stream."""
        ms, n = C.c_double(), C.c_uint32()
        self._check(self.lib.fn("profile_system")(self.handle, SYS[name], repeats, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def dynamic_bounds(self):
        mn = np.empty(3, np.float64); mx = np.empty(3, np.float64)
        self._check(self.lib.fn("dynamic_bounds")(self.handle, _ptr(mn), _ptr(mx)))
        return mn, mx

    def step(self):
        self._check(self.lib.fn("step")(self.handle))

    def synchronize(self):
        self._check(self.lib.fn("synchronize")(self.handle))

    # -- level-2 sharding (one island over several worlds) -----------------------------------------------------------------
    # -- host shapes: AnyCollider::aabb_with_context / contact_manifolds_with_context as callbacks ------------------------------------------
    def host_shapes_set(self, aabb_fn, manifolds_fn):
        """``avn_host_shapes_set``.  aabb_fn(queries, out) and manifolds_fn(queries, out) get numpy structured arrays that VIEW the library's buffers
        (host_shape_dtypes): read `queries`, fill `out` in place.  None, None unregisters."""
        if aabb_fn is None:
            self._hs_keep = None
            self._check(self.lib.fn("host_shapes_set")(self.handle, None, None, None))
            return
        errors = self._hs_errors = []

        def wrap(fn, qi, oi):
            def cb(user, bits, n, q, o):
                try:
                    dt = host_shape_dtypes(int(bits))
                    qa = np.frombuffer((C.c_char * (n * dt[qi].itemsize)).from_address(q), dtype=dt[qi])
                    oa = np.frombuffer((C.c_char * (n * dt[oi].itemsize)).from_address(o), dtype=dt[oi])
                    fn(qa, oa)
                except BaseException as e:  # noqa: BLE001 -- an exception must not unwind through C; the caller re-raises it after the step
                    errors.append(e)
            return HOST_SHAPE_FN(cb)
        a, m = wrap(aabb_fn, 0, 1), wrap(manifolds_fn, 2, 3)
        self._hs_keep = (a, m)   # (ctypes callbacks must outlive the registration)
        self._check(self.lib.fn("host_shapes_set")(self.handle, a, m, None))

    # -- collision hooks: CollisionHooks::filter_pairs / modify_contacts as callbacks -----------------------------------------------------
    def collision_hooks_set(self, filter_fn=None, modify_fn=None):
        """``avn_collision_hooks_set``.  filter_fn(pairs, should_collide): `pairs` a HOOK_PAIR_DTYPE array (emission order), `should_collide` a uint8 array preset to 1,
        written in place.  modify_fn(contacts): a hook_contact_dtype array (ascending contact id) modified in place (touching = the hook's return value).  Both arrays
        VIEW the library's buffers.  None = the trait's default for that hook; None, None unregisters."""
        errors = self._hs_errors = getattr(self, "_hs_errors", [])

        def cb_filter(user, n, q, o):
            try:
                filter_fn(np.frombuffer((C.c_char * (n * HOOK_PAIR_DTYPE.itemsize)).from_address(q), dtype=HOOK_PAIR_DTYPE), np.frombuffer((C.c_char * n).from_address(o), dtype=np.uint8))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        def cb_modify(user, bits, n, r):
            try:
                dt = hook_contact_dtype(int(bits))
                modify_fn(np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(r), dtype=dt))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        f = HOOK_FILTER_FN(cb_filter) if filter_fn is not None else HOOK_FILTER_FN()
        m = HOOK_MODIFY_FN(cb_modify) if modify_fn is not None else HOOK_MODIFY_FN()
        self._hk_keep = (f, m)
        self._check(self.lib.fn("collision_hooks_set")(self.handle, f, m, None))

    def collision_hook_stats(self) -> avn_collision_hook_stats:
        st = avn_collision_hook_stats()
        self._check(self.lib.fn("collision_hook_stats_get")(self.handle, C.byref(st)))
        return st

    def host_shape_errors(self):
        """Exceptions raised inside the Python callbacks since the last call (a C caller cannot propagate them)."""
        e = list(getattr(self, "_hs_errors", []))
        if e:
            self._hs_errors.clear()
        return e

    def host_shape_stats(self) -> avn_host_shape_stats:
        st = avn_host_shape_stats()
        self._check(self.lib.fn("host_shape_stats_get")(self.handle, C.byref(st)))
        return st

    def halo_overflow_levels_upload(self, n_levels: int, level_of):
        """``avn_halo_overflow_levels_upload``: call BEFORE halo_plan_upload when the planner cut the overflow colour into levels."""
        lv = np.ascontiguousarray(level_of, np.uint32)
        self._check(self.lib.fn("halo_overflow_levels_upload")(self.handle, int(n_levels), _ptr(lv), len(lv)))

    def halo_joint_slot_set(self, joint_slot: bool, global_joints: bool):
        """``avn_halo_joint_slot_set``: call BEFORE halo_plan_upload (the joint slot is one more slot behind the colours and overflow levels, 16 scalars per body)."""
        self._check(self.lib.fn("halo_joint_slot_set")(self.handle, 1 if joint_slot else 0, 1 if global_joints else 0))
        self._halo_joint = bool(joint_slot)

    def halo_plan_upload(self, peers, send_offsets, send_bodies, recv_offsets, recv_bodies):
        peers = np.ascontiguousarray(peers, np.int32)
        so = np.ascontiguousarray(send_offsets, np.uint32); sb = np.ascontiguousarray(send_bodies, np.int32)
        ro = np.ascontiguousarray(recv_offsets, np.uint32); rb = np.ascontiguousarray(recv_bodies, np.int32)
        plan = avn_halo_plan(len(peers), _ptr(peers), _ptr(so), _ptr(sb), _ptr(ro), _ptr(rb))
        self._check(self.lib.fn("halo_plan_upload")(self.handle, C.byref(plan)))
        self._halo = (peers, so, ro)

    def run_color_pass(self, system: str, color: int):
        self._check(self.lib.fn("run_color_pass")(self.handle, SYSTEMS.index(system), int(color)))

    def halo_pack(self, color: int, peer: int) -> np.ndarray:
        peers, so, _ = self._halo
        n = int(so[color * len(peers) + peer + 1] - so[color * len(peers) + peer])
        joint = getattr(self, "_halo_joint", False) and len(peers) and color == (len(so) - 1) // len(peers) - 1   # the last slot of a plan with a joint slot
        out = np.zeros((n, 16 if joint else 8), self.dtype)
        cnt = C.c_size_t()
        self._check(self.lib.fn("halo_pack")(self.handle, int(color), int(peer), _ptr(out), C.byref(cnt)))
        assert cnt.value == n
        return out

    def halo_unpack(self, color: int, peer: int, rec: np.ndarray):
        rec = np.ascontiguousarray(rec, self.dtype)
        self._check(self.lib.fn("halo_unpack")(self.handle, int(color), int(peer), _ptr(rec), len(rec)))

    # -- islands and sleeping -----------------------------------------------------------------
    def islands_get(self):
        """``avn_islands_get``: (island label per body = lowest body index of its island, 0xFFFFFFFF for static bodies; island count)."""
        lab = np.zeros(self.n_bodies, np.uint32)
        n = C.c_uint32()
        self._check(self.lib.fn("islands_get")(self.handle, _ptr(lab), C.byref(n)))
        return lab, int(n.value)

    def sleep_update(self, delta_secs: float = 1.0 / 60.0, time_to_sleep: float = 0.5, linear_threshold: float = 0.15, angular_threshold: float = 0.15,
                     length_unit: float = 1.0, body_linear_threshold=None, body_angular_threshold=None, body_sleeping_disabled=None) -> avn_sleep_stats:
        """``avn_sleep_update``: update_sleeping_states + the decision of sleep_islands for the step just taken."""
        bl = None if body_linear_threshold is None else np.ascontiguousarray(body_linear_threshold, np.float32)
        ba = None if body_angular_threshold is None else np.ascontiguousarray(body_angular_threshold, np.float32)
        bd = None if body_sleeping_disabled is None else np.ascontiguousarray(body_sleeping_disabled, np.uint8)
        for a in (bl, ba, bd):
            assert a is None or len(a) == self.n_bodies
        p = avn_sleep_params(C.sizeof(avn_sleep_params), time_to_sleep, linear_threshold, angular_threshold, delta_secs, length_unit, _ptr(bl), _ptr(ba), _ptr(bd))
        st = avn_sleep_stats()
        self._check(self.lib.fn("sleep_update")(self.handle, C.byref(p), C.byref(st)))
        return st

    def sleep_get(self):
        n = self.n_bodies
        out = dict(sleep_timer=np.zeros(n, np.float32), island=np.zeros(n, np.uint32), island_rests=np.zeros(n, np.uint8), island_wakes=np.zeros(n, np.uint8))
        o = avn_sleep_out(_ptr(out["sleep_timer"]), _ptr(out["island"]), _ptr(out["island_rests"]), _ptr(out["island_wakes"]))
        self._check(self.lib.fn("sleep_get")(self.handle, C.byref(o)))
        return out

    def sleep_reset(self, bodies=None):
        if bodies is None or len(bodies) == 0:
            self._check(self.lib.fn("sleep_reset")(self.handle, None, 0))
        else:
            b = np.ascontiguousarray(bodies, np.uint32)
            self._check(self.lib.fn("sleep_reset")(self.handle, _ptr(b), len(b)))

    # -- persistent islands + sleeping ACTUATION in the closed loop (islands/mod.rs, islands/sleeping.rs) --------------------------------
    def sleeping_enable(self, on: bool = True, delta_secs: float = 1.0 / 60.0, time_to_sleep: float = 0.5, linear_threshold: float = 0.15, angular_threshold: float = 0.15,
                        length_unit: float = 1.0, body_linear_threshold=None, body_angular_threshold=None, body_sleeping_disabled=None):
        """``avn_sleeping_enable``: avn_step keeps persistent islands, puts resting ones to sleep and wakes them (needs pipeline_enable)."""
        if not on:
            self._check(self.lib.fn("sleeping_enable")(self.handle, None)); return
        bl = None if body_linear_threshold is None else np.ascontiguousarray(body_linear_threshold, np.float32)
        ba = None if body_angular_threshold is None else np.ascontiguousarray(body_angular_threshold, np.float32)
        bd = None if body_sleeping_disabled is None else np.ascontiguousarray(body_sleeping_disabled, np.uint8)
        for a in (bl, ba, bd):
            assert a is None or len(a) == self.n_bodies
        p = avn_sleep_params(C.sizeof(avn_sleep_params), time_to_sleep, linear_threshold, angular_threshold, delta_secs, length_unit, _ptr(bl), _ptr(ba), _ptr(bd))
        self._check(self.lib.fn("sleeping_enable")(self.handle, C.byref(p)))

    def sleeping_stats(self) -> "avn_sleeping_stats":
        st = avn_sleeping_stats()
        self._check(self.lib.fn("sleeping_stats_get")(self.handle, C.byref(st)))
        return st

    def sleeping_state(self):
        n = self.n_bodies
        out = dict(island=np.zeros(n, np.uint32), next_in_island=np.zeros(n, np.uint32), sleeping=np.zeros(n, np.uint8), sleep_timer=np.zeros(n, np.float32))
        o = avn_sleeping_out(_ptr(out["island"]), _ptr(out["next_in_island"]), _ptr(out["sleeping"]), _ptr(out["sleep_timer"]))
        self._check(self.lib.fn("sleeping_state_get")(self.handle, C.byref(o)))
        return out

    def wake_bodies(self, bodies):
        b = np.ascontiguousarray(bodies, np.uint32)
        self._check(self.lib.fn("wake_bodies")(self.handle, _ptr(b), len(b)))

    def despawn(self, bodies=(), collider_entities=(), joints=()):
        """``avn_despawn``: remove joints, bodies (with their colliders) and / or single colliders inside the closed loop -- ContactGraph edges in
        edge-list order, ConstraintGraph pops, IdPool, islands, renumbering.  Follow it with bodies_upload + colliders_upload (+ joints_upload) of what remains."""
        b = np.ascontiguousarray(bodies, np.uint32); c = np.ascontiguousarray(collider_entities, np.uint32); j = np.ascontiguousarray(joints, np.uint32)
        d = avn_despawn_list(C.sizeof(avn_despawn_list), len(c), _ptr(c) if len(c) else None, len(b), _ptr(b) if len(b) else None, len(j), _ptr(j) if len(j) else None)
        self._check(self.lib.fn("despawn")(self.handle, C.byref(d)))
        self.n_bodies -= len(b)
        self.n_joints = getattr(self, "n_joints", 0) - len(j)

    # -- the device closed loop sharded by islands (avn_dshard_*) -------------------------------------------------------------------------
    def dshard_enable(self, n_ranks: int, rank: int, body_owner):
        """``avn_dshard_enable``: this world simulates the bodies ``body_owner == rank`` and replicates everything else of the closed loop (None: off)."""
        if body_owner is None:
            self._check(self.lib.fn("dshard_enable")(self.handle, None)); return
        o = np.ascontiguousarray(body_owner, np.int32)
        assert len(o) == self.n_bodies
        c = avn_dshard_config(C.sizeof(avn_dshard_config), int(n_ranks), int(rank), _ptr(o))
        self._check(self.lib.fn("dshard_enable")(self.handle, C.byref(c)))
        self._dshard = (int(n_ranks), int(rank), o.copy())

    def dshard_bodies_pack(self) -> np.ndarray:
        """this rank's bodies after the step as [n_own, 16] scalars (Position | inv mass, Rotation, LinearVelocity | gravity scale, AngularVelocity | damping)"""
        n_ranks, rank, o = self._dshard
        out = np.zeros((int((o == rank).sum()), 16), self.dtype)
        nb = C.c_size_t()
        self._check(self.lib.fn("dshard_bodies_pack")(self.handle, _ptr(out) if out.size else _ptr(np.zeros(1, self.dtype)), out.nbytes, C.byref(nb)))
        assert nb.value == out.nbytes
        return out

    def dshard_bodies_unpack(self, from_rank: int, records: np.ndarray):
        r = np.ascontiguousarray(records, self.dtype)
        self._check(self.lib.fn("dshard_bodies_unpack")(self.handle, int(from_rank), _ptr(r) if r.size else None, r.nbytes))

    def dshard_stats(self) -> "avn_dshard_stats":
        st = avn_dshard_stats()
        self._check(self.lib.fn("dshard_stats_get")(self.handle, C.byref(st)))
        return st

    def bounds_exchange(self, max_ranks: int = 64):
        """``avn_bounds_exchange``: (bounds [n_ranks, 6], overlapping rank pairs [k, 2]) -- the library reduces this world's dynamic bounds on the
        device and all-gathers them over the communicator of comm_init (a world without one is its own only rank)."""
        b = np.zeros((max_ranks, 6), np.float64); ov = np.zeros((max_ranks * max_ranks, 2), np.uint32)
        n, k = C.c_uint32(), C.c_uint32()
        self._check(self.lib.fn("bounds_exchange")(self.handle, _ptr(b), max_ranks, C.byref(n), _ptr(ov), len(ov), C.byref(k)))
        return b[: n.value].copy(), ov[: min(k.value, len(ov))].astype(np.int64)

    def comm_init(self, unique_id: bytes, n_ranks: int, rank: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.fn("comm_init")(self.handle, buf, int(n_ranks), int(rank)))

    def diagnostics(self) -> avn_diagnostics:
        """SolverDiagnostics + CollisionDiagnostics of the last step (milliseconds)."""
        d = avn_diagnostics()
        self._check(self.lib.fn("diagnostics_get")(self.handle, C.byref(d)))
        return d

    def timers(self) -> avn_timers:
        t = avn_timers()
        self._check(self.lib.fn("timers_get")(self.handle, C.byref(t)))
        return t


SHARD_PAIR_DTYPE = np.dtype([("collider1", "<u4"), ("collider2", "<u4"), ("body1", "<i4"), ("body2", "<i4"), ("flags", "<u4"), ("owner", "<u4")])


class avn_shard_stats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("pairs_added", "pairs_removed", "pushes", "pops", "next_id", "n_free", "last_status_changes", "reserved")]


class Shard:
    """``avn_shard``: the replicated integer bookkeeping of a closed loop sharded by islands (host C++ behind the ABI; needs no device)."""

    def __init__(self, lib: Library, collider_entities, rank: int):
        self.lib = lib
        f = lib.fn
        f("shard_create").argtypes = [C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]
        f("shard_destroy").restype = None; f("shard_destroy").argtypes = [vp]
        f("shard_last_error").restype = C.c_char_p; f("shard_last_error").argtypes = [vp]
        f("shard_phase2").argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_size_t]
        f("shard_new_local_pairs").argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("shard_active").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("shard_phase3").argtypes = [vp, vp, C.c_size_t]
        f("shard_removed_local").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("shard_handles").argtypes = [vp, C.c_int, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        f("shard_stats_get").argtypes = [vp, vp]
        ent = np.ascontiguousarray(collider_entities, np.uint32)
        h = vp()
        st = f("shard_create")(len(ent), _ptr(ent), rank, C.byref(h))
        if st != 0:
            raise AvnError(st, "shard_create")
        self.handle = h

    def _chk(self, st, what):
        if st != 0:
            raise AvnError(st, what + ": " + (self.lib.fn("shard_last_error")(self.handle) or b"").decode())

    @staticmethod
    def _u32(p, n):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (n,)).copy() if n else np.zeros(0, np.uint32)

    def phase2(self, key_collider, key_min_x, pairs):
        kc = np.ascontiguousarray(key_collider, np.uint32); kx = np.ascontiguousarray(key_min_x, np.float64); pr = np.ascontiguousarray(pairs, SHARD_PAIR_DTYPE)
        self._chk(self.lib.fn("shard_phase2")(self.handle, _ptr(kc), _ptr(kx), len(kc), _ptr(pr) if len(pr) else None, len(pr)), "shard_phase2")
        a, b, c, d, n = vp(), vp(), vp(), vp(), C.c_size_t()
        self._chk(self.lib.fn("shard_new_local_pairs")(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(n)), "shard_new_local_pairs")
        return tuple(self._u32(x, n.value) for x in (a, b, c, d))

    def active(self):
        a, n = vp(), C.c_size_t()
        self._chk(self.lib.fn("shard_active")(self.handle, C.byref(a), C.byref(n)), "shard_active")
        return self._u32(a, n.value)

    def phase3(self, changes):
        ch = np.ascontiguousarray(changes, CHANGE_DTYPE)
        self._chk(self.lib.fn("shard_phase3")(self.handle, _ptr(ch) if len(ch) else None, len(ch)), "shard_phase3")
        a, n = vp(), C.c_size_t()
        self._chk(self.lib.fn("shard_removed_local")(self.handle, C.byref(a), C.byref(n)), "shard_removed_local")
        return self._u32(a, n.value)

    def handles(self, global_lists: bool = False):
        off = np.zeros(GRAPH_COLOR_COUNT + 1, np.uint32)
        a, n = vp(), C.c_size_t()
        self._chk(self.lib.fn("shard_handles")(self.handle, int(global_lists), _ptr(off), C.byref(a), C.byref(n)), "shard_handles")
        return off, self._u32(a, n.value)

    def stats(self):
        s = avn_shard_stats()
        self._chk(self.lib.fn("shard_stats_get")(self.handle, C.byref(s)), "shard_stats_get")
        return s

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fn("shard_destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConstraintGraph:
    """Host ``ConstraintGraph`` (constraint_graph.rs:163-296) behind the ABI."""

    def __init__(self, lib: Library, body_capacity: int = 16):
        self.lib = lib
        h = vp()
        st = lib.fn("constraint_graph_create")(body_capacity, C.byref(h))
        if st != 0:
            raise AvnError(st, "constraint_graph_create")
        self.handle = h

    def push(self, handle: int, body1: int, body2: int, is_static1: bool, is_static2: bool) -> int:
        return int(self.lib.fn("constraint_graph_push")(self.handle, handle, body1, body2, int(is_static1), int(is_static2)))

    def pop(self, handle: int):
        st = self.lib.fn("constraint_graph_pop")(self.handle, handle)
        if st != 0:
            raise AvnError(st, "constraint_graph_pop")

    def lists(self):
        offsets = np.zeros(GRAPH_COLOR_COUNT + 1, np.uint32)
        n = C.c_size_t()
        self.lib.fn("constraint_graph_lists")(self.handle, _ptr(offsets), None, 0, C.byref(n))
        handles = np.zeros(n.value, np.uint64)
        st = self.lib.fn("constraint_graph_lists")(self.handle, _ptr(offsets), _ptr(handles), handles.size, C.byref(n))
        if st != 0:
            raise AvnError(st, "constraint_graph_lists")
        return offsets, handles

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fn("constraint_graph_destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class avn_islands_result(C.Structure):
    _fields_ = [(n, t) for name in ("popped", "pushed", "pairs_slept", "pairs_woken", "bodies_slept", "bodies_woken", "pairs_removed") for n, t in ((name, vp), ("n_" + name, C.c_size_t))]


class IslandManager:
    """``avn_island_manager`` (persistent islands + sleeping bookkeeping, islands/mod.rs, islands/sleeping.rs) behind the ABI: a host structure."""
    NONE = 0xFFFFFFFF

    def __init__(self, lib: Library):
        lib._declare_islands()
        self.lib = lib
        self.handle = lib.fn("islands_create")()
        if not self.handle:
            raise AvnError(4, "islands_create")

    def _chk(self, st, what):
        if st != 0:
            raise AvnError(st, what)

    def body_add(self, body): self._chk(self.lib.fn("islands_body_add")(self.handle, body), "islands_body_add")
    def collider_add(self, collider, body): self._chk(self.lib.fn("islands_collider_add")(self.handle, collider, self.NONE if body is None else body), "islands_collider_add")
    def joint_add(self, joint, b1, b2): self._chk(self.lib.fn("islands_joint_add")(self.handle, joint, b1, b2), "islands_joint_add")
    def pair_add(self, cid, c1, c2): self._chk(self.lib.fn("islands_pair_add")(self.handle, cid, c1, c2), "islands_pair_add")
    def status_change(self, cid, flags, manifold_count=1): self._chk(self.lib.fn("islands_status_change")(self.handle, cid, flags, manifold_count), "islands_status_change")
    def flush_wake(self): self._chk(self.lib.fn("islands_flush_wake")(self.handle), "islands_flush_wake"); return self.last_result()
    def split_candidate(self): self._chk(self.lib.fn("islands_split_candidate")(self.handle), "islands_split_candidate")

    def split_candidate_adjacency(self, off, adj, labels=None):
        """split_island(candidate) with the contact neighbours as a CSR over bodies (the walk's edge order; see the header).  With `labels` the walk of an
        island that is still one piece runs on a worker thread: the arrays are kept alive here until split_join()."""
        o = np.ascontiguousarray(off, np.uint32); a = np.ascontiguousarray(adj, np.uint32)
        if len(a) == 0: a = np.zeros(1, np.uint32)
        l = None if labels is None else np.ascontiguousarray(labels, np.uint32)
        self._adj_keep = (o, a, l)
        self._chk(self.lib.fn("islands_split_candidate_adjacency")(self.handle, _ptr(o), _ptr(a), len(o) - 1, _ptr(l)), "islands_split_candidate_adjacency")

    def split_join(self): self._chk(self.lib.fn("islands_split_join")(self.handle), "islands_split_join")

    def wake_body(self, body): self._chk(self.lib.fn("islands_wake_body")(self.handle, body), "islands_wake_body"); return self.last_result()
    def sleep_body(self, body): self._chk(self.lib.fn("islands_sleep_body")(self.handle, body), "islands_sleep_body"); return self.last_result()
    def collider_remove(self, collider): self._chk(self.lib.fn("islands_collider_remove")(self.handle, collider), "islands_collider_remove"); return self.last_result()
    def body_remove(self, body): self._chk(self.lib.fn("islands_body_remove")(self.handle, body), "islands_body_remove"); return self.last_result()

    def joint_remove(self, joint): self._chk(self.lib.fn("islands_joint_remove")(self.handle, joint), "islands_joint_remove"); return self.last_result()

    def renumber_joints(self, new_index):
        m = np.ascontiguousarray(new_index, np.uint32)
        self._chk(self.lib.fn("islands_renumber_joints")(self.handle, _ptr(m), len(m)), "islands_renumber_joints")

    def renumber_bodies(self, new_index):
        m = np.ascontiguousarray(new_index, np.uint32)
        self._chk(self.lib.fn("islands_renumber_bodies")(self.handle, _ptr(m), len(m)), "islands_renumber_bodies")

    def sleeping_systems(self, sleep_timer, flags, time_to_sleep=0.5):
        t = np.ascontiguousarray(sleep_timer, np.float32); f = np.ascontiguousarray(flags, np.uint8)
        self._chk(self.lib.fn("islands_sleeping_systems")(self.handle, _ptr(t), _ptr(f), len(t), C.c_float(time_to_sleep)), "islands_sleeping_systems")
        return self.last_result()

    def last_result(self):
        r = avn_islands_result()
        self._chk(self.lib.fn("islands_last_result")(self.handle, C.byref(r)), "islands_last_result")
        out = {}
        for name in ("popped", "pushed", "pairs_slept", "pairs_woken", "bodies_slept", "bodies_woken", "pairs_removed"):
            n = getattr(r, "n_" + name)
            out[name] = np.ctypeslib.as_array(C.cast(getattr(r, name), C.POINTER(C.c_uint32)), (n,)).copy() if n else np.zeros(0, np.uint32)
        return out

    def stats(self):
        s = avn_islands_stats()
        self._chk(self.lib.fn("islands_stats_get")(self.handle, C.byref(s)), "islands_stats_get")
        return s

    def state(self, n_bodies):
        a = np.zeros(n_bodies, np.uint32); b = np.zeros(n_bodies, np.uint32); c = np.zeros(n_bodies, np.uint8); d = np.zeros(n_bodies, np.uint32)
        self._chk(self.lib.fn("islands_state")(self.handle, n_bodies, _ptr(a), _ptr(b), _ptr(c), _ptr(d)), "islands_state")
        return dict(island=a, next=b, sleeping=c, removed=d)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fn("islands_destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
