"""Shared checkers for the committed golden fixtures (tests/golden/): run a library (oracle or HIP product) through
the C ABI and compare with (a) the reference's own known-answer tests and (b) the frozen oracle vectors."""
from __future__ import annotations

import json
import math
import os

import numpy as np

from helpers import F, color_and_upload, random_joints, random_world

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kats():
    return json.load(open(os.path.join(GOLDEN, "reference_kats.json")))["cases"]


def run_kat(lib, case, bits=32, **cfgkw):
    """Drive one reference KAT through `lib`: bodies only (no contacts), `steps` x avn_step; returns bodies_download."""
    c = case["config"]
    cfg = F.default_config(bits, substeps=c["substeps"], dt=c["dt"], gravity=c["gravity"], **cfgkw)
    w = F.World(lib, cfg)
    bs = case["bodies"]
    n = len(bs)
    arr = lambda k, d: np.array([b.get(k, d) for b in bs], dtype=np.float64)
    kw = dict(position=arr("position", None), rotation=arr("rotation", None), linear_velocity=arr("linear_velocity", None),
              angular_velocity=arr("angular_velocity", None), inv_mass=arr("inv_mass", None),
              inv_inertia_local=arr("inv_inertia_local", None), rb_type=np.zeros(n, np.uint8),
              accel_linear=arr("accel_linear", [0, 0, 0]), accel_angular=arr("accel_angular", [0, 0, 0]))
    imp = arr("impulse_linear", [0, 0, 0])   # Forces::apply_linear_impulse in FixedUpdate: LinearVelocity += J / m BEFORE every physics step
    impa = arr("impulse_angular", [0, 0, 0])  # apply_angular_impulse: AngularVelocity += I_world^-1 J (isotropic inertia in these tests: I^-1 = inv_inertia_local[0])
    # Forces::apply_local_force / _torque / _linear_acceleration / _angular_acceleration accumulate into AccumulatedLocalAcceleration in the body's LOCAL frame
    # (forces/query_data.rs:344-349,374-379,526-530,554-558); apply_local_acceleration turns it into the world every substep (forces/plugin.rs:207-241)
    lacc_l, lacc_a = arr("local_accel_linear", [0, 0, 0]), arr("local_accel_angular", [0, 0, 0])
    local = np.any(lacc_l) or np.any(lacc_a)
    kick = np.any(imp) or np.any(impa)
    reupload = kick or np.any(kw["accel_linear"]) or np.any(kw["accel_angular"])
    for s in range(case["steps"]):
        # the reference's ForcePlugin re-applies the user's persistent force / acceleration every step and clear_velocity_increments
        # (integrator/mod.rs:316-328) wipes it at the end of the step: re-upload like the ECS would
        if s > 0 and reupload:
            out = w.bodies_download()
            kw.update(position=out["position"], rotation=out["rotation"], linear_velocity=out["linear_velocity"], angular_velocity=out["angular_velocity"])
        if kick:
            kw["linear_velocity"] = np.asarray(kw["linear_velocity"], np.float64) + imp * kw["inv_mass"][:, None]
            kw["angular_velocity"] = np.asarray(kw["angular_velocity"], np.float64) + impa * kw["inv_inertia_local"][:, :1]
        if s == 0 or reupload:
            w.bodies_upload(**kw)
        if local:   # (cleared after every step and accumulated again before the next: clear_accumulated_local_acceleration, forces/plugin.rs:243-251)
            w.local_accelerations_upload(lacc_l, lacc_a)
        w.step()
    w.synchronize()
    out = w.bodies_download()
    w.close()
    return out


def check_kat(out, case):
    for e in case["expect"]:
        b, field, eps = e["body"], e["field"], e["epsilon"]
        if field == "rotation":  # assert_relative_eq!(rotation.0, Quaternion::from_rotation_z(a), epsilon): component-wise
            a = e["value_rotation_z"]
            want = np.array([0.0, 0.0, math.sin(a / 2), math.cos(a / 2)])
            got = out["rotation"][b].astype(np.float64)
            assert np.all(np.abs(got - want) <= eps), f"{case['name']}: rotation {got} vs {want} (eps {eps})"
        elif field == "rotation_angle_between_z":  # glam Quat::angle_between = 2 acos(|dot|)
            a = e["value_rotation_z"]
            want = np.array([0.0, 0.0, math.sin(a / 2), math.cos(a / 2)])
            got = out["rotation"][b].astype(np.float64)
            diff = 2.0 * math.acos(min(1.0, abs(float(got @ want))))
            assert diff < eps, f"{case['name']}: angle difference {diff} is not less than {eps}"
        elif field == "rotation_angle_between":  # against an arbitrary quaternion
            want = np.array(e["value_quat"], np.float64)
            got = out["rotation"][b].astype(np.float64)
            diff = 2.0 * math.acos(min(1.0, abs(float(got @ want))))
            assert diff < eps, f"{case['name']}: angle difference {diff} is not less than {eps}"
        elif field.endswith(".y"):
            got = float(out[field[:-2]][b][1])
            assert abs(got - e["value"]) <= eps, f"{case['name']}: {field} {got} vs {e['value']} (eps {eps})"
        else:
            got = out[field][b].astype(np.float64)
            want = np.array(e["value"])
            assert np.all(np.abs(got - want) <= eps), f"{case['name']}: {field} {got} vs {want} (eps {eps})"


def solver_vectors(lib, bits, graph_lib, **cfgkw):
    """The exact procedure of golden/make_oracle_vectors.py:solver_case, against `lib`."""
    wd = random_world(seed=2026, n_bodies=96, n_manifolds=260, n_joints=40, hub_degree=26)
    w = F.World(lib, F.default_config(bits, substeps=3, **cfgkw))
    color_and_upload(w, graph_lib, wd)
    for _ in range(3):
        w.step()
    w.synchronize()
    out = {}
    for name, d in (("bodies", w.bodies_download()), ("impulses", w.impulses_download()), ("joints", w.joints_download())):
        for k, v in d.items():
            out[f"{name}.{k}"] = v
    w.close()
    return out


def joints_vectors(lib, bits, graph_lib, **cfgkw):
    """The exact procedure of golden/make_oracle_vectors.py:joints_case, against `lib`."""
    wd = random_world(seed=77, n_bodies=80, n_manifolds=150, n_joints=0)
    wd["joints_generic"] = random_joints(np.random.default_rng(77), 80, 90)
    w = F.World(lib, F.default_config(bits, substeps=3, **cfgkw))
    color_and_upload(w, graph_lib, wd)
    for _ in range(2):
        w.step()
    w.synchronize()
    out = {}
    for name, d in (("gj.bodies", w.bodies_download()), ("gj.joints", w.joints_download())):
        for k, v in d.items():
            out[f"{name}.{k}"] = v
    w.close()
    return out


def broadphase_vectors(lib, bits):
    from avian_amd import scenes
    sc = scenes.box_stack(6, 4, 5)
    w = F.World(lib, F.default_config(bits, substeps=2))
    w.bodies_upload(**sc.body_kwargs())
    w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB")
    w.run_system("COLLECT_COLLISION_PAIRS")
    mn, mx, ents = w.aabbs_download()
    p = w.pairs_get()
    out = {"bp.aabb_min": mn, "bp.aabb_max": mx, "bp.interval_entities": ents,
           "bp.pairs": np.stack([p["collider1"], p["collider2"], p["body1"].astype(np.uint32), p["body2"].astype(np.uint32), p["flags"]], axis=1)}
    w.close()
    return out


def narrow_vectors(lib, bits):
    """contact_query::contact_manifolds on 300 seeded shape pairs + a closed-loop pile (device narrow phase) after 12 steps."""
    from narrow_scenes import random_pairs
    from pipeline_scenes import dropped_boxes
    w = F.World(lib, F.default_config(bits))
    q = w.contact_manifolds(**random_pairs(5, 300))
    out = {f"np.query.{k}": v for k, v in q.items()}
    w.close()
    bodies, colliders = dropped_boxes(seed=13, n=36)
    w = F.World(lib, F.default_config(bits, substeps=3))
    w.bodies_upload(**bodies); w.colliders_upload(**colliders)
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.collider_materials_upload(friction=0.7, restitution=0.1)
    w.pipeline_enable()
    for _ in range(12):
        w.step()
    w.synchronize()
    off, handles = w.pipeline_handles()
    out["np.loop.color_offsets"] = off
    out["np.loop.handles"] = handles
    for k, v in w.contacts_download(np.sort(handles)).items():
        out[f"np.loop.contacts.{k}"] = v
    for k, v in w.bodies_download().items():
        out[f"np.loop.bodies.{k}"] = v
    w.close()
    return out


def check_vectors(got: dict, bits: int):
    want = np.load(os.path.join(GOLDEN, f"oracle_vectors_f{bits}.npz"))
    keys = [k for k in want.files if k in got]
    assert keys, "no overlapping arrays"
    for k in keys:
        a, b = got[k], want[k]
        assert a.shape == b.shape and a.dtype == b.dtype, f"{k}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}"
        same = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)
        assert same.all(), f"{k}: {int((~same).sum())} of {same.size} values differ from the committed golden vector (bit-exact bar)"
    return len(keys)


# ---- two more known-answer tests of the reference that touch the path (beyond the stepping KATs in reference_kats.json) ----

def check_coefficient_combine(lib, bits=32):
    """src/dynamics/rigid_body/physics_material.rs:398-446 `coefficient_combine_works`: Restitution(0.3, Average) combined with
    Restitution(0.7, rule) for every CoefficientCombine rule -- the rule of the pair is the LARGER of the two (:205-214, :372-380), the
    expected coefficients are the reference test's (epsilon 1e-4 where it uses assert_relative_eq!, exact for Min / Max).
    Here the combination happens where the reference does it on the path: in NarrowPhase::update_contacts, for the friction and the
    restitution of a touching pair (system_param.rs:596-640).  Two overlapping unit cuboids, one contact pair, one narrow-phase run."""
    expected = {F.COMBINE_AVERAGE: (0.5, 1e-4), F.COMBINE_GEOMETRIC_MEAN: (0.458_257_56, 1e-4), F.COMBINE_MIN: (0.3, 1e-6),
                F.COMBINE_MULTIPLY: (0.21, 1e-4), F.COMBINE_MAX: (0.7, 1e-6)}
    for rule, (want, eps) in expected.items():
        w = F.World(lib, F.default_config(bits, substeps=1))
        n = 2
        rot = np.zeros((n, 4)); rot[:, 3] = 1.0
        w.bodies_upload(position=np.array([[0.0, 0.0, 0.0], [0.0, 0.9, 0.0]]), rotation=rot, linear_velocity=np.zeros((n, 3)),
                        angular_velocity=np.zeros((n, 3)), inv_mass=np.ones(n), inv_inertia_local=np.tile([6.0, 0, 0, 6.0, 0, 6.0], (n, 1)),
                        rb_type=np.zeros(n, np.uint8))
        w.colliders_upload(entity_index=np.array([10, 11], np.uint32), body=np.array([0, 1], np.int32), shape=np.zeros(n, np.uint8),
                           half_extents=np.full((n, 3), 0.5))
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        both = np.array([F.COMBINE_AVERAGE, rule], np.uint8)
        w.collider_materials_upload(friction=np.array([0.3, 0.7]), restitution=np.array([0.3, 0.7]), friction_combine=both, restitution_combine=both)
        w.run_system("UPDATE_AABB")
        w.contact_pairs_add(np.array([0], np.uint32), np.array([10], np.uint32), np.array([11], np.uint32),
                            np.array([F.PAIR_GENERATE_CONSTRAINTS], np.uint32))
        w.active_pairs_set(np.array([0], np.uint32))
        w.run_system("NARROW_PHASE")
        row = w.contacts_download(np.array([0], np.uint32))
        assert row["flags"][0] & F.CP_TOUCHING and row["point_count"][0] > 0, "the cuboids overlap by 0.1: the pair must touch"
        for field in ("restitution", "friction"):
            got = float(row[field][0])
            assert abs(got - want) <= eps, f"combine rule {rule}: {field} {got} != {want} (reference test value)"
        w.close()


def check_solver_body_membership(lib, bits=32):
    """src/dynamics/solver/solver_body/plugin.rs:318-353 `add_remove_solver_bodies`: dynamic and kinematic bodies have a SolverBody,
    static ones do not; RigidBodyDisabled removes it, removing the marker brings it back; so does Sleeping (:38-96)."""
    n = 4
    rot = np.zeros((n, 4)); rot[:, 3] = 1.0
    base = dict(position=np.zeros((n, 3)), rotation=rot, linear_velocity=np.zeros((n, 3)), angular_velocity=np.zeros((n, 3)), inv_mass=np.ones(n),
                inv_inertia_local=np.tile([1.0, 0, 0, 1.0, 0, 1.0], (n, 1)),
                rb_type=np.array([F.RB_DYNAMIC, F.RB_KINEMATIC, F.RB_STATIC, F.RB_DYNAMIC], np.uint8))

    def membership(flags):
        w = F.World(lib, F.default_config(bits, substeps=1))
        w.bodies_upload(**base, body_flags=np.asarray(flags, np.uint8))
        w.run_system("PREPARE_SOLVER_BODIES")
        has = (w.solver_bodies_download()["flags"] >> 31) == 0
        w.close()
        return has.tolist()
    assert membership([0, 0, 0, 0]) == [True, True, False, True]
    assert membership([F.BODY_DISABLED, 0, 0, 0]) == [False, True, False, True]          # insert(RigidBodyDisabled)
    assert membership([0, 0, 0, 0]) == [True, True, False, True]                           # remove::<RigidBodyDisabled>()
    assert membership([0, F.BODY_SLEEPING, 0, F.BODY_DISABLED]) == [True, False, False, False]
