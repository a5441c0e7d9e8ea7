"""CPU: the oracle's contact solver and XPBD distance joint (SURVEY.md §8 rows a10-a23 -- the rows where the FLOPs are) against a THIRD
restatement written from the reference's text alone (tests/solver_second_opinion.py: numpy scalars, one rounding per operation, the
order of the Rust source).  System by system through the C ABI: the oracle's state before a system is handed to the restatement, and what the
restatement makes of it must equal the oracle's state after the system, bit for bit, f32 and f64, on `helpers.random_world` -- random,
physically inconsistent manifolds that reach every branch (speculative points, clamped friction, non-finite tangent masses, locked axes,
dominance, sleeping / disabled / static / kinematic bodies, an overflow colour).  The reference holds no vectors for these rows
(SURVEY.md §8c); two programs by one author agreeing is what the parity suite shows, a third written without looking at either is what can be
added here."""
import numpy as np
import pytest

import solver_second_opinion as S
from avian_amd import _ffi as F
from helpers import color_and_upload, oracle_lib, random_joints, random_world


def build(bits, seed, **kw):
    lib = oracle_lib()
    wd = random_world(seed=seed, **kw)
    w = F.World(lib, F.default_config(bits, substeps=4))
    offsets, perm = color_and_upload(w, lib, wd)
    return w, wd, offsets, perm


def same(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values differ, first at {tuple(np.argwhere(bad)[0])}: {a[tuple(np.argwhere(bad)[0])]!r} vs {b[tuple(np.argwhere(bad)[0])]!r}"


def constraints_to_arrays(A, cons, M):
    """The layout of avn_constraints_download for the restatement's constraints (absent constraints / points: zeros)."""
    T = A.T
    out = {"point_count": np.zeros(M, np.uint8), "relative_dominance": np.zeros(M, np.int16), "tangent1": np.zeros((M, 3), T), "anchor1": np.zeros((M, 4, 3), T),
           "initial_separation": np.zeros((M, 4), T), "normal_impulse": np.zeros((M, 4), T), "total_impulse": np.zeros((M, 4), T), "normal_effective_mass": np.zeros((M, 4), T),
           "tangent_impulse": np.zeros((M, 4, 2), T), "tangent_effective_inverse_mass": np.zeros((M, 4, 3), T), "softness_non_dynamic": np.zeros(M, np.uint8)}
    for m, c in enumerate(cons):
        if c is None: continue
        out["point_count"][m] = len(c["points"]); out["relative_dominance"][m] = c["relative_dominance"]; out["tangent1"][m] = c["tangent1"]
        out["softness_non_dynamic"][m] = c["non_dynamic_softness"]
        for k, p in enumerate(c["points"]):
            out["anchor1"][m, k] = p["anchor1"]; out["initial_separation"][m, k] = p["initial_separation"]; out["normal_impulse"][m, k] = p["impulse"]
            out["total_impulse"][m, k] = p["total_impulse"]; out["normal_effective_mass"][m, k] = p["effective_mass"]
            if p["tangent"]: out["tangent_impulse"][m, k] = p["tangent"]["impulse"]; out["tangent_effective_inverse_mass"][m, k] = p["tangent"]["k"]
    return out


def compare_constraints(A, cons, w, what, keys=None):
    got = constraints_to_arrays(A, cons, w.n_manifolds)
    ref = w.constraints_download()
    live = ref["point_count"] > 0
    for k in (keys or got):
        a, b = got[k], ref[k]
        if k != "point_count":   # rows without a constraint hold whatever the implementation leaves there
            a, b = a[live], b[live]
        same(a, b, f"{what}: constraints.{k}")


def compare_bodies(bodies, w, what):
    sb = w.solver_bodies_download()
    has = (sb["flags"] & S.NO_SOLVER_BODY) == 0
    same(np.array(bodies.lin)[has], sb["linear_velocity"][has], f"{what}: linear_velocity")
    same(np.array(bodies.ang)[has], sb["angular_velocity"][has], f"{what}: angular_velocity")
    same(np.array(bodies.dp)[has], sb["delta_position"][has], f"{what}: delta_position")
    same(np.array(bodies.dq)[has], sb["delta_rotation"][has], f"{what}: delta_rotation")


def generate_all(A, w, wd, perm, cfg):
    mf = {k: np.asarray(v)[perm] for k, v in wd["manifolds"].items()}
    T = A.T
    for k in ("normal", "anchor1", "anchor2", "penetration", "normal_speed", "tangent_velocity"):
        mf[k] = mf[k].astype(T)   # what the ABI hands to the library: the world's scalar type
    bodies = S.Bodies(A, w.solver_bodies_download())
    b = wd["bodies"]
    lin = np.asarray(b["linear_velocity"]).astype(T)
    flags = np.asarray(b.get("body_flags", np.zeros(len(lin), np.uint8)))
    fr, re = wd["friction"][perm].astype(T), wd["restitution"][perm].astype(T)
    wn, wt = wd["warm_n"][perm].astype(T), wd["warm_t"][perm].astype(T)
    cons = [S.generate(A, bodies, lin, mf, m, fr[m], re[m], wn, wt, bool(cfg.match_contacts), np.asarray(b["rb_type"]), flags) for m in range(len(perm))]
    return bodies, cons


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("seed", [3, 11])
def test_contact_constraints_system_by_system(bits, seed):
    A = S.Arith(bits)
    w, wd, offsets, perm = build(bits, seed, n_bodies=90, n_manifolds=260, hub_degree=26)
    cfg = w.cfg
    ts = S.time_scalars(A, cfg.dt_ns, cfg.substeps)
    soft = S.contact_softness(A, cfg, ts)
    order = S.in_solve_order(offsets)
    assert offsets[24] - offsets[23] > 0, "the scene must reach the overflow colour"
    w.run_system("PREPARE_SOLVER_BODIES")
    w.run_system("PREPARE_CONTACT_CONSTRAINTS")
    bodies, cons = generate_all(A, w, wd, perm, cfg)
    assert sum(c is not None for c in cons) > 150 and any(c is None for c in cons)
    compare_constraints(A, cons, w, "generate")
    w.run_system("PRE_PROCESS_VELOCITY_INCREMENTS")
    for sub in range(2):   # two substeps: the second starts from accumulated impulses and non-trivial delta rotations
        w.run_system("INTEGRATE_VELOCITIES")
        bodies = S.Bodies(A, w.solver_bodies_download())
        w.run_system("WARM_START")
        for m in order:
            if cons[m] is not None: S.warm_start(A, bodies, cons[m], A.T(cfg.warm_start_coefficient))
        compare_bodies(bodies, w, f"substep {sub}: warm start")
        w.run_system("SOLVE_CONTACTS_BIAS")
        for m in order:
            if cons[m] is not None: S.solve(A, bodies, cons[m], ts["h_adj"], True, A.T(cfg.max_overlap_solve_speed) * A.T(cfg.length_unit), soft)
        compare_bodies(bodies, w, f"substep {sub}: biased solve")
        compare_constraints(A, cons, w, f"substep {sub}: biased solve", ("normal_impulse", "total_impulse", "tangent_impulse"))
        w.run_system("INTEGRATE_POSITIONS")
        bodies = S.Bodies(A, w.solver_bodies_download())
        w.run_system("SOLVE_CONTACTS_RELAX")
        for m in order:
            if cons[m] is not None: S.solve(A, bodies, cons[m], ts["h_adj"], False, A.T(cfg.max_overlap_solve_speed) * A.T(cfg.length_unit), soft)
        compare_bodies(bodies, w, f"substep {sub}: relax")
        compare_constraints(A, cons, w, f"substep {sub}: relax", ("normal_impulse", "total_impulse", "tangent_impulse"))
    w.run_system("SOLVE_RESTITUTION")
    for m in order:
        if cons[m] is not None: S.restitution(A, bodies, cons[m], A.T(cfg.restitution_threshold) * A.T(cfg.length_unit), int(cfg.restitution_iterations))
    compare_bodies(bodies, w, "restitution")
    compare_constraints(A, cons, w, "restitution", ("normal_impulse", "total_impulse"))
    # store_contact_impulses (plugin.rs:722-755): impulse -> warm_start_normal_impulse, tangent impulse -> warm_start_tangent_impulse, total -> normal_impulse
    w.run_system("STORE_CONTACT_IMPULSES")
    imp = w.impulses_download()
    got = constraints_to_arrays(A, cons, w.n_manifolds)
    pts = np.arange(4)[None, :] < got["point_count"][:, None]   # the zip stops at the constraint's points: the manifold's other slots keep what they held
    same(got["normal_impulse"][pts], imp["warm_start_normal_impulse"][pts], "store: warm_start_normal_impulse")
    same(got["tangent_impulse"][pts], imp["warm_start_tangent_impulse"][pts], "store: warm_start_tangent_impulse")
    same(got["total_impulse"][pts], imp["normal_impulse"][pts], "store: normal_impulse")
    assert pts.sum() > 400


def test_softness_coefficients_against_the_reference_defaults():
    """ContactSoftnessCoefficients::default (plugin.rs:317-324) documents SoftnessParameters::new(10, 30) / (10, 60) at 1/60 s; and the step's own
    coefficients for dt = 1/60, 6 substeps follow hz = 1.5 * min(1 / (2 dt), 0.25 / h)."""
    A = S.Arith(64)
    d = S.softness_coefficients(A, 10.0, A.T(30.0), A.T(1.0 / 60.0))
    omega = 2 * np.pi * 30.0; h = 1.0 / 60.0
    a1 = 20.0 + omega * h; a2 = omega * h * a1
    assert abs(d["bias"] - omega / a1) < 1e-12 and abs(d["impulse_scale"] - 1 / (1 + a2)) < 1e-15 and abs(d["mass_scale"] - a2 / (1 + a2)) < 1e-15
    cfg = F.default_config(64, substeps=6)
    ts = S.time_scalars(A, cfg.dt_ns, cfg.substeps)
    s = S.contact_softness(A, cfg, ts)
    hz = 1.5 * min(1 / (2 * float(ts["dt"])), 0.25 / float(ts["h"]))
    assert abs(float(ts["h"]) * 6 - float(ts["dt"])) < 2e-9 and abs(hz - 45.0) < 1e-5
    assert s["non_dynamic"]["bias"] > s["dynamic"]["bias"] > 0


@pytest.mark.parametrize("bits", [32, 64])
def test_distance_joints_against_the_restatement(bits):
    """XPBD_SOLVE over random DistanceJoints (serial, in joint order; dominance and missing bodies -> DUMMY) for two substeps, with the oracle's
    sin / cos switched to the platform libm (the restatement calls the same libm through ctypes; the default build's polynomial is the
    oracle's own and is covered by tests/test_oracle_tolerance.py)."""
    A = S.Arith(bits)
    lib = oracle_lib()
    lib.dll.avo_use_libm_trig(1)
    try:
        w, wd, offsets, perm = build(bits, 5, n_bodies=60, n_manifolds=40, n_joints=80)
        cfg = w.cfg
        ts = S.time_scalars(A, cfg.dt_ns, cfg.substeps)
        T = A.T
        b = wd["bodies"]; J = wd["joints"]
        pos = [A.v(x) for x in np.asarray(b["position"]).astype(T)]
        rot = [tuple(T(c) for c in q) for q in np.asarray(b["rotation"]).astype(T)]
        com = [A.v(x) for x in np.asarray(b["center_of_mass"]).astype(T)]
        w.run_system("PREPARE_SOLVER_BODIES"); w.run_system("PREPARE_JOINTS"); w.run_system("PRE_PROCESS_VELOCITY_INCREMENTS")
        joints = [dict(body1=int(J["body1"][j]), body2=int(J["body2"][j]), local_anchor1=A.v(np.asarray(J["local_anchor1"][j]).astype(T)),
                       local_anchor2=A.v(np.asarray(J["local_anchor2"][j]).astype(T)), limit_min=T(J["limit_min"][j]), limit_max=T(J["limit_max"][j]),
                       compliance=T(J["compliance"][j])) for j in range(len(J["body1"]))]
        flags = np.asarray(b["body_flags"])
        disabled = lambda j: bool((flags[j["body1"]] | flags[j["body2"]]) & 2)
        data = [S.unprepared(A) if disabled(j) else S.distance_joint_prepare(A, pos, rot, com, j) for j in joints]
        moved = 0
        for sub in range(2):
            w.run_system("INTEGRATE_VELOCITIES"); w.run_system("INTEGRATE_POSITIONS")
            bodies = S.Bodies(A, w.solver_bodies_download())
            before = np.array(bodies.dp)
            w.run_system("XPBD_SOLVE")
            for j, d in zip(joints, data):   # (a joint with a disabled body is never prepared but IS solved, against a DUMMY: S.unprepared)
                S.distance_joint_solve(A, bodies, j, d, ts["h_adj"])
            compare_bodies(bodies, w, f"substep {sub}: distance joints")
            moved += int((np.array(bodies.dp) != before).any(axis=1).sum())
            w.run_system("XPBD_VELOCITY_PROJECTION"); w.run_system("JOINT_DAMPING")
        assert moved > 40
    finally:
        lib.dll.avo_use_libm_trig(0)


@pytest.mark.parametrize("bits", [32, 64])
def test_fixed_joints_against_the_restatement(bits):
    """The angular side of XPBD: FixedJoint = FixedAngleConstraintShared (rotation difference through three quaternion products, align_orientation,
    apply_angular_impulse) followed by PointConstraintShared, on random frames; serial in joint order, two substeps, libm sin / cos on both sides."""
    A = S.Arith(bits)
    lib = oracle_lib()
    lib.dll.avo_use_libm_trig(1)
    try:
        rng = np.random.default_rng(17)
        wd = random_world(seed=9, n_bodies=50, n_manifolds=30, n_joints=0)
        J = random_joints(rng, 50, 60)
        J["joint_type"][:] = F.JOINT_FIXED
        wd["joints_generic"] = J
        w = F.World(lib, F.default_config(bits, substeps=4))
        color_and_upload(w, lib, wd)
        cfg = w.cfg
        ts = S.time_scalars(A, cfg.dt_ns, cfg.substeps)
        T = A.T
        b = wd["bodies"]
        pos = [A.v(x) for x in np.asarray(b["position"]).astype(T)]
        rot = [tuple(T(c) for c in q) for q in np.asarray(b["rotation"]).astype(T)]
        com = [A.v(x) for x in np.asarray(b["center_of_mass"]).astype(T)]
        w.run_system("PREPARE_SOLVER_BODIES"); w.run_system("PREPARE_JOINTS"); w.run_system("PRE_PROCESS_VELOCITY_INCREMENTS")
        q4 = lambda a: tuple(T(c) for c in np.asarray(a).astype(T))
        joints = [dict(body1=int(J["body1"][j]), body2=int(J["body2"][j]), local_anchor1=A.v(np.asarray(J["local_anchor1"][j]).astype(T)),
                       local_anchor2=A.v(np.asarray(J["local_anchor2"][j]).astype(T)), local_basis1=q4(J["local_basis1"][j]), local_basis2=q4(J["local_basis2"][j]),
                       compliance=tuple(T(c) for c in np.asarray(J["compliance"][j]).astype(T))) for j in range(len(J["body1"]))]
        flags = np.asarray(b["body_flags"])
        disabled = lambda j: bool((flags[j["body1"]] | flags[j["body2"]]) & 2)
        data = [S.unprepared(A) if disabled(j) else S.fixed_joint_prepare(A, pos, rot, com, j) for j in joints]
        assert any(disabled(j) for j in joints), "the scene must hold a joint whose prepare is skipped"

        turned = 0
        for sub in range(2):
            w.run_system("INTEGRATE_VELOCITIES"); w.run_system("INTEGRATE_POSITIONS")
            bodies = S.Bodies(A, w.solver_bodies_download())
            before = np.array(bodies.dq)
            w.run_system("XPBD_SOLVE")
            for j, d in zip(joints, data):
                S.fixed_joint_solve(A, bodies, j, d, ts["h_adj"])
            compare_bodies(bodies, w, f"substep {sub}: fixed joints")
            turned += int((np.array(bodies.dq) != before).any(axis=1).sum())
            w.run_system("XPBD_VELOCITY_PROJECTION"); w.run_system("JOINT_DAMPING")
        jd = w.joints_download()
        same(np.array([d["total_lagrange"] for d in data]), jd["total_lagrange"], "total_position_lagrange")
        same(np.array([d["total_rotation_lagrange"] for d in data]), jd["total_rotation_lagrange"], "total_rotation_lagrange")
        assert turned > 40
    finally:
        lib.dll.avo_use_libm_trig(0)


@pytest.mark.parametrize("bits", [32, 64])
def test_apply_local_acceleration_against_the_restatement(bits):
    """apply_local_acceleration (forces/plugin.rs:207-241) in the oracle against the third restatement: the SolverBody state in front of INTEGRATE_VELOCITIES
    goes to the restatement (delta_rotation as the oracle's integrate_positions left it -- the trigonometry is not what is compared here), and the velocities after the
    system must be what it makes of them, bit for bit.  The bodies have nothing else that integrate_velocities would add (no gravity, no damping, isotropic inertia:
    v * 1 + 0), kinematic bodies and locked axes included."""
    A = S.Arith(bits)
    rng = np.random.default_rng(7 + bits)
    n = 48
    rot = rng.normal(size=(n, 4)); rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    rb = np.zeros(n, np.uint8); rb[:6] = F.RB_KINEMATIC
    iso = rng.uniform(0.5, 4.0, n)
    inv_i = np.zeros((n, 6)); inv_i[:, 0] = inv_i[:, 3] = inv_i[:, 5] = iso
    locked = np.where(rng.random(n) < 0.4, rng.integers(0, 64, n), 0).astype(np.uint8)
    bodies = dict(position=rng.uniform(-5, 5, (n, 3)), rotation=rot, linear_velocity=rng.normal(size=(n, 3)), angular_velocity=rng.normal(scale=2.0, size=(n, 3)),
                  inv_mass=rng.uniform(0.3, 2.0, n), inv_inertia_local=inv_i, rb_type=rb, locked_axes=locked)
    lin = np.where(rng.random((n, 1)) < 0.8, rng.normal(scale=5.0, size=(n, 3)), 0.0)
    ang = np.where(rng.random((n, 1)) < 0.8, rng.normal(scale=3.0, size=(n, 3)), 0.0)
    cfg = F.default_config(bits, substeps=3, gravity=(0.0, 0.0, 0.0))
    w = F.World(oracle_lib(), cfg)
    w.bodies_upload(**bodies)
    w.local_accelerations_upload(lin, ang)
    ts = S.time_scalars(A, int(cfg.dt_ns), 3)
    T = A.T
    rot_t = np.asarray(bodies["rotation"], T)   # the Rotation component as the world holds it
    lin_t, ang_t = np.asarray(lin, T), np.asarray(ang, T)
    w.run_system("PREPARE_SOLVER_BODIES"); w.run_system("PRE_PROCESS_VELOCITY_INCREMENTS")
    moved = 0
    for sub in range(3):
        before = w.solver_bodies_download()
        w.run_system("INTEGRATE_VELOCITIES")
        after = w.solver_bodies_download()
        exp_l, exp_a = np.array(before["linear_velocity"]), np.array(before["angular_velocity"])
        for i in range(n):
            l, a = S.apply_local_acceleration(A, A.v(before["linear_velocity"][i]), A.v(before["angular_velocity"][i]), tuple(T(x) for x in before["delta_rotation"][i]),
                                              tuple(rot_t[i]), A.v(lin_t[i]), A.v(ang_t[i]), int(locked[i]), ts["h"])
            exp_l[i], exp_a[i] = l, a
        same(after["linear_velocity"], exp_l, f"substep {sub}: linear_velocity"); same(after["angular_velocity"], exp_a, f"substep {sub}: angular_velocity")
        moved += int((np.asarray(after["linear_velocity"]) != np.asarray(before["linear_velocity"])).any(axis=1).sum())
        w.run_system("INTEGRATE_POSITIONS")
    assert moved > n and not np.array_equal(after["delta_rotation"], np.tile([0, 0, 0, 1], (n, 1))), "the later substeps must see a turned body"
