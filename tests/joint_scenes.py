"""Small two-body joint scenes + the invariants each XPBD joint type must enforce (used on the oracle and on the HIP path)."""
import numpy as np

from helpers import F


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def quat_rot(q, v):
    qv = np.array([v[0], v[1], v[2], 0.0])
    qc = np.array([-q[0], -q[1], -q[2], q[3]])
    return quat_mul(quat_mul(q, qv), qc)[:3]


def _bodies(kinematic_first=True):
    """body 0 kinematic at the origin (the "anchor"), body 1 a dynamic unit cube at x = 1 with some spin and velocity."""
    n = 2
    return dict(position=np.array([[0.0, 0, 0], [1.0, 0, 0]]), rotation=np.tile([0, 0, 0, 1.0], (n, 1)),
                linear_velocity=np.array([[0.0, 0, 0], [0.3, 1.0, -0.5]]), angular_velocity=np.array([[0.0, 0, 0], [1.5, -2.0, 1.0]]),
                inv_mass=np.array([1.0, 1.0]), inv_inertia_local=np.tile([6.0, 0, 0, 6, 0, 6], (n, 1)),
                rb_type=np.array([F.RB_KINEMATIC if kinematic_first else F.RB_DYNAMIC, F.RB_DYNAMIC], np.uint8))


def _joint(jtype, **kw):
    d = dict(joint_type=np.array([jtype], np.uint8), body1=np.array([0], np.int32), body2=np.array([1], np.int32),
             local_anchor1=np.array([[0.5, 0, 0.0]]), local_anchor2=np.array([[-0.5, 0, 0.0]]), compliance=np.zeros((1, 3)))
    d.update(kw)
    return d


JOINT_CASES = {
    "fixed": _joint(F.JOINT_FIXED),
    "revolute_free": _joint(F.JOINT_REVOLUTE, axis=np.array([[0.0, 0, 1]])),
    "revolute_limited": _joint(F.JOINT_REVOLUTE, axis=np.array([[0.0, 0, 1]]), limit_min=np.array([-0.3]), limit_max=np.array([0.4]),
                               limit_flags=np.array([F.JOINT_HAS_LIMIT1], np.uint8)),
    "spherical_free": _joint(F.JOINT_SPHERICAL, axis=np.array([[1.0, 0, 0]])),
    "spherical_limited": _joint(F.JOINT_SPHERICAL, axis=np.array([[1.0, 0, 0]]), limit_min=np.array([-0.5]), limit_max=np.array([0.5]),
                                limit2_min=np.array([-0.4]), limit2_max=np.array([0.4]),
                                limit_flags=np.array([F.JOINT_HAS_LIMIT1 | F.JOINT_HAS_LIMIT2], np.uint8)),
    "prismatic_limited": _joint(F.JOINT_PRISMATIC, axis=np.array([[0.0, 1, 0]]), limit_min=np.array([-0.5]), limit_max=np.array([0.75]),
                                limit_flags=np.array([F.JOINT_HAS_LIMIT1], np.uint8)),
    "distance": _joint(F.JOINT_DISTANCE, limit_min=np.array([0.2]), limit_max=np.array([0.6])),
}


def run_joint_case(lib, case, bits=32, steps=120):
    w = F.World(lib, F.default_config(bits, substeps=8, gravity=[0.0, -9.81, 0.0]))
    w.bodies_upload(**_bodies())
    w.joints_upload(**JOINT_CASES[case])
    for _ in range(steps):
        w.step()
    w.synchronize()
    out = w.bodies_download()
    out.update({"j." + k: v for k, v in w.joints_download().items()})
    w.close()
    return {k: np.asarray(v, np.float64) for k, v in out.items()}


def check_joint_case(case, out):
    j = JOINT_CASES[case]
    p0, p1 = out["position"]; q0, q1 = out["rotation"]
    a0 = p0 + quat_rot(q0, j["local_anchor1"][0]); a1 = p1 + quat_rot(q1, j["local_anchor2"][0])
    sep = a1 - a0
    assert np.isfinite(out["position"]).all() and np.isfinite(out["rotation"]).all()
    assert abs(np.linalg.norm(q1) - 1.0) < 1e-3
    assert np.abs(out["j.force"]).max() > 0.0, "the joint carried the body's weight: force must be reported"
    tol = 2e-2
    if case == "distance":
        d = np.linalg.norm(sep)
        assert j["limit_min"][0] - tol <= d <= j["limit_max"][0] + tol, d
        return
    if case.startswith("prismatic"):
        axis = quat_rot(q0, j["axis"][0])
        along = float(sep @ axis); off = sep - along * axis
        assert np.linalg.norm(off) < tol, off
        assert j["limit_min"][0] - tol <= along <= j["limit_max"][0] + tol, along
        assert abs(abs(float(q1 @ q0)) - 1.0) < 1e-2, "prismatic joints lock the relative rotation"
        assert along < -0.3, "gravity pulls the slider to its lower limit"
        return
    assert np.linalg.norm(sep) < tol, f"{case}: anchors drifted apart by {np.linalg.norm(sep)}"
    if case == "fixed":
        assert abs(abs(float(q1 @ q0)) - 1.0) < 1e-2
        assert np.abs(out["j.torque"]).max() > 0.0
    if case.startswith("revolute"):
        ax0 = quat_rot(q0, j["axis"][0]); ax1 = quat_rot(q1, j["axis"][0])
        assert np.linalg.norm(np.cross(ax0, ax1)) < 3e-2, "hinge axes must stay aligned"
        if "limit_min" in j:
            b0 = quat_rot(q0, [1.0, 0, 0]); b1 = quat_rot(q1, [1.0, 0, 0])
            ang = np.arctan2(np.cross(b0, b1) @ ax0, b0 @ b1)
            assert j["limit_min"][0] - 0.05 <= ang <= j["limit_max"][0] + 0.05, ang
    if case == "spherical_limited":
        # swing: angle between the swing axes (any_orthonormal_vector of the twist axis) stays within the cone limit
        tw = j["axis"][0]
        s = np.copysign(1.0, tw[2]); a = -1.0 / (s + tw[2]); b = tw[0] * tw[1] * a
        swing = np.array([b, s + tw[1] * tw[1] * a, -tw[1]])
        s0 = quat_rot(q0, swing); s1 = quat_rot(q1, swing)
        ang = np.arccos(np.clip(s0 @ s1, -1, 1))
        assert ang <= j["limit_max"][0] + 0.08, ang
