"""Shared test helpers: seeded random worlds that exercise every branch of the hot path, and drivers that feed
the SAME arrays to the CPU oracle and to the HIP product through the common C ABI."""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from avian_amd import _ffi as F  # noqa: E402

ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.environ.get("AVO_LIB_PATH") or os.path.join(ORACLE_DIR, "liboracle.so")   # (AVO_LIB_PATH: a sanitizer build of the checker, tools/sanitize_cpu.sh)
_oracle = None


def oracle_lib() -> F.Library:
    """The CPU oracle (test infrastructure).  Built on demand with its own Makefile (g++, seconds)."""
    global _oracle
    if _oracle is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_capi.cpp", "avo_world.hpp", "avo_islands.hpp", "avo_math.hpp", "avo_narrow.hpp", "avo_parallel.hpp")]
        stale = not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
        if stale:
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        _oracle = F.Library(ORACLE_SO, "avo_")
    return _oracle


def hip_lib() -> F.Library:
    import avian_amd
    return avian_amd.load_library()


_measure = None


def hip_measure_lib() -> F.Library:
    """The `make measure` build of the product (-DAVN_MEASURE, avian_amd/csrc/measure/): the ONLY build that reads AVN_* environment switches -- the
    release library ignores the environment (avn_device.h: avn_env).  Tests that force an older or a rarely taken code path through a switch create THAT
    world on this library; everything else runs on the release build."""
    global _measure
    if _measure is None:
        path = os.path.join(REPO, "avian_amd", "csrc", "measure", "libavian_mi355x.so")
        assert os.path.exists(path), f"{path} is missing: `make -C avian_amd/csrc measure` (build() does it)"
        import torch  # noqa: F401  (one HIP runtime per process, as in avian_amd.load_library)
        _measure = F.Library(path, "avn_")
    return _measure


MEASURE_LIB_PATH = os.path.join(REPO, "avian_amd", "csrc", "measure", "libavian_mi355x.so")


def random_unit_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def random_spd_inverse_inertia(rng, n, isotropic_fraction=0.3):
    """Random SPD local inverse inertia tensors (m00,m01,m02,m11,m12,m22); a fraction isotropic (not gyroscopic)."""
    out = np.zeros((n, 6))
    for i in range(n):
        if rng.random() < isotropic_fraction:
            s = rng.uniform(0.5, 8.0)
            out[i] = [s, 0, 0, s, 0, s]
        else:
            a = rng.normal(size=(3, 3))
            m = a @ a.T + np.eye(3) * rng.uniform(0.3, 2.0)
            out[i] = [m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]]
    return out


def random_world(seed=0, n_bodies=200, n_manifolds=600, n_joints=0, n_static=5, n_kinematic=5, hub_degree=0,
                 with_odd_features=True):
    """Random bodies + random manifolds (+ random distance joints).  Geometry is NOT physically consistent on
    purpose: the solver arithmetic does not care, and random anchors/normals hit far more code paths than a
    resting stack (speculative contacts, clamped friction, locked axes, dominance, zero-mass axes...).
    ``hub_degree`` > 0 makes body `n_static + n_kinematic` touch that many other dynamic bodies so the greedy
    colouring overflows into colour 23 (needs > 20)."""
    rng = np.random.default_rng(seed)
    n = n_bodies
    rb = np.zeros(n, np.uint8)
    rb[:n_static] = F.RB_STATIC
    rb[n_static:n_static + n_kinematic] = F.RB_KINEMATIC
    pos = rng.uniform(-10, 10, size=(n, 3))
    rot = random_unit_quats(rng, n)
    lv = rng.normal(scale=2.0, size=(n, 3))
    av = rng.normal(scale=1.5, size=(n, 3))
    inv_mass = rng.uniform(0.2, 3.0, size=n)
    inv_i = random_spd_inverse_inertia(rng, n)
    inv_mass[rb == F.RB_STATIC] = 0.0
    inv_i[rb == F.RB_STATIC] = 0.0
    lv[rb == F.RB_STATIC] = 0.0
    av[rb == F.RB_STATIC] = 0.0
    bodies = dict(position=pos, rotation=rot, linear_velocity=lv, angular_velocity=av, inv_mass=inv_mass,
                  inv_inertia_local=inv_i, rb_type=rb, center_of_mass=rng.normal(scale=0.1, size=(n, 3)))
    if with_odd_features:
        bodies["linear_damping"] = np.where(rng.random(n) < 0.3, rng.uniform(0, 2, n), 0.0)
        bodies["angular_damping"] = np.where(rng.random(n) < 0.3, rng.uniform(0, 2, n), 0.0)
        bodies["gravity_scale"] = np.where(rng.random(n) < 0.2, rng.uniform(-1, 2, n), 1.0)
        bodies["accel_linear"] = np.where(rng.random((n, 1)) < 0.2, rng.normal(size=(n, 3)), 0.0)
        bodies["accel_angular"] = np.where(rng.random((n, 1)) < 0.2, rng.normal(size=(n, 3)), 0.0)
        bodies["max_linear_speed"] = np.where(rng.random(n) < 0.15, rng.uniform(0.5, 3, n), -1.0)
        bodies["max_angular_speed"] = np.where(rng.random(n) < 0.15, rng.uniform(0.5, 3, n), -1.0)
        bodies["locked_axes"] = np.where(rng.random(n) < 0.2, rng.integers(0, 64, n), 0).astype(np.uint8)
        bodies["dominance"] = np.where(rng.random(n) < 0.15, rng.integers(-3, 4, n), 0).astype(np.int8)
        fl = np.zeros(n, np.uint8)
        fl[rng.random(n) < 0.04] |= F.BODY_SLEEPING
        fl[rng.random(n) < 0.03] |= F.BODY_DISABLED
        fl[rng.random(n) < 0.03] |= F.BODY_CUSTOM_VEL
        fl[rng.random(n) < 0.03] |= F.BODY_CUSTOM_POS
        bodies["body_flags"] = fl
    # manifolds: random distinct body pairs (+ a high-degree hub to force the overflow colour)
    b1 = rng.integers(0, n, n_manifolds)
    b2 = (b1 + 1 + rng.integers(0, n - 1, n_manifolds)) % n
    if hub_degree:
        hub = n_static + n_kinematic
        others = rng.choice(np.arange(hub + 1, n), size=hub_degree, replace=False)
        swap = rng.random(hub_degree) < 0.5
        hb1 = np.where(swap, others, hub); hb2 = np.where(swap, hub, others)
        b1 = np.concatenate([b1, hb1]); b2 = np.concatenate([b2, hb2])
    m = len(b1)
    normal = rng.normal(size=(m, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    mf = dict(body1=b1.astype(np.int32), body2=b2.astype(np.int32), normal=normal,
              point_count=rng.integers(1, 5, m).astype(np.uint8),
              anchor1=rng.normal(scale=0.7, size=(m, 4, 3)), anchor2=rng.normal(scale=0.7, size=(m, 4, 3)),
              penetration=rng.normal(scale=0.05, size=(m, 4)), normal_speed=rng.normal(scale=2.0, size=(m, 4)))
    friction = np.where(rng.random(m) < 0.15, 0.0, rng.uniform(0.05, 1.2, m))
    restitution = np.where(rng.random(m) < 0.6, 0.0, rng.uniform(0.1, 0.9, m))
    mf["tangent_velocity"] = np.where(rng.random((m, 1)) < 0.1, rng.normal(size=(m, 3)), 0.0)
    mflags = np.full(m, F.MANIFOLD_GENERATES_CONSTRAINTS, np.uint8)
    mflags[rng.random(m) < 0.03] = 0
    mf["manifold_flags"] = mflags
    if with_odd_features:
        mf["point_count"][rng.random(m) < 0.02] = 0
    warm_n = np.abs(rng.normal(scale=0.3, size=(m, 4)))
    warm_t = rng.normal(scale=0.1, size=(m, 4, 2))
    joints = None
    if n_joints:
        jb1 = rng.integers(0, n, n_joints)
        jb2 = (jb1 + 1 + rng.integers(0, n - 1, n_joints)) % n
        lo = rng.uniform(0.0, 2.0, n_joints)
        joints = dict(body1=jb1.astype(np.int32), body2=jb2.astype(np.int32),
                      local_anchor1=rng.normal(scale=0.3, size=(n_joints, 3)),
                      local_anchor2=rng.normal(scale=0.3, size=(n_joints, 3)),
                      limit_min=lo, limit_max=lo + rng.uniform(0.0, 1.0, n_joints),
                      compliance=np.where(rng.random(n_joints) < 0.5, 0.0, rng.uniform(0, 1e-3, n_joints)),
                      damping_linear=rng.uniform(0, 3, n_joints), damping_angular=rng.uniform(0, 3, n_joints))
    return dict(bodies=bodies, manifolds=mf, friction=friction, restitution=restitution, warm_n=warm_n, warm_t=warm_t,
                joints=joints)


def random_joints(rng, n_bodies, n_joints, with_damping=True):
    """Random joints of ALL five XPBD types (avn_joints arrays): random frames, axes, limits (some absent), compliances.
    Like random_world the configuration is not physically consistent on purpose — every branch of every solve() runs."""
    J = n_joints
    jb1 = rng.integers(0, n_bodies, J)
    jb2 = (jb1 + 1 + rng.integers(0, n_bodies - 1, J)) % n_bodies
    jt = rng.integers(0, 5, J).astype(np.uint8)
    axis = rng.normal(size=(J, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    axis[rng.random(J) < 0.2] = [0.0, 0.0, 1.0]; axis[rng.random(J) < 0.1] = [0.0, 1.0, 0.0]; axis[rng.random(J) < 0.1] = [1.0, 0.0, 0.0]
    lo = np.where(jt == F.JOINT_DISTANCE, rng.uniform(0.0, 2.0, J), np.where(jt == F.JOINT_PRISMATIC, rng.uniform(-1.0, 0.2, J), rng.uniform(-2.0, 0.5, J)))
    hi = lo + np.where(jt == F.JOINT_DISTANCE, rng.uniform(0.0, 1.0, J), rng.uniform(0.0, 1.5, J))
    lo2 = rng.uniform(-2.5, 0.2, J); hi2 = lo2 + rng.uniform(0.0, 2.0, J)
    flags = rng.integers(0, 4, J).astype(np.uint8)
    comp = np.where(rng.random((J, 3)) < 0.5, 0.0, rng.uniform(0, 1e-3, (J, 3)))
    out = dict(joint_type=jt, body1=jb1.astype(np.int32), body2=jb2.astype(np.int32),
               local_anchor1=rng.normal(scale=0.3, size=(J, 3)), local_anchor2=rng.normal(scale=0.3, size=(J, 3)),
               local_basis1=random_unit_quats(rng, J), local_basis2=random_unit_quats(rng, J), axis=axis,
               limit_min=lo, limit_max=hi, limit2_min=lo2, limit2_max=hi2, limit_flags=flags, compliance=comp,
               collision_disabled=(rng.random(J) < 0.2).astype(np.uint8))
    if with_damping:
        out["damping_linear"] = rng.uniform(0, 3, J); out["damping_angular"] = rng.uniform(0, 3, J)
    return out


def color_and_upload(world: F.World, lib_for_graph: F.Library, wd: dict):
    """Colour the manifolds (persistent greedy, in manifold order) and upload everything to `world`.
    Returns the colour-major permutation so that results can be mapped back."""
    from avian_amd import scenes
    world.bodies_upload(**wd["bodies"])
    mf = wd["manifolds"]
    offsets, perm = scenes.color_manifolds(lib_for_graph, mf, np.asarray(wd["bodies"]["rb_type"]))
    pm = scenes.permute_manifolds(mf, perm)
    scenes.upload_manifolds(world, pm, offsets, wd["friction"][perm], wd["restitution"][perm],
                            warm_n=wd["warm_n"][perm], warm_t=wd["warm_t"][perm])
    if wd.get("joints_generic"):
        world.joints_upload(**wd["joints_generic"])
    elif wd.get("joints"):
        world.distance_joints_upload(**wd["joints"])
    return offsets, perm


def assert_same(a: np.ndarray, b: np.ndarray, what: str, tol: float = 0.0):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if tol == 0.0:
        bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    else:
        bad = ~(np.isclose(a, b, rtol=tol, atol=tol) | (np.isnan(a) & np.isnan(b)))
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} differ; first at {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r}")


def compare_dicts(da: dict, db: dict, what: str, tol: float = 0.0, skip=()):
    for k in da:
        if k in skip:
            continue
        assert_same(da[k], db[k], f"{what}.{k}", tol)
