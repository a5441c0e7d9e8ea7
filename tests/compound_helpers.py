"""Child colliders (include/avian_mi355x.h "child colliders"): compound bodies whose colliders are CHILD entities with a ColliderTransform
(reference collision/collider/collider_transform: update_child_collider_position, plugin.rs:62-91)."""
from __future__ import annotations

import numpy as np

from helpers import F
from pipeline_scenes import random_unit_quats


def compound_scene(seed=0, n_bodies=24, balls=True):
    """A static slab + n compound bodies dropped over it.  Every body has a collider on its own entity (a small core cuboid) and 1-3 CHILD colliders (cuboids / balls) at
    random offsets and orientations; masses / inertias are those of a solid box around the compound (the host's job in Avian: ComputedMass ...), the centre of mass
    is deliberately off-centre for some bodies."""
    rng = np.random.default_rng(seed)
    m = n_bodies + 1
    pos = np.zeros((m, 3)); pos[0] = [0, -0.5, 0]
    side = int(np.ceil(n_bodies ** (1 / 3)))
    k = 0
    for i in range(side):
        for j in range(side):
            for l in range(side):
                if k < n_bodies:
                    pos[1 + k] = [2.6 * i + rng.uniform(-0.1, 0.1), 1.6 + 2.6 * j, 2.6 * l + rng.uniform(-0.1, 0.1)]; k += 1
    rot = np.tile([0.0, 0, 0, 1], (m, 1)); rot[1:] = random_unit_quats(rng, n_bodies)
    lv = np.zeros((m, 3)); av = np.zeros((m, 3))
    lv[1:] = rng.normal(scale=0.5, size=(n_bodies, 3)); av[1:] = rng.normal(scale=1.5, size=(n_bodies, 3))
    rb = np.zeros(m, np.uint8); rb[0] = F.RB_STATIC
    inv_mass = np.full(m, 1.0 / 4.0); inv_mass[0] = 0
    ii = np.zeros((m, 6)); ii[1:] = [0.6, 0, 0, 0.6, 0, 0.6]
    com = np.zeros((m, 3)); com[1:] = rng.uniform(-0.15, 0.15, (n_bodies, 3)) * (rng.random((n_bodies, 1)) < 0.5)
    bodies = dict(position=pos, rotation=rot, linear_velocity=lv, angular_velocity=av, inv_mass=inv_mass, inv_inertia_local=ii, rb_type=rb, center_of_mass=com)
    ent, body, shape, he, child, lt, lr = [500], [0], [F.SHAPE_CUBOID], [[30, 0.5, 30]], [0], [[0, 0, 0]], [[0, 0, 0, 1]]
    e = 501
    for b in range(1, m):
        ent.append(e); e += 1; body.append(b); shape.append(F.SHAPE_CUBOID); he.append(list(rng.uniform(0.25, 0.4, 3))); child.append(0); lt.append([0, 0, 0]); lr.append([0, 0, 0, 1])
        for _ in range(int(rng.integers(1, 4))):
            is_ball = balls and rng.random() < 0.35
            ent.append(e); e += 1; body.append(b); shape.append(F.SHAPE_BALL if is_ball else F.SHAPE_CUBOID)
            h = rng.uniform(0.2, 0.45, 3)
            he.append([h[0], 0, 0] if is_ball else list(h))
            child.append(1); lt.append(list(rng.uniform(-0.7, 0.7, 3))); lr.append(list(random_unit_quats(rng, 1)[0]))
    colliders = dict(entity_index=np.array(ent, np.uint32), body=np.array(body, np.int32), shape=np.array(shape, np.uint8), half_extents=np.array(he, float))
    transforms = dict(is_child=np.array(child, np.uint8), translation=np.array(lt, float), rotation=np.array(lr, float))
    return bodies, colliders, transforms


def compound_world(lib, bits, bodies, colliders, transforms, substeps=4, friction=0.6, speculative_margin=None):
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    cols = dict(colliders)
    if speculative_margin is not None:
        cols["speculative_margin"] = np.full(len(colliders["shape"]), speculative_margin, float)
    w.bodies_upload(**bodies); w.colliders_upload(**cols)
    w.collider_transforms_upload(**transforms)
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction, restitution=0.0)
    return w


def assert_same_compound_step(a: F.World, b: F.World, step):
    x, y = a.bodies_download(), b.bodies_download()
    for k in x:
        assert np.array_equal(x[k], y[k]), f"step {step}: bodies.{k}"
    mna, mxa, _ = a.aabbs_download(); mnb, mxb, _ = b.aabbs_download()
    assert np.array_equal(mna, mnb) and np.array_equal(mxa, mxb), f"step {step}: ColliderAabb"
    assert np.array_equal(a.pairs_get(), b.pairs_get()), f"step {step}: the new pairs (order included)"
    assert np.array_equal(a.pipeline_new_pair_ids(), b.pipeline_new_pair_ids())
    (oa, ha), (ob, hb) = a.pipeline_handles(), b.pipeline_handles()
    assert np.array_equal(oa, ob) and np.array_equal(ha, hb), f"step {step}: colour lists"
    ids = np.sort(ha)
    ra, rb = a.contacts_download(ids), b.contacts_download(ids)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), f"step {step}: contact rows.{k}"


def collider_poses_f64(bodies_now, colliders, transforms):
    """Position / Rotation of every collider in float64 from textbook quaternion algebra (an independent derivation of update_child_collider_position, not its operation order)."""
    def qmul(a, b):
        ax, ay, az, aw = a; bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
    def rotm(q):
        x, y, z, w = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    pos, rot = [], []
    for i, b in enumerate(colliders["body"]):
        bp = np.asarray(bodies_now["position"][b], np.float64); br = np.asarray(bodies_now["rotation"][b], np.float64)
        if transforms["is_child"][i]:
            p = bp + rotm(br) @ np.asarray(transforms["translation"][i], np.float64)
            r = qmul(br, np.asarray(transforms["rotation"][i], np.float64)); r = r / np.linalg.norm(r)
        else:
            p, r = bp, br
        pos.append(p); rot.append(r)
    return np.array(pos), np.array(rot), rotm
