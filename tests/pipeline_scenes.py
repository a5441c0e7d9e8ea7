"""Closed-loop scenes for the device narrow phase: bodies start apart / overlapping / tumbling so that contacts start,
persist (warm-start matching), change their point count and stop over a few steps."""
from __future__ import annotations

import numpy as np

from avian_amd import scenes
from helpers import F, random_unit_quats


def dropped_boxes(seed=0, n=60, balls=True):
    """n cuboids / balls dropped in a loose pile over a static slab, random orientations and spins."""
    rng = np.random.default_rng(seed)
    sc = scenes.box_stack(1, 1, 1)  # ground + one cube: reuse its ground definition
    ground_he = sc.half_extents[0]; ground_pos = sc.position[0]
    m = n + 1
    pos = np.zeros((m, 3)); he = np.zeros((m, 3)); shape = np.zeros(m, np.uint8)
    pos[0] = ground_pos; he[0] = ground_he
    side = int(np.ceil(n ** (1 / 3)))
    k = 0
    for i in range(side):
        for j in range(side):
            for l in range(side):
                if k >= n:
                    break
                pos[1 + k] = [1.3 * i + rng.uniform(-0.1, 0.1), 0.8 + 1.25 * j, 1.3 * l + rng.uniform(-0.1, 0.1)]
                k += 1
    he[1:] = rng.uniform(0.3, 0.6, (n, 3))
    if balls:
        b = rng.random(n) < 0.3
        shape[1:][b] = F.SHAPE_BALL
        he[1:][b, 1:] = 0.0
    rot = np.tile([0.0, 0, 0, 1], (m, 1))
    rot[1:] = random_unit_quats(rng, n)
    rot[1 + np.flatnonzero(rng.random(n) < 0.3)] = [0, 0, 0, 1]
    lv = np.zeros((m, 3)); av = np.zeros((m, 3))
    lv[1:] = rng.normal(scale=0.5, size=(n, 3)); av[1:] = rng.normal(scale=1.0, size=(n, 3))
    rb = np.zeros(m, np.uint8); rb[0] = F.RB_STATIC
    vol = np.where(shape == F.SHAPE_BALL, 4 / 3 * np.pi * he[:, 0] ** 3, 8 * he.prod(1).clip(1e-9))
    inv_mass = 1.0 / vol; inv_mass[0] = 0.0
    ii = np.zeros((m, 6))
    for i in range(1, m):
        mass = vol[i]
        if shape[i] == F.SHAPE_BALL:
            d = [2 / 5 * mass * he[i, 0] ** 2] * 3
        else:
            hx, hy, hz = he[i]
            d = [mass / 3 * (hy * hy + hz * hz), mass / 3 * (hx * hx + hz * hz), mass / 3 * (hx * hx + hy * hy)]
        ii[i] = [1 / d[0], 0, 0, 1 / d[1], 0, 1 / d[2]]
    bodies = dict(position=pos, rotation=rot, linear_velocity=lv, angular_velocity=av, inv_mass=inv_mass, inv_inertia_local=ii, rb_type=rb)
    colliders = dict(entity_index=np.arange(m, dtype=np.uint32) + 100, body=np.arange(m, dtype=np.int32), shape=shape, half_extents=he)
    return bodies, colliders


def stack_and_projectile(nx=3, ny=3, nz=3, height=32.0, offset=(0.3, 0.0, 0.2)):
    """A small box stack (settles and falls asleep within a second or two) and one more box high above it that lands on it later:
    bodies = [ground, the stack ..., the projectile]."""
    base = scenes.box_stack(nx, ny, nz)
    centers = np.vstack([base.position[1:], [[offset[0], height, offset[2]]]])
    sc = scenes._assemble(centers, (0.5, 0.5, 0.5), base.position[0], base.half_extents[0])
    return sc


def stack_chain_and_projectile(nx=3, ny=3, nz=3, links=8, height=30.0, link_r=0.2, spacing=0.5):
    """Sleeping WITH joints: a small box stack, a chain of `links` balls joined by DistanceJoints whose first link is kinematic (at rest) and
    whose lower links lie on the stack's top layer -- joints and contacts in ONE island (PhysicsIslands::add_joint, islands/mod.rs:668-735) --,
    and a box high above that lands on the stack later and wakes the whole island.  bodies = [ground, stack ..., chain links ..., projectile];
    returns (scene, joints dict)."""
    base = scenes.box_stack(nx, ny, nz)
    top = 0.99 * ny
    n0 = base.n
    # the chain hangs over the stack's centre column; the last three links already rest on the top layer, side by side along x
    k = np.arange(links)
    hang = links - 3
    pos = np.zeros((links, 3))
    pos[:hang, 0] = 0.0; pos[:hang, 1] = top + link_r + (hang - k[:hang]) * spacing; pos[:hang, 2] = 0.1
    pos[hang:, 0] = (k[hang:] - hang + 1) * spacing * 0.9; pos[hang:, 1] = top + link_r; pos[hang:, 2] = 0.1
    m = 4.0 / 3.0 * np.pi * link_r ** 3
    inertia = 0.4 * m * link_r * link_r
    proj = np.array([[0.3 - 1.0, height, -0.8]])
    n = n0 + links + 1
    mc, (ixx, iyy, izz) = scenes.cuboid_mass_properties(0.5, 0.5, 0.5)
    sc = scenes.Scene(np.concatenate([base.position, pos, proj]), np.concatenate([base.rotation, np.tile([0, 0, 0, 1.0], (links + 1, 1))]),
                      np.zeros((n, 3)), np.zeros((n, 3)),
                      np.concatenate([base.inv_mass, np.full(links, 1.0 / m), [1.0 / mc]]),
                      np.concatenate([base.inv_inertia_local, np.tile([1.0 / inertia, 0, 0, 1.0 / inertia, 0, 1.0 / inertia], (links, 1)), [[1.0 / ixx, 0, 0, 1.0 / iyy, 0, 1.0 / izz]]]),
                      np.concatenate([base.rb_type, np.zeros(links + 1, np.uint8)]),
                      np.concatenate([base.half_extents, np.full((links, 3), link_r), [[0.5, 0.5, 0.5]]]),
                      np.concatenate([base.shape, np.full(links, F.SHAPE_BALL, np.uint8), [0]]).astype(np.uint8))
    sc.rb_type[n0] = F.RB_KINEMATIC
    b1 = n0 + np.arange(links - 1)
    d = np.linalg.norm(sc.position[b1 + 1] - sc.position[b1], axis=1)
    J = links - 1
    joints = dict(body1=b1.astype(np.int32), body2=(b1 + 1).astype(np.int32), local_anchor1=np.zeros((J, 3)), local_anchor2=np.zeros((J, 3)),
                  limit_min=d.copy(), limit_max=d.copy(), compliance=np.full(J, 1e-5), collision_disabled=np.ones(J, np.uint8))
    return sc, joints
