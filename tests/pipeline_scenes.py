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
