"""Synthetic scenes (numpy) for tests and ``bench.py``: deterministic inputs of the shape the reference's
benches use (``benches/src/dim3/large_pyramid.rs:15-40``, ``many_pyramids.rs:15-64``, ``src/tests/mod.rs:51-85``)
and BASELINE.json's configs.

The narrow phase is OUT of the hot path (parry3d): contact manifolds are produced here by a
small axis-aligned box/box face-contact generator (unrotated cuboids only) purely as solver INPUT.  It is not a
restatement of parry and nothing is claimed about its parity.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _ffi as F


@dataclass
class Scene:
    """Host-side ECS tables of one scene (float64 masters; converted to the world's scalar on upload)."""
    position: np.ndarray
    rotation: np.ndarray
    linear_velocity: np.ndarray
    angular_velocity: np.ndarray
    inv_mass: np.ndarray
    inv_inertia_local: np.ndarray
    rb_type: np.ndarray
    half_extents: np.ndarray            # cuboid half extents per body (one collider per body)
    shape: np.ndarray
    friction: float = 0.5               # Friction::default, combined with CoefficientCombine::Average
    restitution: float = 0.0
    extra: Dict[str, np.ndarray] = field(default_factory=dict)
    entity_index: Optional[np.ndarray] = None   # Entity::index() per collider (default: the body index)

    @property
    def n(self) -> int:
        return len(self.inv_mass)

    def body_kwargs(self):
        kw = dict(position=self.position, rotation=self.rotation, linear_velocity=self.linear_velocity,
                  angular_velocity=self.angular_velocity, inv_mass=self.inv_mass,
                  inv_inertia_local=self.inv_inertia_local, rb_type=self.rb_type)
        kw.update(self.extra)
        return kw

    def collider_kwargs(self):
        n = self.n
        ent = self.entity_index if self.entity_index is not None else np.arange(n, dtype=np.uint32)
        return dict(entity_index=ent, body=np.arange(n, dtype=np.int32), shape=self.shape, half_extents=self.half_extents)

    def subset(self, idx: np.ndarray) -> "Scene":
        """The sub-world of the bodies `idx` (ascending global indices): same relative order, GLOBAL entity indices
        (PairKeys stay comparable across ranks) — what one rank of a sharded run uploads (avian_amd/shard.py)."""
        ent = (self.entity_index if self.entity_index is not None else np.arange(self.n, dtype=np.uint32))[idx]
        return Scene(self.position[idx], self.rotation[idx], self.linear_velocity[idx], self.angular_velocity[idx],
                     self.inv_mass[idx], self.inv_inertia_local[idx], self.rb_type[idx], self.half_extents[idx], self.shape[idx],
                     self.friction, self.restitution, {k: v[idx] for k, v in self.extra.items()}, ent)


def cuboid_mass_properties(hx, hy, hz, density=1.0):
    """parry mass properties of a cuboid (collider/parry/mod.rs:509-522): m = rho*8hxhyhz, I = m/3*(hy²+hz², ...)."""
    m = density * 8.0 * hx * hy * hz
    ixx = m / 3.0 * (hy * hy + hz * hz)
    iyy = m / 3.0 * (hx * hx + hz * hz)
    izz = m / 3.0 * (hx * hx + hy * hy)
    return m, (ixx, iyy, izz)


def _assemble(centers, half, ground_center, ground_half) -> Scene:
    """Static ground (body 0) + dynamic unit-density cuboids."""
    n = len(centers) + 1
    pos = np.zeros((n, 3))
    pos[0] = ground_center
    pos[1:] = centers
    rot = np.zeros((n, 4)); rot[:, 3] = 1.0
    he = np.zeros((n, 3)); he[0] = ground_half; he[1:] = half
    m, (ixx, iyy, izz) = cuboid_mass_properties(*half)
    inv_mass = np.full(n, 1.0 / m); inv_mass[0] = 0.0
    inv_i = np.zeros((n, 6)); inv_i[1:, 0] = 1.0 / ixx; inv_i[1:, 3] = 1.0 / iyy; inv_i[1:, 5] = 1.0 / izz
    rb = np.zeros(n, np.uint8); rb[0] = F.RB_STATIC
    return Scene(pos, rot, np.zeros((n, 3)), np.zeros((n, 3)), inv_mass, inv_i, rb, he, np.zeros(n, np.uint8))


def box_stack(nx: int, ny: int, nz: int, spacing_xz: float = 1.0, y_scale: float = 0.99) -> Scene:
    """cfg2 of SURVEY.md §8d: nx*ny*nz unit cubes, y = (2j+1)*0.5*0.99 as large_pyramid.rs:28, ground 800x40x800."""
    h = 0.5
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    x = (ix - (nx - 1) * 0.5) * spacing_xz
    z = (iz - (nz - 1) * 0.5) * spacing_xz
    y = (2.0 * iy + 1.0) * h * y_scale
    centers = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    return _assemble(centers, (h, h, h), (0.0, -20.0, 0.0), (400.0, 20.0, 400.0))


def falling_grid(n_side: int = 10, spacing: float = 1.5, lowest_y: float = 2.0) -> Scene:
    """cfg1 of SURVEY.md §8d: n_side³ unit cuboids, grid spacing 1.5, ground cuboid(200,1,200) centred y=-0.5."""
    ix, iy, iz = np.meshgrid(np.arange(n_side), np.arange(n_side), np.arange(n_side), indexing="ij")
    c = np.stack([(ix.ravel() - (n_side - 1) * 0.5) * spacing, lowest_y + iy.ravel() * spacing,
                  (iz.ravel() - (n_side - 1) * 0.5) * spacing], axis=1)
    return _assemble(c, (0.5, 0.5, 0.5), (0.0, -0.5, 0.0), (100.0, 0.5, 100.0))


def large_pyramid(base_count: int = 100) -> Scene:
    """The reference's "Large Pyramid 3D" bench scene (benches/src/dim3/large_pyramid.rs:15-40), f32 arithmetic."""
    h = np.float32(0.5)
    cs = []
    for i in range(base_count):
        y = (np.float32(2.0) * np.float32(i) + np.float32(1.0)) * h * np.float32(0.99)
        for j in range(i, base_count):
            x = (np.float32(i) + np.float32(1.0)) * h + np.float32(2.0) * np.float32(j - i) * h - h * np.float32(base_count)
            cs.append((float(x), float(y), 0.0))
    return _assemble(np.array(cs), (0.5, 0.5, 0.5), (0.0, -20.0, 0.0), (400.0, 20.0, 400.0))


def many_pyramids(base_count: int = 10, row_count: int = 10, column_count: int = 10) -> Scene:
    """The reference's "Many Pyramids 3D" bench scene (benches/src/dim3/many_pyramids.rs:15-64): `row_count` static ground
    plates cuboid(w, 0.01, w) stacked `ground_delta_y` apart, each carrying `column_count` small pyramids; f32 arithmetic.
    Bodies 0..row_count-1 are the grounds (spawned first, like the reference)."""
    f = np.float32
    h = f(0.5)
    ground_delta_y = f(2.0) * h * f(base_count + 1)
    ground_width = f(2.0) * h * f(column_count) * f(base_count + 1)
    base_width = f(2.0) * h * f(base_count)
    centers = []
    for i in range(row_count):
        base_y = f(i) * ground_delta_y
        for j in range(column_count):
            center_x = -ground_width / f(2.0) + f(j) * (base_width + f(2.0) * h) + h
            for a in range(base_count):
                y = f(2 * a + 1) * h + base_y
                for b in range(a, base_count):
                    x = f(a + 1) * h + f(2.0) * f(b - a) * h + center_x - f(0.5)
                    centers.append((float(x), float(y), 0.0))
    nd = len(centers)
    n = row_count + nd
    pos = np.zeros((n, 3)); he = np.zeros((n, 3))
    for i in range(row_count):
        pos[i] = (0.0, float(f(i) * ground_delta_y), 0.0)
        he[i] = (float(ground_width) / 2, 0.005, float(ground_width) / 2)
    pos[row_count:] = centers; he[row_count:] = 0.5
    rot = np.zeros((n, 4)); rot[:, 3] = 1.0
    m, (ixx, iyy, izz) = cuboid_mass_properties(0.5, 0.5, 0.5)
    inv_mass = np.full(n, 1.0 / m); inv_mass[:row_count] = 0.0
    inv_i = np.zeros((n, 6)); inv_i[row_count:, 0] = 1.0 / ixx; inv_i[row_count:, 3] = 1.0 / iyy; inv_i[row_count:, 5] = 1.0 / izz
    rb = np.zeros(n, np.uint8); rb[:row_count] = F.RB_STATIC
    return Scene(pos, rot, np.zeros((n, 3)), np.zeros((n, 3)), inv_mass, inv_i, rb, he, np.zeros(n, np.uint8))


def cubes_test_scene() -> Scene:
    """src/tests/mod.rs:51-85: 4x4x4 cubes of side 2 dropped on an 80x1x80 floor."""
    radius = 1.0
    cs = []
    for y in range(4):
        for x in range(4):
            for z in range(4):
                cs.append(((x - 2.0) * 2.1 * radius, 10.0 * radius * y + 5.0, (z - 2.0) * 2.1 * radius))
    return _assemble(np.array(cs), (1.0, 1.0, 1.0), (0.0, -1.0, 0.0), (40.0, 0.5, 40.0))


# ---------------------------------------------------------------------------------------------------------------
def axis_aligned_manifolds(scene: Scene, pairs: np.ndarray, prediction: float = 0.005):
    """Face-contact manifolds between UNROTATED cuboids for the given (body1, body2) pairs.

    Returns a dict of per-manifold arrays in ``pairs`` order (pairs without a face contact are dropped):
    body1, body2, normal (from body1 to body2), point_count (=4), anchor1/2 [M,4,3], penetration [M,4],
    normal_speed [M,4].  Synthetic solver input only (see module docstring).
    """
    b1 = pairs[:, 0].astype(np.int64); b2 = pairs[:, 1].astype(np.int64)
    c1, c2 = scene.position[b1], scene.position[b2]
    h1, h2 = scene.half_extents[b1], scene.half_extents[b2]
    d = c2 - c1
    # per-axis penetration (positive = overlapping)
    pen = (h1 + h2) - np.abs(d)
    # contact axis: smallest penetration among the axes (all must be > -prediction to touch at all)
    touching = np.all(pen > -prediction, axis=1)
    axis = np.argmin(pen, axis=1)
    # the overlap rectangle in the two tangent axes must have positive area (face contact, not edge/corner)
    lo = np.maximum(c1 - h1, c2 - h2); hi = np.minimum(c1 + h1, c2 + h2)
    ext = hi - lo
    rows = np.arange(len(b1))
    t1 = (axis + 1) % 3; t2 = (axis + 2) % 3
    face = (ext[rows, t1] > 1e-9) & (ext[rows, t2] > 1e-9)
    keep = touching & face
    b1, b2, c1, c2, h1, h2, d, pen, axis, lo, hi, t1, t2 = (a[keep] for a in (b1, b2, c1, c2, h1, h2, d, pen, axis, lo, hi, t1, t2))
    m = len(b1); rows = np.arange(m)
    sign = np.where(d[rows, axis] >= 0.0, 1.0, -1.0)
    normal = np.zeros((m, 3)); normal[rows, axis] = sign
    # 4 corners of the overlap rectangle (counter-clockwise in (t1, t2))
    corner_t1 = np.stack([lo[rows, t1], hi[rows, t1], hi[rows, t1], lo[rows, t1]], axis=1)
    corner_t2 = np.stack([lo[rows, t2], lo[rows, t2], hi[rows, t2], hi[rows, t2]], axis=1)
    p1 = np.zeros((m, 4, 3)); p2 = np.zeros((m, 4, 3))
    for k in range(4):
        p1[rows, k, t1] = corner_t1[:, k]; p1[rows, k, t2] = corner_t2[:, k]
        p2[rows, k, t1] = corner_t1[:, k]; p2[rows, k, t2] = corner_t2[:, k]
        p1[rows, k, axis] = c1[rows, axis] + sign * h1[rows, axis]   # on the surface of body1
        p2[rows, k, axis] = c2[rows, axis] - sign * h2[rows, axis]   # on the surface of body2
    anchor1 = p1 - c1[:, None, :]   # centre of mass = body origin
    anchor2 = p2 - c2[:, None, :]
    penetration = np.repeat(pen[rows, axis][:, None], 4, axis=1)
    v1, v2 = scene.linear_velocity[b1], scene.linear_velocity[b2]
    w1, w2 = scene.angular_velocity[b1], scene.angular_velocity[b2]
    rel = (v2[:, None, :] + np.cross(w2[:, None, :], anchor2)) - (v1[:, None, :] + np.cross(w1[:, None, :], anchor1))
    normal_speed = np.einsum("mkc,mc->mk", rel, normal)
    return dict(body1=b1.astype(np.int32), body2=b2.astype(np.int32), normal=normal,
                point_count=np.full(m, 4, np.uint8), anchor1=anchor1, anchor2=anchor2, penetration=penetration,
                normal_speed=normal_speed)


def color_manifolds(lib: F.Library, mf: Dict[str, np.ndarray], rb_type: np.ndarray):
    """Greedy persistent colouring in manifold order through the host ConstraintGraph of ``lib``
    (constraint_graph.rs:163-236), then the colour-major permutation.  Returns (color_offsets, perm)."""
    g = F.ConstraintGraph(lib, len(rb_type))
    b1, b2 = mf["body1"], mf["body2"]
    s1 = rb_type[b1] == F.RB_STATIC; s2 = rb_type[b2] == F.RB_STATIC
    m = len(b1)
    hs = np.arange(m, dtype=np.uint64)
    a1 = np.ascontiguousarray(b1, np.uint32); a2 = np.ascontiguousarray(b2, np.uint32)
    f1 = np.ascontiguousarray(s1, np.uint8); f2 = np.ascontiguousarray(s2, np.uint8)
    st = lib.fn("constraint_graph_push_batch")(g.handle, m, F._ptr(hs), F._ptr(a1), F._ptr(a2), F._ptr(f1), F._ptr(f2), None)
    if st != 0:
        raise F.AvnError(st, "constraint_graph_push_batch")
    offsets, handles = g.lists()
    g.close()
    return offsets, handles.astype(np.int64)


def permute_manifolds(mf: Dict[str, np.ndarray], perm: np.ndarray) -> Dict[str, np.ndarray]:
    return {k: v[perm] for k, v in mf.items()}


def upload_manifolds(world: F.World, mf: Dict[str, np.ndarray], color_offsets: np.ndarray, friction, restitution,
                     warm_n: Optional[np.ndarray] = None, warm_t: Optional[np.ndarray] = None):
    m = len(mf["body1"])
    # (arrays already in the world's scalar type pass through untouched: a host that keeps page-locked staging buffers hands them over as they are)
    fr = friction if isinstance(friction, np.ndarray) and friction.shape == (m,) else np.broadcast_to(np.asarray(friction, dtype=np.float64), (m,))
    re = restitution if isinstance(restitution, np.ndarray) and restitution.shape == (m,) else np.broadcast_to(np.asarray(restitution, dtype=np.float64), (m,))
    world.manifolds_upload(color_offsets=color_offsets, body1=mf["body1"], body2=mf["body2"], normal=mf["normal"],
                           friction=fr, restitution=re, point_count=mf["point_count"], anchor1=mf["anchor1"],
                           anchor2=mf["anchor2"], penetration=mf["penetration"], normal_speed=mf["normal_speed"],
                           tangent_velocity=mf.get("tangent_velocity"), manifold_flags=mf.get("manifold_flags"),
                           warm_start_normal_impulse=warm_n, warm_start_tangent_impulse=warm_t)


def brute_force_pairs(scene: Scene, margin: float = 0.005) -> np.ndarray:
    """All (i<j) body pairs whose margin-grown AABBs overlap (unrotated cuboids), by a uniform grid; order =
    ascending (min.x-sorted i, j) like the broad phase would emit.  Host-side helper for scenes whose manifolds are
    built without running the device broad phase."""
    mn = scene.position - scene.half_extents - margin
    mx = scene.position + scene.half_extents + margin
    order = np.argsort(mn[:, 0], kind="stable")
    smn, smx = mn[order], mx[order]
    n = len(order)
    out = []
    # sweep in chunks with numpy searchsorted on the sorted min.x
    ends = np.searchsorted(smn[:, 0], smx[:, 0], side="right")
    for i in range(n):
        j0, j1 = i + 1, ends[i]
        if j1 <= j0:
            continue
        js = np.arange(j0, j1)
        ok = ~((smn[i, 1] > smx[js, 1]) | (smx[i, 1] < smn[js, 1]) | (smn[i, 2] > smx[js, 2]) | (smx[i, 2] < smn[js, 2]))
        js = js[ok]
        if len(js):
            out.append(np.stack([np.full(len(js), order[i]), order[js]], axis=1))
    if not out:
        return np.zeros((0, 2), np.int64)
    p = np.concatenate(out)
    both_static = (scene.rb_type[p[:, 0]] == F.RB_STATIC) & (scene.rb_type[p[:, 1]] == F.RB_STATIC)
    return p[~both_static]


def lattice_edges(first_body: int, nx: int, ny: int, nz: int) -> np.ndarray:
    """Face-neighbour body pairs of an nx*ny*nz lattice whose bodies are numbered (ix, iy, iz) row-major from
    `first_body` — a cheap structural stand-in for the broad-phase pair list when planning a partition."""
    idx = first_body + np.arange(nx * ny * nz).reshape(nx, ny, nz)
    e = [np.stack([idx[:-1].ravel(), idx[1:].ravel()], 1), np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1),
         np.stack([idx[:, :, :-1].ravel(), idx[:, :, 1:].ravel()], 1)]
    return np.concatenate(e)


def box_stacks(n_stacks: int, nx: int, ny: int, nz: int, gap: float = 6.0) -> Scene:
    """`n_stacks` independent box stacks (each as :func:`box_stack`) side by side along x on ONE static slab: the
    weak-scaling scene of bench.py --gpus N (one interaction island per stack) and of the sharding tests."""
    one = box_stack(nx, ny, nz)
    pitch = nx * 1.0 + gap
    centers = []
    for s in range(n_stacks):
        c = one.position[1:].copy()
        c[:, 0] += (s - (n_stacks - 1) * 0.5) * pitch
        centers.append(c)
    return _assemble(np.concatenate(centers), (0.5, 0.5, 0.5), (0.0, -20.0, 0.0), (400.0 + n_stacks * pitch, 20.0, 400.0))


def _xorshift64star(seed: int, count: int) -> np.ndarray:
    """xorshift64* stream (SURVEY.md §8d cfg4 seed 0x9E3779B97F4A7C15), vectorised per lane: lane k starts from
    splitmix64(seed + k) so the generator is reproducible and O(count) in numpy."""
    k = np.arange(count, dtype=np.uint64)
    z = (np.uint64(seed) + (k + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = z ^ (z >> np.uint64(31))
    x = np.where(x == 0, np.uint64(1), x)
    x ^= x >> np.uint64(12); x ^= x << np.uint64(25); x ^= x >> np.uint64(27)
    return x * np.uint64(0x2545F4914F6CDD1D)


def _uniform01(bits: np.ndarray) -> np.ndarray:
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sparse_mixed(n: int = 1_000_000, side: float = 1000.0, seed: int = 0x9E3779B97F4A7C15) -> Scene:
    """cfg4 of SURVEY.md §8d: `n` colliders alternating ball(r = 0.5) / cuboid(1, 1, 1), centres uniform in a cube of
    the given side, velocities uniform in [-1, 1]^3, random unit quaternions; no ground.  Broad-phase dominant."""
    with np.errstate(over="ignore"):
        r = _uniform01(_xorshift64star(seed, 10 * n)).reshape(n, 10)
    pos = (r[:, 0:3] - 0.5) * side
    vel = r[:, 3:6] * 2.0 - 1.0
    q = r[:, 6:10] * 2.0 - 1.0
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    shape = (np.arange(n) % 2 == 0).astype(np.uint8)      # even = ball, odd = cuboid
    he = np.full((n, 3), 0.5)
    m_c, (ixx, _, _) = cuboid_mass_properties(0.5, 0.5, 0.5)
    m_b = 4.0 / 3.0 * np.pi * 0.5 ** 3
    i_b = 0.4 * m_b * 0.25
    inv_mass = np.where(shape == 1, 1.0 / m_b, 1.0 / m_c)
    inv_i = np.zeros((n, 6))
    ii = np.where(shape == 1, 1.0 / i_b, 1.0 / ixx)
    inv_i[:, 0] = ii; inv_i[:, 3] = ii; inv_i[:, 5] = ii
    return Scene(pos, q, vel, np.zeros((n, 3)), inv_mass, inv_i, np.zeros(n, np.uint8), he, shape)


def stack_with_chains(nx: int = 50, ny: int = 20, nz: int = 50, n_chains: int = 100, links: int = 100, swing: bool = False):
    """cfg3 of SURVEY.md §8d: an nx*ny*nz box stack (as cfg2) plus `n_chains` chains of `links` unit-density balls
    (r = 0.06, mass properties as examples/chain_3d.rs:59) hanging beside the stack, spacing 0.132, joined by
    DistanceJoints with_limits(0.132, 0.132), compliance 1e-5, anchors at the body centres, first link kinematic.
    Returns (scene, joints dict).  Chain bodies carry no collider interaction with the stack (they hang clear of it).

    ``swing=True`` (the closed-loop parity scene): the same bodies and joints, placed so that chains and stack INTERACT within a
    few steps -- the first half of the chains hangs 0.25..1.45 m off the stack's +x face with a pendulum-like velocity towards it
    (the lower links hit the face first), the second half hangs above the stack, lowest link 0.2 m over the top layer, and is
    lowered onto it by its kinematic first link at 3 m/s."""
    base = box_stack(nx, ny, nz)
    r = 0.06
    spacing = 0.132
    m = 4.0 / 3.0 * np.pi * r ** 3
    inertia = 0.4 * m * r * r
    n0 = base.n
    nb = n_chains * links
    pos = np.zeros((nb, 3))
    cx = np.arange(n_chains) // 10; cz = np.arange(n_chains) % 10
    top = 0.99 * ny + 30.0
    for c in range(n_chains):
        k = np.arange(links)
        pos[c * links:(c + 1) * links, 0] = nx * 0.5 + 5.0 + cx[c] * 1.0
        pos[c * links:(c + 1) * links, 1] = top - k * spacing
        pos[c * links:(c + 1) * links, 2] = (cz[c] - 4.5) * 1.0
    n = n0 + nb
    sc = Scene(np.concatenate([base.position, pos]), np.concatenate([base.rotation, np.tile([0, 0, 0, 1.0], (nb, 1))]),
               np.zeros((n, 3)), np.zeros((n, 3)), np.concatenate([base.inv_mass, np.full(nb, 1.0 / m)]),
               np.concatenate([base.inv_inertia_local, np.tile([1.0 / inertia, 0, 0, 1.0 / inertia, 0, 1.0 / inertia], (nb, 1))]),
               np.concatenate([base.rb_type, np.zeros(nb, np.uint8)]), np.concatenate([base.half_extents, np.full((nb, 3), r)]),
               np.concatenate([base.shape, np.ones(nb, np.uint8)]))
    first = n0 + np.arange(n_chains) * links
    if swing:
        k = np.arange(links)
        stack_top = 0.99 * ny
        for c in range(n_chains):
            rows = slice(n0 + c * links, n0 + (c + 1) * links)
            if c < n_chains // 2:      # beside the +x face, swinging in
                sc.position[rows, 0] = nx * 0.5 + 0.25 + (c // 10) * 0.3
                sc.position[rows, 1] = 1.0 + (links - 1 - k) * spacing
                sc.position[rows, 2] = ((c % 10) - 4.5) * 1.0 + 0.37
                sc.linear_velocity[rows, 0] = -4.0 * k / links
            else:                      # above the top layer, lowered onto it
                d = c - n_chains // 2
                sc.position[rows, 0] = ((d // 10) - 2.0) * 3.0 + 0.21
                sc.position[rows, 1] = stack_top + 0.2 + r + (links - 1 - k) * spacing
                sc.position[rows, 2] = ((d % 10) - 4.5) * 3.0 + 0.13
                sc.linear_velocity[rows, 1] = -3.0
    sc.rb_type[first] = F.RB_KINEMATIC
    b1 = (n0 + np.arange(nb)).reshape(n_chains, links)[:, :-1].ravel()
    J = len(b1)
    joints = dict(body1=b1.astype(np.int32), body2=(b1 + 1).astype(np.int32), local_anchor1=np.zeros((J, 3)),
                  local_anchor2=np.zeros((J, 3)), limit_min=np.full(J, spacing), limit_max=np.full(J, spacing),
                  compliance=np.full(J, 1e-5))
    return sc, joints
