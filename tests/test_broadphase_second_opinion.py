"""CPU: the oracle's sweep-and-prune against an O(n^2) numpy statement of what a broad phase must return (reference
collision/broad_phase.rs:345-475): the SET of pairs = every two colliders of different bodies, not both static, whose ColliderAabbs intersect
(closed intervals: touching counts) -- and the SEQUENCE = the sweep's: pairs grouped by their earlier member in ascending (min.x, previous
order), candidates in the same order.  The AABBs themselves are checked against the closed form for resting cuboids and balls
(collider/backend.rs:498-624: extents grown by the contact tolerance)."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, oracle_lib


def brute(mn, mx, static, body):
    n = len(mn)
    hit = np.all(mn[:, None, :] <= mx[None, :, :], axis=2) & np.all(mx[:, None, :] >= mn[None, :, :], axis=2)
    hit &= ~(static[:, None] & static[None, :]) & (body[:, None] != body[None, :])
    i, j = np.nonzero(np.triu(hit, 1))
    return set(zip(i.tolist(), j.tolist()))


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("scene", ["sparse", "lattice"])
def test_pair_set_and_sequence(bits, scene):
    lib = oracle_lib()
    sc = scenes.sparse_mixed(500, side=9.0) if scene == "sparse" else scenes.box_stack(6, 5, 6)
    w = F.World(lib, F.default_config(bits, substeps=1))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get()
    mn, mx, order = w.aabbs_download()
    mn = mn.astype(np.float64); mx = mx.astype(np.float64)
    static = np.asarray(sc.rb_type) == F.RB_STATIC
    want = brute(mn, mx, static, np.arange(sc.n))
    got = [(int(a), int(b)) for a, b in zip(p["collider1"], p["collider2"])]
    assert len(got) == len(set(got)) and {tuple(sorted(g)) for g in got} == want and len(want) > 100
    # the sequence: position of every collider in the sorted interval order
    pos = np.full(sc.n, -1); pos[order] = np.arange(len(order))
    assert np.all(np.diff(mn[order, 0]) >= 0), "the persistent order is sorted by min.x after the sweep"
    k1 = pos[[g[0] for g in got]]; k2 = pos[[g[1] for g in got]]
    assert np.all(k1 < k2), "collider1 is the EARLIER interval of the pair (broad_phase.rs:443)"
    key = k1.astype(np.int64) * (sc.n + 1) + k2
    assert np.all(np.diff(key) > 0), "pairs come i-major, candidates ascending: the sweep's emission order"
    # second frame: nothing new (every pair is in the pair set now), third after a shuffle of velocities: only NEW overlaps
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    assert len(w.pairs_get()) == 0


def test_resting_aabbs_have_the_closed_form():
    lib = oracle_lib()
    sc = scenes.sparse_mixed(200, side=30.0)
    sc.linear_velocity[:] = 0; sc.angular_velocity[:] = 0
    w = F.World(lib, F.default_config(64, substeps=1))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.run_system("UPDATE_AABB")
    mn, mx, _ = w.aabbs_download()
    tol = 0.005
    for i in range(sc.n):
        q = sc.rotation[i]; he = sc.half_extents[i]
        if sc.shape[i] == 1:
            ext = np.full(3, he[0])
        else:
            x, y, z, ww = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                          [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                          [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
            ext = np.abs(R) @ he
        assert np.allclose(mn[i], sc.position[i] - ext - tol, atol=1e-9) and np.allclose(mx[i], sc.position[i] + ext + tol, atol=1e-9), i
