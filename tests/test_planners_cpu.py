"""CPU: the host planners behind the C ABI (avn_slab_select, avn_interval_orders_merge) -- product (host C++), oracle (written the slow
obvious way) and the numpy functions of avian_amd/shard.py that the multi-GPU tests use -- agree on random inputs, on lattices full of
equal keys, with -0.0 / +0.0, with non-finite extents and over frames with a persistent order."""
import numpy as np
import pytest

from avian_amd import shard
from helpers import F, hip_lib, oracle_lib

LIBS = [("product", hip_lib), ("oracle", oracle_lib)]


def numpy_select(mn, mx, prev, R, r):
    p = shard.slab_plan(mn, R)
    local, owned = shard.slab_colliders(p, r, mn, mx, prev)
    return local, owned, shard.slab_next_order(prev, mn, len(mn))


def cases():
    rng = np.random.default_rng(0)
    n = 300
    mn = rng.uniform(-10, 10, n); yield "random", mn, mn + rng.uniform(0.1, 3.0, n)
    lat = np.repeat(np.arange(10.0), 30); yield "lattice of equal keys", lat, lat + 1.0
    z = lat.copy(); z[::7] = -0.0; z[1::7] = 0.0; yield "signed zeros", z, z + 1.0
    bad = mn.copy(); bad[5] = np.nan; bad[9] = np.inf; bad[11] = -np.inf; mxb = bad + 1.0; mxb[20] = np.nan; yield "non-finite", bad, mxb
    span = mn.copy(); mxs = mn + 0.5; span[0] = -50.0; mxs[0] = 50.0; yield "a ground that spans every slab", span, mxs


@pytest.mark.parametrize("name,getlib", LIBS)
def test_slab_select_equals_numpy(name, getlib):
    lib = getlib()
    for label, mn, mx in cases():
        for R in (1, 2, 3, 8):
            prev = None
            for frame in range(3):
                for r in range(R):
                    want_l, want_o, want_n = numpy_select(mn, mx, prev, R, r)
                    got_l, got_o, got_n = lib.slab_select(mn, mx, prev, R, r)
                    assert np.array_equal(got_l, want_l) and np.array_equal(got_o, want_o), (name, label, R, r, frame)
                    assert np.array_equal(got_n, want_n), (name, label, R, r, frame, "next order")
                prev = want_n
                # next frame: shuffle a few keys so that the persistent order matters (ties keep LAST frame's order)
                rng = np.random.default_rng(frame + 17)
                mn = mn.copy(); idx = rng.integers(0, len(mn), 20)
                fin = np.isfinite(mn[idx]); mn[idx[fin]] = np.round(mn[idx[fin]] + rng.normal(scale=0.5, size=fin.sum()))
                mx = np.where(np.isfinite(mn), mn + 1.0, mx)
    with pytest.raises(F.AvnError):
        lib.slab_select(np.zeros(3), np.ones(3), np.array([0, 0, 1]), 2, 0)   # duplicate entry in prev_order
    with pytest.raises(F.AvnError):
        lib.slab_select(np.zeros(3), np.ones(3), None, 2, 2)                  # rank out of range


@pytest.mark.parametrize("name,getlib", LIBS)
def test_interval_orders_merge_equals_numpy(name, getlib):
    lib = getlib()
    rng = np.random.default_rng(3)
    for trial in range(20):
        L = int(rng.integers(1, 5))
        ents, keys, states = [], [], []
        pool = rng.permutation(200)
        for l in range(L):
            m = int(rng.integers(0, 40))
            e = np.concatenate([[0], pool[l * 40: l * 40 + m] + 1]) if rng.random() < 0.7 else pool[l * 40: l * 40 + m] + 1   # entity 0: on several ranks
            k = np.sort(np.round(rng.uniform(-3, 3, len(e)) * 2) / 2)                      # many ties, ascending like a sorted order
            if len(k) and rng.random() < 0.3:
                k[:] = np.nan                                                               # a rank that never swept
            if len(e) and e[0] == 0:
                k[0] = -9.0 if not np.isnan(k[0]) else np.nan
            ents.append(e.astype(np.int64)); keys.append(k)
            states.append(shard.RankState(np.zeros(0, np.int64), {}, {}, np.zeros((0, 2), np.int64), e.astype(np.int64), k))
        want = shard.merge_interval_orders(states)
        got = lib.interval_orders_merge(ents, keys)
        assert np.array_equal(got, want), (name, trial)


def test_bounds_exchange_of_a_single_rank_is_its_own_bounds():
    """avn_bounds_exchange on the oracle (no transport: one rank): the row equals avn_dynamic_bounds."""
    from avian_amd import scenes
    sc = scenes.box_stack(3, 3, 3)
    w = F.World(oracle_lib(), F.default_config(32))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.run_system("UPDATE_AABB")
    b, ov = w.bounds_exchange()
    mn, mx = w.dynamic_bounds()
    assert b.shape == (1, 6) and np.array_equal(b[0], np.concatenate([mn, mx])) and len(ov) == 0
