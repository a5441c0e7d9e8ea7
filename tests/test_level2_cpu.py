"""CPU (oracle backend): level-2 sharding -- ONE contact island split into x-slab worlds with a global colouring and a halo exchange of
boundary-body velocities after every colour (include/avian_mi355x.h avn_halo_plan, avian_amd/shard.py level2_*) -- is bit-identical to the
unsplit world: bodies of every slab and impulses of every owned manifold, after several steps, f32 and f64, 2 / 3 / 4 slabs."""
import numpy as np
import pytest

from avian_amd import shard
from helpers import F, oracle_lib
from level2_helpers import compare_with_single, global_problem, make_single, make_split, step_split_in_process


@pytest.mark.parametrize("bits,world_size,restitution", [(32, 2, 0.0), (32, 3, 0.3), (64, 2, 0.3), (32, 4, 0.0)])
def test_split_island_equals_single_world(bits, world_size, restitution):
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=bits + world_size)
    single = make_single(lib, bits, sc, pm, offs, restitution, 3)
    plan, worlds = make_split(lib, bits, sc, pm, offs, restitution, 3, world_size)
    assert sum(len(p.manifolds) for p in plan) == len(pm["body1"]) and all(len(p.manifolds) for p in plan)
    assert sum(len(p.send_bodies) for p in plan) == sum(len(p.recv_bodies) for p in plan) > 0, "the slabs must actually share bodies"
    for step in range(3):
        single.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, restitution > 0)
        compare_with_single(single, plan, worlds)
    v = single.bodies_download()["linear_velocity"]
    assert float(np.abs(v).max()) > 0.05


def test_plan_properties():
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 9, 3, 4)
    plan = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, 3)
    owned = np.concatenate([p.manifolds for p in plan])
    assert np.array_equal(np.sort(owned), np.arange(len(pm["body1"]))), "every manifold has exactly one owner"
    for r, p in enumerate(plan):
        assert np.all(np.diff(p.bodies) > 0) and 0 in p.bodies, "ascending bodies, the static ground everywhere"
        n_p = len(p.peers)
        for c in range(24):
            for k, q in enumerate(p.peers):
                other = plan[int(q)]
                k2 = int(np.flatnonzero(other.peers == r)[0])
                a = p.bodies[p.send_bodies[p.send_offsets[c * n_p + k]:p.send_offsets[c * n_p + k + 1]]]
                b = other.bodies[other.recv_bodies[other.recv_offsets[c * len(other.peers) + k2]:other.recv_offsets[c * len(other.peers) + k2 + 1]]]
                assert np.array_equal(a, b), "send list of one side == receive list of the other, same order (global body ids)"
    # a body shared between slabs in the overflow colour is refused
    offs_bad = offs.copy(); offs_bad[:] = 0; offs_bad[24] = len(pm["body1"])   # everything in colour 23
    with pytest.raises(ValueError):
        shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs_bad, 2)


def test_library_planner_equals_the_numpy_planner():
    """avn_level2_plan_* of the product (host C++), of the oracle (a differently organised implementation) and shard.level2_plan (numpy)
    give the same plan: bodies, manifolds, colour offsets, peers and every send / receive list."""
    from helpers import hip_lib
    lib = oracle_lib()
    for dims, R in (((9, 3, 4), 3), ((8, 4, 5), 2), ((6, 2, 2), 4), ((5, 2, 2), 1)):
        sc, pm, offs, _ = global_problem(lib, *dims)
        want = shard.level2_plan(sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R)
        for L in (hip_lib(), lib):
            got = shard.level2_plan_lib(L, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, R)
            assert len(got) == len(want) == R
            for a, b in zip(got, want):
                for f in ("bodies", "manifolds", "color_offsets", "peers", "send_offsets", "send_bodies", "recv_offsets", "recv_bodies"):
                    assert np.array_equal(np.asarray(getattr(a, f)).astype(np.int64), np.asarray(getattr(b, f)).astype(np.int64)), (dims, R, L.prefix, f)
    sc, pm, offs, _ = global_problem(lib, 9, 3, 4)
    offs_bad = offs.copy(); offs_bad[:] = 0; offs_bad[24] = len(pm["body1"])
    for L in (hip_lib(), lib):
        with pytest.raises(ValueError):
            shard.level2_plan_lib(L, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs_bad, 2)


def test_level2_over_gloo_world_size_2(tmp_path):
    """Two real processes (torch.distributed, gloo): each builds only its slab, steps it with shard.level2_solver and exchanges the boundary
    records point to point after every colour; merged result == the single world, bit for bit."""
    from test_shard_gloo import launch
    out = str(tmp_path / "level2")
    launch("level2", out, 3)
    lib = oracle_lib()
    sc, pm, offs, _ = global_problem(lib, 8, 4, 5, seed=7)
    single = make_single(lib, 32, sc, pm, offs, 0.3, 3)
    for _ in range(3):
        single.run_system("SOLVER")
    ref, imp = single.bodies_download(), single.impulses_download()
    seen = np.zeros(len(pm["body1"]), bool)
    for r in range(2):
        d = np.load(out + f".rank{r}.npz")
        for k in ref:
            assert np.array_equal(ref[k][d["bodies"]], d["b_" + k]), f"rank {r}: bodies.{k}"
        for k in imp:
            assert np.array_equal(imp[k][d["manifolds"]], d["i_" + k]), f"rank {r}: impulses.{k}"
        seen[d["manifolds"]] = True
    assert seen.all()
