"""GPU: the island-sharded closed loop with replicated integer bookkeeping (avian_amd/shard.py: ShardedClosedLoop) on the HIP backend -- two
worlds on ONE device, each running only its pile's physics, the exchanges handed around in-process -- against the single HIP world and the
oracle, tolerance 0, through ContactId reuse and swap_removes across ranks (tests/test_sharded_closed_loop_cpu.py is the same on the oracle
and over gloo).  No multi-GPU hardware is involved: what this pins is that the SHARDING changes no bit."""
import numpy as np
import pytest

from avian_amd import shard
from helpers import F, hip_lib, oracle_lib
from test_sharded_closed_loop_cpu import compare, piles, plan_by_pile, single_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [32, 64])
def test_two_sharded_worlds_on_one_device_equal_the_single_world(bits):
    n = 24
    bodies, colliders = piles(2, n)
    ref_o = single_world(oracle_lib(), bits, bodies, colliders)
    ref_h = single_world(hip_lib(), bits, bodies, colliders)
    p = plan_by_pile(bodies, 2, n)
    ranks = shard.sharded_closed_loop_worlds(hip_lib(), bits, bodies, colliders, p)
    loops = [r[1] for r in ranks]
    for s in range(90):
        ref_o.step(); ref_h.step()
        shard.step_in_process(loops)
        compare(s, ref_h, ranks)
        bo, bh = ref_o.bodies_download(), ref_h.bodies_download()
        for k in bo:
            assert np.array_equal(bo[k], bh[k]), f"step {s}: the single HIP world left the oracle: {k}"
    st = ref_h.pipeline_stats()
    assert st.pairs_removed > 0 and max(loops[0].pairs) < st.pairs_added


def test_native_sharded_closed_loop_many_pyramids_on_one_device_and_its_bookkeeping_cost():
    """Round 5: the bookkeeping in the library (avn_shard_*), flat payloads.  The reference's Many Pyramids scene (5 500 boxes, 100 islands) as 4 sub-worlds
    on ONE device against the single HIP world every step for 60 steps; and what a step of the replicated bookkeeping costs the host next to the step
    itself (printed)."""
    import time
    from avian_amd import scenes
    base, rows, cols = 10, 10, 10
    sc = scenes.many_pyramids(base, rows, cols)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    per = base * (base + 1) // 2
    world = 4
    pyramid = (np.arange(sc.n) - rows) // per
    rank = np.where(np.arange(sc.n) < rows, -1, pyramid * world // (rows * cols)).astype(np.int32)
    p = shard.ShardPlan(world, rank.copy(), rank, rows * cols)
    single = single_world(hip_lib(), 32, bodies, colliders)
    ranks = shard.sharded_closed_loop_worlds(hip_lib(), 32, bodies, colliders, p, native=True)
    loops = [r[1] for r in ranks]
    t_book = 0.0
    for s in range(60):
        single.step()
        p1 = [l.phase1() for l in loops]
        kc, kx, pr = (np.concatenate([x[i] for x in p1]) for i in range(3))
        ch = np.concatenate([l.phase2(kc, kx, pr) for l in loops])
        t0 = time.perf_counter()
        for l in loops:
            l.shard.phase3(ch)          # the integer replay alone (every rank replays every rank's changes), timed
        t_book += time.perf_counter() - t0
        for l in loops:                 # ... then the world calls of phase 3, from the state the replay left
            off, handles = l.shard.handles()
            rem = F.Shard._u32(*_removed(l))
            if len(rem):
                l.w.contact_pairs_remove(rem); l.w.active_pairs_set(l.shard.active())
            l.w.manifold_handles_upload(off, handles); l.w.run_system("SOLVER")
        compare(s, single, ranks)
    print(f"replicated bookkeeping (phase 3 of 4 ranks, {int(single.pipeline_stats().manifolds)} manifolds): {t_book / 60 / world * 1e3:.3f} ms per rank per step")
    assert single.pipeline_stats().manifolds > 10_000


def _removed(loop):
    a, n = F.vp(), F.C.c_size_t()
    assert loop.lib.fn("shard_removed_local")(loop.shard.handle, F.C.byref(a), F.C.byref(n)) == 0
    return a, n.value
