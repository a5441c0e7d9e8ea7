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
