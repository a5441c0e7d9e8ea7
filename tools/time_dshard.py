#!/usr/bin/env python3
"""The device closed loop sharded by islands (avn_dshard_*) on ONE GPU, as a cost model of a rank: cfg2 with two stacks (200 000 boxes) as a single world, and as two
worlds of which each simulates one stack.  Reported per settled step: the single world's time; a RANK's avn_step (the replicated front on 200 000 bodies + its own
solver on 100 000) timed alone; its exchange (pack, 6.4 MB through the host here, unpack); the bytes the host reads.  Not a scaling number -- the two ranks share
one device and run one after the other; it says what a rank costs.  usage: python tools/time_dshard.py [steps=120]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def world(lib, sc):
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    return w


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    lib = avian_amd.load_library()
    sc = scenes.box_stacks(2, 50, 40, 50, gap=60.0)   # (far enough apart for the two piles never to meet while they spread)
    owner = np.full(sc.n, -1, np.int32); owner[1:100_001] = 0; owner[100_001:] = 1
    ref = world(lib, sc)
    t_ref = []
    for s in range(steps):
        t0 = time.perf_counter(); ref.step(); ref.synchronize(); t_ref.append((time.perf_counter() - t0) * 1e3)
    m_ref = ref.pipeline_stats().manifolds
    ref.close()
    ranks = [world(lib, sc) for _ in range(2)]
    for r, w in enumerate(ranks): w.dshard_enable(2, r, owner)
    t_step, t_x = [[], []], []
    for s in range(steps):
        for r, w in enumerate(ranks):
            t0 = time.perf_counter(); w.step(); w.synchronize(); t_step[r].append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        recs = [w.dshard_bodies_pack() for w in ranks]
        ranks[0].dshard_bodies_unpack(1, recs[1]); ranks[1].dshard_bodies_unpack(0, recs[0])
        t_x.append((time.perf_counter() - t0) * 1e3 / 2)
    d = [w.dshard_stats() for w in ranks]
    a = slice(steps - 20, steps)
    print(f"single world, 200 000 boxes, steps {steps - 20}..{steps - 1}: {np.mean(t_ref[a]):.3f} ms/step, {m_ref} manifolds")
    for r in range(2):
        print(f"rank {r} of 2: avn_step {np.mean(t_step[r][a]):.3f} ms (ratio to the single world {np.mean(t_step[r][a]) / np.mean(t_ref[a]):.3f}), own manifolds {d[r].own_manifolds} of {d[r].global_manifolds}, "
              f"sends {d[r].bytes_sent_per_step / 1e6:.1f} MB per step, launches {ranks[r].timers().kernel_launches}")
    print(f"host-mediated exchange (pack + D2H + H2D + unpack, per rank): {np.mean(t_x[a]):.3f} ms; inside avn_step with avn_comm_init it is one ncclAllGather on the world's stream")


if __name__ == "__main__":
    main()
