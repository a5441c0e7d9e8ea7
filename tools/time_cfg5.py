#!/usr/bin/env python3
"""cfg5 (500 000 cuboids, f64, 8 substeps) in the device closed loop: ms per step while the lattice collapses (steps 3..12) and once it has settled (steps 100..119).
With the `make measure` build AVN_OVF_TICKETS=1 selects the ticket hand-over of the overflow colour instead of the mailboxes (A/B).  usage: python tools/time_cfg5.py [settled=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def main():
    settled = len(sys.argv) < 2 or sys.argv[1] != "0"
    lib = avian_amd.load_library()
    sc = scenes.box_stack(100, 50, 100)
    w = F.World(lib, F.default_config(64, substeps=8))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()

    def window(n, label):
        w.synchronize(); t0 = time.perf_counter(); ovf = 0
        for _ in range(n):
            w.step(); w.synchronize(); ovf = max(ovf, w.pipeline_stats().last_overflow_manifolds)
        ps = w.pipeline_stats()
        print(f"cfg5 closed loop, {label}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step, manifolds {ps.manifolds}, overflow manifolds (max) {ovf}, changes in the last step {ps.last_status_changes}, launches {w.timers().kernel_launches}", flush=True)
    for _ in range(3):
        w.step()
    window(10, "steps 3..12 (the lattice collapses)")
    if settled:
        for _ in range(87):
            w.step()
        window(20, "steps 100..119")
    w.close()


if __name__ == "__main__":
    main()
