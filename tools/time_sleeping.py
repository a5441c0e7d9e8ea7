#!/usr/bin/env python3
"""The bench's sleeping scene (Many Pyramids 3D + one dropped box, avn_sleeping_enable) stand-alone: ms per step, awake bodies and launches per
window of 50 steps.  usage: python tools/time_sleeping.py [windows=6]   (under rocprofv3 --kernel-trace: tools/step_timeline-style analysis)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lib = avian_amd.load_library()
    base = scenes.many_pyramids(10, 10, 10)
    top = base.position[10:][np.argmax(base.position[10:, 1])]
    extra = np.array([[top[0] + 0.2, top[1] + 35.0, top[2] + 0.1]])
    sc = scenes.Scene(np.vstack([base.position, extra]), np.vstack([base.rotation, [[0, 0, 0, 1.0]]]), np.vstack([base.linear_velocity, [[0, 0, 0.0]]]),
                      np.vstack([base.angular_velocity, [[0, 0, 0.0]]]), np.append(base.inv_mass, base.inv_mass[-1]), np.vstack([base.inv_inertia_local, base.inv_inertia_local[-1:]]),
                      np.append(base.rb_type, 0).astype(np.uint8), np.vstack([base.half_extents, [[0.5, 0.5, 0.5]]]), np.append(base.shape, 0).astype(np.uint8))
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable(); w.sleeping_enable()
    for wi in range(windows):
        t0 = time.perf_counter(); awake = launches = changes = 0
        for _ in range(50):
            w.step()
            awake += w.sleeping_stats().n_awake_bodies; launches += w.timers().kernel_launches; changes += w.pipeline_stats().last_status_changes
        w.synchronize()
        st = w.sleeping_stats(); tm = w.timers()
        print(f"steps {wi * 50}..{wi * 50 + 49}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step, awake {awake / 50:.0f}, launches/step {launches / 50:.1f}, changes/step {changes / 50:.1f}, "
              f"sleeping islands {st.islands.n_sleeping_islands}/{st.islands.n_islands}, island blocks {tm.island_blocks}, manifolds {w.pipeline_stats().manifolds}", flush=True)


if __name__ == "__main__":
    main()
