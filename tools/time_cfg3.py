#!/usr/bin/env python3
"""cfg3 (50 000 cuboids + 100 chains of 100 links, 9 900 distance joints): frozen-manifold step and the device closed loop (steps 60..79), ms per step.
With the `make measure` build AVN_NO_JOINT_LDS=1 selects the global-memory joint walk (A/B of k_joint_schedule_lds).  usage: python tools/time_cfg3.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def main():
    lib = avian_amd.load_library()
    sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100)
    joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.distance_joints_upload(**joints)
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    for _ in range(60):
        w.step()
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w.step(); w.synchronize()
    print(f"cfg3 closed loop, steps 60..79: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step, launches {w.timers().kernel_launches}, manifolds {w.pipeline_stats().manifolds}", flush=True)
    ms, n = w.profile_system("XPBD_SOLVE", 20)
    print(f"XPBD_SOLVE alone: {ms / 20 * 1e3:.1f} us per pass ({n // 20} launches)", flush=True)
    w.close()


if __name__ == "__main__":
    main()
