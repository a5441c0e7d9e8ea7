#!/usr/bin/env python3
"""Time the device broad phase (UPDATE_AABB + COLLECT_COLLISION_PAIRS, steady state: every pair already in the pair set)
on the cfg2 stack and the cfg4 sparse scene.  usage: python tools/time_broadphase.py [n_sparse]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def run(name, sc):
    w = F.World(avian_amd.load_library(), F.default_config(32))
    w.bodies_upload(**sc.body_kwargs())
    w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB")
    ms0, _ = w.profile_system("COLLECT_COLLISION_PAIRS", 1)
    n_pairs = len(w.pairs_get())
    ms, launches = w.profile_system("COLLECT_COLLISION_PAIRS", 10)
    print(f"{name}: colliders {sc.n}, first frame {ms0:.3f} ms ({n_pairs} pairs), steady {ms / 10:.3f} ms/frame, "
          f"{sc.n / (ms / 10) / 1e3:.1f} M AABB/s, {launches // 10} launches")


if __name__ == "__main__":
    run("cfg2 stack 100k", scenes.box_stack(50, 40, 50))
    run("cfg4 sparse", scenes.sparse_mixed(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000))
