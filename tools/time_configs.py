#!/usr/bin/env python3
"""Whole-step device time of BASELINE.json's other configurations (the parity-test cases of tests/test_gpu_configs.py; the
bench line itself is cfg2): cfg1 1 000 cuboids / 1 substep, cfg3 50 k cuboids + 9 900 distance joints / 4 substeps,
cfg4 1 M sparse colliders (broad phase only: first frame and steady state), cfg5 500 k cuboids f64 / 8 substeps.
Manifolds are the fixed synthetic face manifolds of the bench (narrow phase out of path).  usage: time_configs.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import avian_amd
from avian_amd import _ffi as F, scenes


def setup(w, lib, sc, joints=None):
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    if joints is not None:
        w.distance_joints_upload(**joints)
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get().copy()
    if len(p) == 0:
        return 0, 0
    mf = scenes.axis_aligned_manifolds(sc, np.stack([p["body1"], p["body2"]], axis=1))
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, sc.restitution)
    return len(p), len(perm)


def time_steps(w, substeps, warmup=3, steps=20):
    for _ in range(warmup):
        w.step()
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    w.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = w.timers()
    return {"ms_per_step": round(dt * 1e3, 4), "substeps_per_s": round(substeps / dt, 1), "kernel_launches_per_step": int(tm.kernel_launches),
            "island_blocks": int(tm.island_blocks), "side_island_bodies": int(tm.side_island_bodies),
            "last_step_ms": {"broad_phase": round(tm.broad_phase_ms, 4), "prepare": round(tm.prepare_ms, 4), "substeps": round(tm.substeps_ms, 4), "finalize": round(tm.finalize_ms, 4)}}


def main():
    lib = avian_amd.load_library()
    out = {}
    # cfg1
    sc = scenes.falling_grid(10, 1.5, 2.0)
    w = F.World(lib, F.default_config(32, substeps=1))
    pairs, mfs = setup(w, lib, sc)
    out["cfg1_1k_cuboids_1_substep"] = dict(bodies=sc.n, pairs=pairs, manifolds=mfs, **time_steps(w, 1, steps=100))
    w.close()
    # cfg3
    sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100)
    joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
    w = F.World(lib, F.default_config(32, substeps=4))
    pairs, mfs = setup(w, lib, sc, joints)
    out["cfg3_50k_cuboids_9900_distance_joints_4_substeps"] = dict(bodies=sc.n, joints=len(joints["body1"]), pairs=pairs, manifolds=mfs, **time_steps(w, 4))
    w.close()
    # cfg4: broad phase only
    sc = scenes.sparse_mixed(1_000_000)
    w = F.World(lib, F.default_config(32))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    ms_aabb, _ = w.profile_system("UPDATE_AABB", 1)
    ms0, _ = w.profile_system("COLLECT_COLLISION_PAIRS", 1)
    n_pairs = len(w.pairs_get())
    ms_a, _ = w.profile_system("UPDATE_AABB", 10)
    ms, launches = w.profile_system("COLLECT_COLLISION_PAIRS", 10)
    out["cfg4_1M_sparse_broad_phase"] = {"colliders": sc.n, "first_frame_ms": round(ms0, 3), "pairs_first_frame": n_pairs,
                                         "pairs_per_s_first_frame": round(n_pairs / (ms0 / 1e3)), "update_aabb_ms": round(ms_a / 10, 4),
                                         "steady_collect_ms": round(ms / 10, 4), "aabbs_per_s_steady": round(sc.n / ((ms + ms_a) / 10 / 1e3)),
                                         "launches_per_frame": launches // 10}
    w.close()
    # cfg5
    sc = scenes.box_stack(100, 50, 100)
    w = F.World(lib, F.default_config(64, substeps=8))
    pairs, mfs = setup(w, lib, sc)
    out["cfg5_500k_cuboids_f64_8_substeps"] = dict(bodies=sc.n, pairs=pairs, manifolds=mfs, **time_steps(w, 8, warmup=2, steps=8))
    w.close()
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
