#!/usr/bin/env python3
"""Whole-step device time of BASELINE.json's other configurations (the parity-test cases of tests/test_gpu_configs.py; the
bench line itself is cfg2): cfg1 1 000 cuboids / 1 substep, cfg3 50 k cuboids + 9 900 distance joints / 4 substeps,
cfg4 1 M sparse colliders (broad phase only: first frame and steady state), cfg5 500 k cuboids f64 / 8 substeps, with the fixed synthetic
face manifolds of the bench (narrow phase out of path) -- and, since round 4, the same configurations STEPPED in the device closed loop
(real contacts): cfg1 after the boxes have landed, cfg4 as a simulated scene of 1 M bodies, cfg5 in f64, cfg2 with and without
avn_sleeping_enable.  usage: time_configs.py [out.json]"""
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


def closed_loop(lib, sc, bits=32, substeps=4, joints=None, sleeping=False):
    """the device closed loop (avn_pipeline_enable(1)): broad phase -> narrow phase -> bookkeeping -> solver, real contacts"""
    w = F.World(lib, F.default_config(bits, substeps=substeps))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    if joints is not None:
        w.distance_joints_upload(**joints)
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    if sleeping:
        w.sleeping_enable()
    return w


def time_closed(w, substeps, warmup, steps):
    for _ in range(warmup):
        w.step()
    w.synchronize()
    changes = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
        changes += w.pipeline_stats().last_status_changes
    w.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st, tm = w.pipeline_stats(), w.timers()
    return {"window": f"steps {warmup}..{warmup + steps - 1} of the closed loop", "ms_per_step": round(dt * 1e3, 4), "substeps_per_s": round(substeps / dt, 1),
            "manifolds_at_end": int(st.manifolds), "active_pairs": int(st.active_pairs), "status_changes_per_step": round(changes / steps, 1),
            "overflow_manifolds_at_end": int(st.last_overflow_manifolds), "kernel_launches_per_step": int(tm.kernel_launches), "host_bookkeeping_ms": round(st.last_host_ms, 4)}


def main():
    lib = avian_amd.load_library()
    out = {}
    # ---- closed-loop legs (round 4): the configurations stepped with real contacts, each a parity case of tests/test_gpu_configs_stepped.py /
    #      tests/test_gpu_closed_loop_configs.py ----
    sc = scenes.falling_grid(10, 1.5, 2.0)
    w = closed_loop(lib, sc, substeps=1)
    out["cfg1_closed_loop_after_landing"] = dict(bodies=sc.n, note="1 000 cuboids, 1 substep: timed AFTER the layers have landed and piled up (steps 250..299)", **time_closed(w, 1, 250, 50))
    w.close()
    sc = scenes.sparse_mixed(1_000_000)
    w = closed_loop(lib, sc)
    r = time_closed(w, 4, 5, 20)
    out["cfg4_1M_mixed_bodies_stepped_closed_loop"] = dict(bodies=sc.n, note="1 M ball / cuboid bodies simulated: broad phase + narrow phase + bookkeeping + 4 substeps", bodies_substeps_per_s=round(sc.n * r["substeps_per_s"]), **r)
    w.close()
    sc = scenes.box_stack(100, 50, 100)
    w = closed_loop(lib, sc, bits=64, substeps=8)
    out["cfg5_500k_f64_closed_loop"] = dict(bodies=sc.n, **time_closed(w, 8, 3, 10))
    w.close()
    sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100)
    joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
    w = closed_loop(lib, sc, joints=joints)
    out["cfg3_closed_loop_steps_60_79"] = dict(bodies=sc.n, joints=len(joints["body1"]), note="50 k cuboids + 100 chains of 100 links (9 900 distance joints) in the device closed loop: the stack collapses, the chains swing",
                                               **time_closed(w, 4, 60, 20))
    w.close()
    sc = scenes.box_stack(50, 40, 50)
    for name, slp in (("cfg2_closed_loop_steps_20_39", False), ("cfg2_closed_loop_steps_20_39_sleeping_enabled", True)):
        w = closed_loop(lib, sc, sleeping=slp)
        r = time_closed(w, 4, 20, 20)
        if slp:
            st = w.sleeping_stats()
            r.update(awake_bodies=int(st.n_awake_bodies), islands=int(st.islands.n_islands), island_manager_host_ms=round(st.last_host_ms, 4))
        out[name] = dict(bodies=sc.n, **r)
        w.close()
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
