#!/usr/bin/env python3
"""The reference's own 3D bench scenes, closed loop (benches/src/dim3/large_pyramid.rs, many_pyramids.rs; protocol of
benches/src/cli.rs:358-405: one un-timed step, then the mean step time): the HIP library (device narrow phase,
avn_pipeline_enable) next to the single-thread CPU oracle on the same inputs, and a bit-for-bit comparison of the two.

usage: python tools/bench_reference_scenes.py [steps] [substeps] [out.json]   (default 300 steps, 4 substeps = BASELINE.md rows ref-A / ref-B; dt 1/60)
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import avian_amd
from avian_amd import _ffi as F, scenes
from helpers import oracle_lib   # the oracle is the CPU baseline and the checker here


def run(lib, sc, steps, sync, substeps, oracle_threads=None):
    if oracle_threads:
        os.environ["AVO_THREADS"] = str(oracle_threads)   # read at world creation (oracle/avo_parallel.hpp)
    try:
        w = F.World(lib, F.default_config(32, substeps=substeps))
    finally:
        os.environ.pop("AVO_THREADS", None)
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    w.step()
    if sync: w.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    if sync: w.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return w, dt


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    substeps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    results = []
    for name, sc in (("Large Pyramid 3D (base 100)", scenes.large_pyramid(100)), ("Many Pyramids 3D (10 x 10 x base 10)", scenes.many_pyramids(10, 10, 10))):
        wh, th = run(avian_amd.load_library(), sc, steps, True, substeps)
        wo, to = run(oracle_lib(), sc, steps, False, substeps)
        threads = min(64, os.cpu_count() or 1)
        wm, tm_ = run(oracle_lib(), sc, steps, False, substeps, oracle_threads=threads)
        bm = wm.bodies_download()
        bh, bo = wh.bodies_download(), wo.bodies_download()
        same = all(np.array_equal(bh[k], bo[k]) and np.array_equal(bm[k], bo[k]) for k in bh)
        st = wh.pipeline_stats()
        n_dyn = int((sc.rb_type == F.RB_DYNAMIC).sum())
        tm = wh.timers()
        results.append({"scene": name, "dynamic_bodies": n_dyn, "substeps": substeps, "steps": steps, "mi355x_ms_per_step": round(th * 1e3, 4),
                        "mi355x_substeps_per_s": round(substeps / th, 1), "cpu_oracle_1_thread_ms_per_step": round(to * 1e3, 3),
                        "cpu_oracle_substeps_per_s": round(substeps / to, 2),
                        "cpu_oracle_threads": threads, "cpu_oracle_threaded_ms_per_step": round(tm_ * 1e3, 3), "cpu_oracle_threaded_substeps_per_s": round(substeps / tm_, 2),
                        "cpu_oracle_best_ms_per_step": round(min(to, tm_) * 1e3, 3), "cpu_oracle_best_is": "1 thread" if to <= tm_ else f"{threads} threads",
                        "mi355x_over_cpu_best": round(min(to, tm_) / th, 1), "bodies_bit_identical": bool(same), "manifolds": int(st.manifolds),
                        "active_pairs": int(st.active_pairs), "host_cores": os.cpu_count(),
                        "island_blocks": int(tm.island_blocks), "kernel_launches_per_step": int(tm.kernel_launches),
                        "last_step_ms": {"broad_phase": round(tm.broad_phase_ms, 4), "prepare": round(tm.prepare_ms, 4), "substeps": round(tm.substeps_ms, 4),
                                         "finalize": round(tm.finalize_ms, 4), "step": round(tm.step_ms, 4)},
                        "path": "closed loop: device broad phase + device narrow phase (Ball/Cuboid) + ContactGraph / ConstraintGraph bookkeeping on the device (k_graph.hip) + solver (avn_pipeline_enable(1))"})
        print(f"{name}: {n_dyn} boxes, {substeps} substeps, {steps} steps | MI355X {th * 1e3:.3f} ms/step ({substeps / th:.0f} substeps/s) | "
              f"CPU oracle 1 thread {to * 1e3:.2f} ms/step ({substeps / to:.1f} substeps/s), {threads} threads {tm_ * 1e3:.2f} ms/step | x{min(to, tm_) / th:.0f} | bodies bit-identical after {steps + 1} steps: {same} | "
              f"manifolds {st.manifolds}, active pairs {st.active_pairs}, status changes last step {st.last_status_changes}, max |v| {float(np.abs(bh['linear_velocity']).max()):.3f} | "
              f"island blocks {tm.island_blocks}, launches {tm.kernel_launches}, bp {tm.broad_phase_ms:.3f} prep {tm.prepare_ms:.3f} sub {tm.substeps_ms:.3f} fin {tm.finalize_ms:.3f} ms")
    if out_path:
        import json
        json.dump(results, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
