#!/usr/bin/env python3
"""Closed-loop step (device broad phase -> device narrow phase -> host status processing -> solver) on a box stack:
per-system device times and the wall time of a whole step.  usage: python tools/time_pipeline.py [nx ny nz] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes
from avian_amd.pipeline import ContactPipeline


def main():
    nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 40, 50)
    steps = int(sys.argv[4]) if len(sys.argv) >= 5 else 20
    lib = avian_amd.load_library()
    sc = scenes.box_stack(nx, ny, nz)
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.collider_materials_upload(friction=0.5)
    pl = ContactPipeline(w, lib)
    t0 = time.perf_counter(); pl.step(); w.synchronize()
    print(f"first step (adds {pl.stats['pairs_added']} pairs, {pl.stats['pushes']} manifolds): {time.perf_counter() - t0:.2f} s")
    for _ in range(3):
        pl.step()
    w.synchronize()
    t_np = t_host = t_all = 0.0
    n_changes = 0
    for _ in range(steps):
        a = time.perf_counter()
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        pl.add_new_pairs(w.pairs_get())
        b = time.perf_counter()
        w.run_system("NARROW_PHASE")
        c = time.perf_counter()
        n_changes += pl.process_status_changes()
        d = time.perf_counter()
        w.run_system("SOLVER")
        e = time.perf_counter()
        t_np += c - b; t_host += d - c; t_all += e - a
    ms, _ = w.profile_system("NARROW_PHASE", 5)
    tm = w.timers()
    print(f"bodies {sc.n - 1}, active pairs {len(pl.active)}, manifolds {w.n_manifolds}, status changes/step {n_changes / steps:.1f}")
    print(f"NARROW_PHASE (kernel + count read-back): {ms / 5:.3f} ms; wall per step: narrow {t_np / steps * 1e3:.3f} ms, host status {t_host / steps * 1e3:.3f} ms, "
          f"whole closed-loop step {t_all / steps * 1e3:.3f} ms (solver device {tm.prepare_ms + tm.substeps_ms + tm.finalize_ms:.3f} ms)")
    # the same scene through the library's own closed loop (avn_pipeline_enable: C++ bookkeeping, no Python in the step)
    w2 = F.World(lib, F.default_config(32, substeps=4))
    w2.bodies_upload(**sc.body_kwargs()); w2.colliders_upload(**sc.collider_kwargs())
    w2.existing_pairs_upload(np.zeros(0, np.uint64)); w2.collider_materials_upload(friction=0.5)
    w2.pipeline_enable()
    t0 = time.perf_counter(); w2.step(); w2.synchronize(); first = time.perf_counter() - t0
    for _ in range(3):
        w2.step()
    w2.synchronize()
    t0 = time.perf_counter()
    host = 0.0; ch = 0; ovf = 0
    for _ in range(steps):
        w2.step()
        st = w2.pipeline_stats(); host += st.last_host_ms; ch += st.last_status_changes; ovf += st.last_overflow_manifolds
    w2.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm2 = w2.timers()
    print(f"library pipeline: first step {first:.2f} s; {dt * 1e3:.3f} ms/step = {4 / dt:.1f} substeps/s closed loop; host bookkeeping {host / steps:.3f} ms/step, "
          f"status changes/step {ch / steps:.0f}, overflow manifolds {ovf / steps:.0f}, substeps device {tm2.substeps_ms:.3f} ms, manifolds {st.manifolds}")
    b = w.bodies_download()
    print("max |v|", float(np.abs(b["linear_velocity"]).max()), "max drift", float(np.abs(b["position"][1:] - sc.position[1:]).max()))


if __name__ == "__main__":
    main()
