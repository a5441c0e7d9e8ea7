#!/usr/bin/env python3
"""Closed loop of the library (avn_pipeline_enable: device broad phase -> device narrow phase -> device ContactGraph / ConstraintGraph
bookkeeping -> solver) on a box stack: wall time and pipeline counters of every step.
usage: python tools/time_closed_loop.py [nx ny nz] [steps] [host]      ("host" = the host-side bookkeeping of round 1, for A/B)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def main():
    args = [a for a in sys.argv[1:] if a not in ("host", "nosync")]
    host = "host" in sys.argv[1:]
    nosync = "nosync" in sys.argv[1:]   # no avn_synchronize between steps (what bench.py's closed-loop windows do): wall time per WINDOW of 20 steps
    nx, ny, nz = (int(a) for a in args[0:3]) if len(args) >= 3 else (50, 40, 50)
    steps = int(args[3]) if len(args) >= 4 else 24
    out = args[4] if len(args) >= 5 else None
    if os.environ.get("AVN_TOOL_TORCH"):   # diagnosis: the bench process has torch's HIP context next to the library's
        import torch
        torch.cuda.set_device(0); torch.cuda.synchronize()
        if os.environ["AVN_TOOL_TORCH"] == "2": _keep = torch.zeros(1 << 20, device="cuda")
    lib = avian_amd.load_library()
    sc = scenes.box_stack(nx, ny, nz)
    if os.environ.get("AVN_TOOL_EXTRA_WORLD"):   # diagnosis: another world of the same size alive (and stepped) in the process
        w2 = F.World(lib, F.default_config(32, substeps=4))
        w2.bodies_upload(**sc.body_kwargs()); w2.colliders_upload(**sc.collider_kwargs())
        w2.existing_pairs_upload(np.zeros(0, np.uint64)); w2.collider_materials_upload(friction=0.5)
        w2.pipeline_enable()
        for _ in range(30): w2.step()
        w2.synchronize()
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable(host_bookkeeping=host)
    rows = []
    if nosync:
        w.synchronize(); t0 = time.perf_counter(); marks = {}
        for s in range(steps):
            w.step()
            if s in (3, 23, 43, 99, 119):
                w.synchronize(); marks[s] = time.perf_counter()
        for a, b in ((3, 23), (23, 43), (99, 119)):
            if a in marks and b in marks: print(f"nosync window steps {a + 1}..{b}: {(marks[b] - marks[a]) / (b - a) * 1e3:.3f} ms/step")
        return
    for s in range(steps):
        t0 = time.perf_counter(); w.step(); w.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        st = w.pipeline_stats(); tm = w.timers()
        rows.append(dict(step=s, wall_ms=round(dt, 3), changes=st.last_status_changes, manifolds=st.manifolds, overflow=st.last_overflow_manifolds,
                         host_ms=round(st.last_host_ms, 3), broad_ms=round(tm.broad_phase_ms, 3), prepare_ms=round(tm.prepare_ms, 3), substeps_ms=round(tm.substeps_ms, 3),
                         finalize_ms=round(tm.finalize_ms, 3), step_ms=round(tm.step_ms, 3), launches=tm.kernel_launches, active_pairs=st.active_pairs, new_pairs=tm.pair_count))
        print(rows[-1], flush=True)
    tail = rows[4:]
    if tail:
        print("mean of steps 4..%d: wall %.3f ms, device step %.3f ms, host bookkeeping %.3f ms" % (steps - 1, np.mean([r["wall_ms"] for r in tail]),
              np.mean([r["step_ms"] for r in tail]), np.mean([r["host_ms"] for r in tail])))
    if out:
        json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
