#!/usr/bin/env python3
"""cfg2 (100 000 boxes) in the device closed loop with and without avn_sleeping_enable: ms per step, window by window, next to the island manager's host
time.  With the `make measure` build and AVN_SLP_TRACE=1 the library also prints the host phases of every step on stderr.
usage: python tools/time_sleeping_cfg2.py [steps=140] [nx ny nz]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def run(lib, sc, steps, sleeping):
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    if sleeping:
        w.sleeping_enable()
    ms, host, awake, chg = [], [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        w.step(); w.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        chg.append(w.pipeline_stats().last_status_changes)
        if sleeping:
            st = w.sleeping_stats(); host.append(st.last_host_ms); awake.append(st.n_awake_bodies)
    w.close()
    return np.array(ms), np.array(host), np.array(awake), np.array(chg)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 140
    dims = tuple(int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (50, 40, 50)
    lib = avian_amd.load_library()
    sc = scenes.box_stack(*dims)
    res = {}
    for slp in (False, True):
        ms, host, awake, chg = run(lib, sc, steps, slp)
        res[slp] = ms
        for a in range(0, steps, 20):
            b = min(a + 20, steps)
            extra = f", island manager host {host[a:b].mean():.3f} ms, awake {awake[a:b].mean():.0f}" if slp else ""
            print(f"sleeping={int(slp)} steps {a}..{b - 1}: {ms[a:b].mean():.3f} ms/step (median {np.median(ms[a:b]):.3f}, min {ms[a:b].min():.3f}), changes/step {chg[a:b].mean():.0f}{extra}", flush=True)
        print(f"sleeping={int(slp)} per step, last 40: " + " ".join(f"{x:.2f}" for x in ms[-40:]), flush=True)
    for a in range(0, steps, 20):
        b = min(a + 20, steps)
        print(f"ratio steps {a}..{b - 1}: {res[True][a:b].mean() / res[False][a:b].mean():.3f}")


if __name__ == "__main__":
    main()
