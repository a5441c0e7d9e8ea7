#!/usr/bin/env python3
"""How far is k_color_pass<f32, SOLVE_BIAS> from what its memory accesses alone cost?  Two processes on the cfg2 world, isolated passes
(HIP events on the world's stream): the real kernel, and its memory skeleton (AVN_BIAS_SKELETON=1: the same loads, gathers and stores,
values written back unchanged, no solve).  Prints one JSON object.

usage: python tools/measure_floor.py [--scene cfg2_box_stack_100k] [--reps 40]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(scene, reps):
    import bench
    import avian_amd
    from avian_amd import _ffi as F
    lib = avian_amd.load_library()
    sc, substeps, _ = bench.build_inputs(lib, scene)
    w = F.World(lib, F.default_config(32, substeps=substeps, use_graph=0))
    meta = bench.setup_world(w, lib, sc)
    for _ in range(3):
        w.step()
    w.profile_system("SOLVE_CONTACTS_BIAS", 3)
    best = None
    for _ in range(5):
        ms, launches = w.profile_system("SOLVE_CONTACTS_BIAS", reps)
        us = ms * 1e3 / max(launches, 1)
        best = us if best is None else min(best, us)
    algo = 248 * meta["n_manifolds"] + 88 * meta["points"]
    print(json.dumps({"avg_launch_us": round(best, 3), "launches_per_pass": launches // reps, "algorithmic_bytes_per_pass": algo}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    scene = "cfg2_box_stack_100k"; reps = 40
    a = sys.argv[1:]
    if "--scene" in a:
        scene = a[a.index("--scene") + 1]
    if "--reps" in a:
        reps = int(a[a.index("--reps") + 1])
    out = {}
    # AVN_BIAS_SKELETON only exists in the measurement build of the library (make -C avian_amd/csrc measure: -DAVN_MEASURE)
    mlib = os.path.join(REPO, "avian_amd", "csrc", "measure", "libavian_mi355x.so")
    if not os.path.exists(mlib):
        subprocess.run(["make", "-C", os.path.join(REPO, "avian_amd", "csrc"), "-j8", "measure"], check=True)
    os.environ["AVN_LIB_PATH"] = mlib
    for name, env in (("solve", {}), ("memory_skeleton", {"AVN_BIAS_SKELETON": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scene, str(reps)], env=dict(os.environ, **env), capture_output=True, text=True, cwd=REPO)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[name] = json.loads(line[-1]) if line else {"error": r.stderr[-800:]}
    if "avg_launch_us" in out["solve"] and "avg_launch_us" in out["memory_skeleton"]:
        s, k = out["solve"], out["memory_skeleton"]
        per_launch = s["algorithmic_bytes_per_pass"] / s["launches_per_pass"]
        out["summary"] = {"scene": scene, "solve_GBps": round(per_launch / s["avg_launch_us"] / 1e3, 1), "skeleton_GBps": round(per_launch / k["avg_launch_us"] / 1e3, 1),
                          "solve_over_skeleton": round(s["avg_launch_us"] / k["avg_launch_us"], 3),
                          "reading": "the skeleton is the launch with the solve's arithmetic removed: what remains above it is the dependent impulse chain"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
