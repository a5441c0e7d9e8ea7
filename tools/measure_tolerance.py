#!/usr/bin/env python3
"""The f32 tolerance of "the same inputs through the reference" (SURVEY.md §8 N1), MEASURED on the CPU restatement.

Avian's default build is only reproducible up to its platform: sin/cos come from the platform libm (crates/avian3d/Cargo.toml:38-44:
`enhanced-determinism` -> libm is opt-in), glam's f32 quaternion product sums pairwise on SSE2 and left to right in its scalar
implementation, and a compiler may or may not contract a*b+c.  This script runs the closed loop (broad phase -> narrow phase -> solver)
of one scene through four variants of the oracle -- baseline (deterministic polynomial sin/cos, SSE2 association, no contraction: what the
HIP kernels match bit for bit), host-libm trig, scalar quaternion product, FMA-contracted build -- and reports max |dx|, |dv|, |dq| of
each variant against the baseline after the requested step counts.

usage: python tools/measure_tolerance.py scene [steps ...] [--json out.json]      scene: large_pyramid[:base] | stack:nx,ny,nz"""
import ctypes as C
import json
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from avian_amd import _ffi as F, scenes  # noqa: E402
from helpers import oracle_lib  # noqa: E402

VARIANTS = ("libm_trig", "scalar_quat", "fma")


def fma_lib():
    oracle_lib()   # (builds both)
    return F.Library(os.path.join(R, "oracle", "liboracle_fma.so"), "avo_")


def build_scene(spec):
    if spec.startswith("large_pyramid"):
        return scenes.large_pyramid(int(spec.split(":")[1]) if ":" in spec else 100)
    nx, ny, nz = (int(a) for a in spec.split(":")[1].split(","))
    return scenes.box_stack(nx, ny, nz)


def run(lib, sc, checkpoints, substeps=4):
    w = F.World(lib, F.default_config(32, substeps=substeps))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=sc.friction, restitution=sc.restitution)
    w.pipeline_enable()
    out = {}
    for s in range(1, max(checkpoints) + 1):
        w.step()
        if s in checkpoints:
            out[s] = w.bodies_download()
    return out


def measure(spec, checkpoints):
    """{variant: {steps: (max|dx|, max|dv|, max|dw|, max|dq|)}} against the baseline oracle."""
    sc = build_scene(spec)
    base_lib = oracle_lib()
    toggles = {"libm_trig": base_lib.dll.avo_use_libm_trig, "scalar_quat": base_lib.dll.avo_use_scalar_quat}
    for f in toggles.values():
        f.argtypes = [C.c_int]; f.restype = None
    base = run(base_lib, sc, checkpoints)
    res = {}
    for v in VARIANTS:
        if v == "fma":
            got = run(fma_lib(), sc, checkpoints)
        else:
            toggles[v](1)
            try:
                got = run(base_lib, sc, checkpoints)
            finally:
                toggles[v](0)
        res[v] = {}
        for s in checkpoints:
            a, b = base[s], got[s]
            dq = np.minimum(np.abs(a["rotation"] - b["rotation"]).max(), np.abs(a["rotation"] + b["rotation"]).max())
            res[v][s] = (float(np.abs(a["position"] - b["position"]).max()), float(np.abs(a["linear_velocity"] - b["linear_velocity"]).max()),
                         float(np.abs(a["angular_velocity"] - b["angular_velocity"]).max()), float(dq))
    return res


def main():
    argv = list(sys.argv[1:])
    out = None
    if "--json" in argv:
        i = argv.index("--json"); out = argv[i + 1]; del argv[i:i + 2]
    args = argv
    spec = args[0]
    checkpoints = [int(a) for a in args[1:]] or [1, 10, 300]
    res = measure(spec, checkpoints)
    print(f"{spec}: max |dx| [m]  |dv| [m/s]  |dw| [rad/s]  |dq| against the baseline oracle")
    for v in VARIANTS:
        for s in checkpoints:
            print("  %-12s after %4d steps: %.3e  %.3e  %.3e  %.3e" % ((v, s) + res[v][s]))
    if out:
        json.dump({"scene": spec, "columns": ["max_dx_m", "max_dv_m_per_s", "max_dw_rad_per_s", "max_dq"],
                   "result": {v: {str(s): res[v][s] for s in checkpoints} for v in VARIANTS}}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
