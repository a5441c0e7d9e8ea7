#!/usr/bin/env python3
"""Where the PCIe-inclusive step of bench.py goes: every call of the host-manifold flow timed on its own (pinned host arrays).
usage: python tools/time_pcie.py [steps]"""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import avian_amd
from avian_amd import _ffi as F, scenes
sys.path.insert(0, R)
from bench import setup_world


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cl_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # > 0: the manifold set of the device closed loop after that many steps (with its warm-start impulses) instead of the frozen synthetic one
    lib = avian_amd.load_library()
    warm = (None, None)
    if cl_steps:
        from avian_amd import level2_bench
        sc, mf, offs, warm = level2_bench.closed_loop_island(lib, F, scenes, 32, (50, 40, 50), cl_steps)
        w = F.World(lib, F.default_config(32, substeps=4))
        fr_a, re_a = mf.pop("friction"), mf.pop("restitution")
        meta = {"manifolds": mf, "offsets": offs, "n_manifolds": len(mf["body1"])}
    else:
        sc = scenes.box_stack(50, 40, 50)
        w = F.World(lib, F.default_config(32, substeps=4))
        meta = setup_world(w, lib, sc)
    keep = []
    ints = {"body1": np.int32, "body2": np.int32, "point_count": np.uint8, "manifold_flags": np.uint8, "rb_type": np.uint8}
    def pinned(k, a):
        if a is None or not hasattr(a, "nbytes") or a.nbytes == 0:
            return a
        a = np.ascontiguousarray(a)
        a = a.astype(ints[k]) if k in ints else (a.astype(np.float32) if a.dtype.kind == "f" else a)
        t = torch.from_numpy(a).pin_memory(); keep.append(t); return t.numpy()
    bk = {k: pinned(k, v) for k, v in sc.body_kwargs().items()}
    mfp = {k: pinned(k, v) for k, v in meta["manifolds"].items()}
    fr = pinned("f", fr_a if cl_steps else np.full(meta["n_manifolds"], sc.friction, np.float32)); re_ = pinned("f", re_a if cl_steps else np.full(meta["n_manifolds"], sc.restitution, np.float32))
    wn, wt = (pinned("f", warm[0]), pinned("f", warm[1])) if cl_steps else (None, None)
    bout = {k: pinned("f", np.zeros(sh, np.float32)) for k, sh in (("position", (sc.n, 3)), ("rotation", (sc.n, 4)), ("linear_velocity", (sc.n, 3)), ("angular_velocity", (sc.n, 3)))}
    M = meta["n_manifolds"]
    iout = {k: pinned("f", np.zeros(sh, np.float32)) for k, sh in (("warm_start_normal_impulse", (M, 4)), ("warm_start_tangent_impulse", (M, 4, 2)), ("normal_impulse", (M, 4)))}
    acc = {}
    def timed(name, f):
        t0 = time.perf_counter(); r = f(); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
    for it in range(steps + 1):
        if it == 1:
            acc.clear()
        timed("bodies_upload", lambda: w.bodies_upload(**bk))
        timed("manifolds_upload", lambda: scenes.upload_manifolds(w, mfp, meta["offsets"], fr, re_, wn, wt))
        timed("step+sync", lambda: (w.step(), w.synchronize()))
        timed("bodies_download", lambda: w.bodies_download(out=bout))
        timed("impulses_download", lambda: w.impulses_download(out=iout))
    out = {k: round(float(np.median(v)) * 1e3, 3) for k, v in acc.items()}   # medians: one slow DMA setup does not move them
    out["total_ms"] = round(sum(out.values()), 3)
    out["mean_ms"] = {k: round(float(np.mean(v)) * 1e3, 3) for k, v in acc.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
