#!/usr/bin/env python3
"""How deep are the dependency chains INSIDE a suffix of colours?  cfg2's closed loop is stepped, the ContactId -> bodies map is kept from the steps' new pairs,
and for every c0 the manifolds of colours c0..22 are levelled (level = 1 + the highest level of an earlier manifold of the suffix on one of its bodies).
usage: python tools/tail_depth.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 110
    lib = avian_amd.load_library()
    sc = scenes.box_stack(50, 40, 50)
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    cap = 4_000_000
    b1 = np.full(cap, -1, np.int64); b2 = np.full(cap, -1, np.int64)
    static = np.asarray(sc.rb_type) == F.RB_STATIC
    per_step = "per-step" in sys.argv   # the overflow colour's dataflow depth after EVERY step (to set against the per-launch durations of a kernel trace)
    for s in range(steps):
        w.step(); w.synchronize()
        p = w.pairs_get()
        ids = w.pipeline_new_pair_ids()
        if len(ids):
            b1[ids] = p["body1"][:len(ids)]; b2[ids] = p["body2"][:len(ids)]
        if per_step:
            off, h = w.pipeline_handles()
            last = np.zeros(sc.n, np.int32); depth = 0
            for cid in h[off[23]:off[24]]:
                x, y = int(b1[cid]), int(b2[cid])
                lev = 1 + max(0 if static[x] else last[x], 0 if static[y] else last[y])
                if not static[x]: last[x] = lev
                if not static[y]: last[y] = lev
                depth = max(depth, lev)
            print(f"after step {s}: overflow colour {int(off[24] - off[23])} manifolds, depth {depth}", flush=True)
    if per_step:
        return
    off, h = w.pipeline_handles()
    print("colour sizes:", [int(off[c + 1] - off[c]) for c in range(24)])
    # the overflow colour (index 23) is solved serially in list order: its dataflow depth
    last = np.zeros(sc.n, np.int32); hist = {}
    for cid in h[off[23]:off[24]]:
        x, y = int(b1[cid]), int(b2[cid])
        lev = 1 + max(0 if static[x] else last[x], 0 if static[y] else last[y])
        if not static[x]: last[x] = lev
        if not static[y]: last[y] = lev
        hist[lev] = hist.get(lev, 0) + 1
    print("overflow colour:", int(off[24] - off[23]), "manifolds, levels", dict(sorted(hist.items())))
    # how much of the overflow colour's dependency structure stays inside a wave of the dataflow pass (64 consecutive list entries)?
    prev = {}; n_pred = n_same = 0
    cost_now = np.zeros(int(off[24] - off[23])); cost_fwd = np.zeros_like(cost_now)   # critical path: every hop 1.0 | a hop inside a wave 0.5 (no memory round trip)
    for i, cid in enumerate(h[off[23]:off[24]]):
        a = b = 0.0; has = same = False
        for body in (int(b1[cid]), int(b2[cid])):
            if static[body]: continue
            j = prev.get(body)
            if j is not None:
                has = True; intra = (j // 64) == (i // 64); same |= intra
                a = max(a, cost_now[j]); b = max(b, cost_fwd[j] - (0.5 if intra else 0.0))
            prev[body] = i
        cost_now[i] = a + 1.0; cost_fwd[i] = b + 1.0
        n_pred += has; n_same += same
    if len(cost_now):
        print(f"overflow colour: {n_pred} of {len(cost_now)} manifolds have a predecessor, {n_same} of them inside their own wave; critical path {cost_now.max():.1f} hops, "
              f"{cost_fwd.max():.1f} if a hop inside a wave costs half")
    for c0 in (4, 5, 6, 7, 8, 9, 10, 12, 14):
        last = np.zeros(sc.n, np.int32)
        hist = {}
        total = 0
        for c in range(c0, 23):
            ids = h[off[c]:off[c + 1]]
            x, y = b1[ids], b2[ids]
            assert (x >= 0).all() and (y >= 0).all()
            lx = np.where(static[x], 0, last[x]); ly = np.where(static[y], 0, last[y])
            lev = 1 + np.maximum(lx, ly)
            last[x[~static[x]]] = lev[~static[x]]; last[y[~static[y]]] = lev[~static[y]]
            for v, n in zip(*np.unique(lev, return_counts=True)): hist[int(v)] = hist.get(int(v), 0) + int(n)
            total += len(ids)
        print(f"c0 = {c0:2d}: {total:6d} manifolds in {23 - c0} colours, levels {dict(sorted(hist.items()))}")


if __name__ == "__main__":
    main()
