#!/usr/bin/env python3
"""Debug aid: the spawn-with-renumbering scene of tests/test_gpu_pipeline_edges.py, row by row at the first diverging step."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import F, hip_lib, oracle_lib
from pipeline_scenes import dropped_boxes
from test_gpu_pipeline_edges import _respawn, upload

bodies, colliders = dropped_boxes(seed=11, n=40)
worlds = []
for lib in (oracle_lib(), hip_lib()):
    w = F.World(lib, F.default_config(32, substeps=4))
    upload(w, bodies, colliders); w.collider_materials_upload(friction=0.6, restitution=0.0); w.pipeline_enable()
    worlds.append(w)
wo, wh = worlds
def rows(w, n):
    out = {}
    for i in range(n):
        try:
            r = w.contacts_download(np.array([i], np.uint32))
            out[i] = (int(r["flags"][0]), int(r["point_count"][0]))
        except Exception as e:
            out[i] = None
    return out
for _ in range(25):
    wo.step(); wh.step()
state = [(bodies, colliders), (bodies, colliders)]
s = 25
for k, (pos, first) in enumerate((([1.5, 6.0, 1.5], False), ([2.5, 7.0, 2.0], True))):
    state = [_respawn(w, st[0], st[1], pos, [0.4, 0.3, 0.5], 5000 + k, first) for w, st in zip((wo, wh), state)]
    for _ in range(30):
        n = wh.pipeline_stats().pairs_added + 4
        before_o, before_h = rows(wo, n), rows(wh, n)
        wo.step(); wh.step(); wo.synchronize(); wh.synchronize()
        so, sh = wo.pipeline_stats(), wh.pipeline_stats()
        if (so.pairs_removed, so.pairs_added, so.manifolds) != (sh.pairs_removed, sh.pairs_added, sh.manifolds):
            print("step", s, "oracle", so.pairs_added, so.pairs_removed, so.manifolds, "device", sh.pairs_added, sh.pairs_removed, sh.manifolds)
            ao, ah = rows(wo, n), rows(wh, n)
            for i in range(n):
                if (ao[i] is None or ao[i][0] == 0) != (ah[i] is None or ah[i][0] == 0) or (before_o[i] is None) != (before_h[i] is None or before_h[i][0] == 0):
                    print(" id", i, "before o/h", before_o[i], before_h[i], "after o/h", ao[i], ah[i])
            po, ph = wo.pairs_get(), wh.pairs_get()
            print("new pairs o", po, "h", ph)
            mo, xo, eo = wo.aabbs_download(); mh, xh, eh = wh.aabbs_download()
            print("aabb equal", np.array_equal(mo, mh), np.array_equal(xo, xh), "order equal", np.array_equal(eo, eh))
            sys.exit(0)
        s += 1
print("no divergence")
