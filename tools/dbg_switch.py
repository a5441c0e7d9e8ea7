import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
sc = scenes.box_stack(4, 3, 4)
ws = []
for lib in (oracle_lib(), hip_lib()):
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    ws.append((w, lib))
def cmp(tag):
    bo, bh = ws[0][0].bodies_download(), ws[1][0].bodies_download()
    bad = [k for k in bo if not np.array_equal(bo[k], bh[k])]
    print(tag, "OK" if not bad else ("DIFF " + str(bad)), flush=True)
for s in range(3):
    for w, _ in ws: w.step()
    cmp("phase1 step %d" % s)
for w, lib in ws:
    w.pipeline_enable(False)
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get()
    print("pairs", len(p))
    mf = scenes.axis_aligned_manifolds(sc, np.stack([p["body1"], p["body2"]], axis=1))
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, sc.restitution)
for s in range(3):
    for w, _ in ws: w.step()
    cmp("phase2 step %d" % s)
for w, _ in ws: w.pipeline_enable()
for s in range(2):
    for w, _ in ws: w.step()
    cmp("phase3 step %d" % s)
    ho, hh = ws[0][0].pipeline_handles(), ws[1][0].pipeline_handles()
    print(" handles equal:", np.array_equal(ho[0], hh[0]) and np.array_equal(ho[1], hh[1]), ho[0][-1], hh[0][-1])
