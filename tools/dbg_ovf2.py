import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
sc = scenes.box_stack(9, 8, 9)
order = [int(a) for a in sys.argv[1:]] or [1, 0]
call_handles = os.environ.get("CALL_HANDLES", "1") == "1"
for use_graph in order:
    ws = []
    for lib in (oracle_lib(), hip_lib()):
        cfg = F.default_config(32, substeps=4); cfg.use_graph = use_graph
        w = F.World(lib, cfg)
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        ws.append(w)
    for s in range(10):
        for w in ws: w.step()
        mode = os.environ.get("CALL_HANDLES", "1")
        if mode == "1":
            oo, oh = ws[0].pipeline_handles(), ws[1].pipeline_handles()
            assert np.array_equal(oo[0], oh[0]) and np.array_equal(oo[1], oh[1])
        elif mode == "hip":
            ws[1].pipeline_handles()
        elif mode == "oracle":
            ws[0].pipeline_handles()
        elif mode == "sync":
            ws[1].synchronize()
        elif mode == "stats":
            ws[1].pipeline_stats()
        bo, bh = ws[0].bodies_download(), ws[1].bodies_download()
        bad = [k for k in bo if not np.array_equal(bo[k], bh[k])]
        st = ws[1].pipeline_stats()
        dv = np.abs(bo["linear_velocity"] - bh["linear_velocity"]).max(1)
        print("graph", use_graph, "step", s, "overflow", st.last_overflow_manifolds, "manifolds", st.manifolds, "removed", st.pairs_removed, "added", st.pairs_added,
              "OK" if not bad else "DIFF n=%d first bodies %s maxdv %.3g" % (int((dv > 0).sum()), np.flatnonzero(dv > 0)[:8], dv.max()), flush=True)
        if bad: break
