import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
nx, ny, nz = (int(a) for a in sys.argv[1:4])
use_graph = int(sys.argv[4])
sc = scenes.box_stack(nx, ny, nz)
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
    bo, bh = ws[0].bodies_download(), ws[1].bodies_download()
    bad = [k for k in bo if not np.array_equal(bo[k], bh[k])]
    st = ws[1].pipeline_stats(); tm = ws[1].timers()
    nd = int((np.abs(bo["linear_velocity"] - bh["linear_velocity"]).max(1) > 0).sum())
    print("step", s, "graph", use_graph, "ISL", os.environ.get("AVN_ISLAND_BLOCKS"), "island_blocks", tm.island_blocks, "overflow", st.last_overflow_manifolds, "manifolds", st.manifolds, "changes", st.last_status_changes,
          "OK" if not bad else "DIFF bodies differing: %d" % nd, flush=True)
    if bad: break
