import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import avian_amd
from avian_amd import _ffi as F, scenes
lib = avian_amd.load_library()
sc = scenes.box_stack(50, 40, 50)
w = F.World(lib, F.default_config(32, substeps=4))
w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
w.pipeline_enable()
ms = []
for s in range(60):
    w.step()
    if s >= 30: ms.append(w.diagnostics().narrow_phase_ms)
print(os.environ.get("AVN_LIB_PATH", "default")[-14:], "narrow_phase_ms (incl. bookkeeping) mean", round(float(np.mean(ms)), 4))
