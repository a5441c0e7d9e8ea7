import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import avian_amd
from avian_amd import _ffi as F, scenes
sc = scenes.many_pyramids(10, 10, 10)
w = F.World(avian_amd.load_library(), F.default_config(32, substeps=4))
w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
w.pipeline_enable()
for _ in range(30): w.step()
w.synchronize(); t0 = time.perf_counter()
for _ in range(300): w.step()
w.synchronize(); print("ms/step", round((time.perf_counter() - t0) / 300 * 1e3, 4), "host_ms", round(w.pipeline_stats().last_host_ms, 4))
