"""Stress of the overflow colour's dataflow pass: the same dense stack N times, HIP closed loop vs the oracle (computed once), count of runs whose
bodies differ at any step.  usage: python tools/stress_ovf.py nx ny nz repeats steps"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib
nx, ny, nz, reps, steps = (int(a) for a in sys.argv[1:6])
sc = scenes.box_stack(nx, ny, nz)
def mk(lib):
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    return w
wo = mk(oracle_lib())
ref = []
for s in range(steps):
    wo.step(); ref.append(wo.bodies_download())
bad = 0; first = []
for r in range(reps):
    w = mk(hip_lib())
    for s in range(steps):
        w.step()
        b = w.bodies_download()
        if any(not np.array_equal(b[k], ref[s][k]) for k in b):
            bad += 1; first.append(s); break
    del w
print("mode", os.environ.get("AVN_OVF_MODE", "0"), "runs", reps, "bad", bad, "first bad steps", first[:10], flush=True)
