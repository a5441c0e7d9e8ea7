#!/usr/bin/env python3
"""colour-list lengths of the cfg2 closed loop at a few steps (what the solver's launches are sized by)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avian_amd
from avian_amd import _ffi as F, scenes
lib = avian_amd.load_library()
sc = scenes.box_stack(50, 40, 50)
w = F.World(lib, F.default_config(32, substeps=4))
w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
w.pipeline_enable()
for s in range(121):
    w.step()
    if s in (10, 30, 110, 120):
        off, _ = w.pipeline_handles()
        print(s, np.diff(np.asarray(off)).tolist())
