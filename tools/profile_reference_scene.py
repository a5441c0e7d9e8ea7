#!/usr/bin/env python3
"""120 closed-loop steps of one of the reference's bench scenes (many | large), for `rocprofv3 --kernel-trace --stats`
(tools/final_measure.sh): per-kernel time of the island-block path (Many Pyramids) and the colour-launch path (Large Pyramid)."""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import avian_amd
from avian_amd import _ffi as F, scenes
name = sys.argv[1]
sc = scenes.many_pyramids(10, 10, 10) if name == "many" else scenes.large_pyramid(100)
w = F.World(avian_amd.load_library(), F.default_config(32, substeps=4))
w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
w.pipeline_enable()
for _ in range(120): w.step()
w.synchronize()
tm = w.timers(); print(name, "island blocks", tm.island_blocks, "launches", tm.kernel_launches, "step_ms", tm.step_ms)
