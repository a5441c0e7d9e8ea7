#!/usr/bin/env python3
"""per-step device spans of the first 60 steps of a fresh cfg2 world (AVN_HOST_TRACE=3): is the bench's default warm-up long enough?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AVN_HOST_TRACE"] = "3"
import bench, avian_amd
from avian_amd import _ffi as F
lib = avian_amd.load_library()
sc, substeps, _ = bench.build_inputs(lib, "cfg2_box_stack_100k")
w = F.World(lib, F.default_config(32, substeps=substeps, use_graph=1))
bench.setup_world(w, lib, sc)
for _ in range(60): w.step()
w.synchronize()
