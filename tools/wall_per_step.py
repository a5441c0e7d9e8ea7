import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, avian_amd
from avian_amd import _ffi as F
lib = avian_amd.load_library()
sc, substeps, _ = bench.build_inputs(lib, "cfg2_box_stack_100k")
w = F.World(lib, F.default_config(32, substeps=substeps, use_graph=1))
bench.setup_world(w, lib, sc)
for _ in range(5): w.step()
w.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(60): w.step()
    w.synchronize()
    print("wall/step %.4f ms" % ((time.perf_counter() - t0) / 60 * 1e3))
