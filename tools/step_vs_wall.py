import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, avian_amd
from avian_amd import _ffi as F
lib = avian_amd.load_library()
sc, substeps, _ = bench.build_inputs(lib, "cfg2_box_stack_100k")
for graph in (1, 0):
    w = F.World(lib, F.default_config(32, substeps=substeps, use_graph=graph))
    bench.setup_world(w, lib, sc)
    for _ in range(5): w.step()
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): w.step()
    w.synchronize()
    wall = (time.perf_counter() - t0) / 40 * 1e3
    s = [w.timers() for _ in range(1)]
    xs = []
    for _ in range(7):
        w.step(); xs.append(w.timers())
    med = lambda f: float(np.median([getattr(x, f) for x in xs]))
    print("graph", graph, "wall/step %.4f" % wall, {f: round(med(f), 4) for f in ("broad_phase_ms", "prepare_ms", "substeps_ms", "finalize_ms", "step_ms")})
