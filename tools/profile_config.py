#!/usr/bin/env python3
"""One configuration of tools/time_configs.py under a profiler: `rocprofv3 --kernel-trace --stats -- python tools/profile_config.py cfg3|cfg5|cfg5f32 [steps]`."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import avian_amd
from avian_amd import _ffi as F, scenes
from time_configs import setup, time_steps


def main():
    which = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lib = avian_amd.load_library()
    if which == "cfg3":
        sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100)
        joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
        w = F.World(lib, F.default_config(32, substeps=4))
        setup(w, lib, sc, joints)
        print(time_steps(w, 4, steps=steps))
    elif which == "cfg5":
        sc = scenes.box_stack(100, 50, 100)
        w = F.World(lib, F.default_config(64, substeps=8))
        setup(w, lib, sc)
        print(time_steps(w, 8, warmup=2, steps=steps))
    elif which == "cfg5f32":   # cfg5's scene in f32: the f32 colour kernel at launches large enough to fill the chip (312 k manifolds per colour)
        sc = scenes.box_stack(100, 50, 100)
        w = F.World(lib, F.default_config(32, substeps=8))
        setup(w, lib, sc)
        print(time_steps(w, 8, warmup=2, steps=steps))
    else:
        raise SystemExit("cfg3 | cfg5 | cfg5f32")


if __name__ == "__main__":
    main()
