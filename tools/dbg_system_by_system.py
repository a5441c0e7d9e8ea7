import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from helpers import F, hip_lib, oracle_lib
from avian_amd import scenes
from test_gpu_configs import setup
sc = scenes.box_stack(12, 6, 12)
for subs in (1, 4, 8):
    wo = F.World(oracle_lib(), F.default_config(64, substeps=subs)); wh = F.World(hip_lib(), F.default_config(64, substeps=subs))
    setup(wo, oracle_lib(), sc); setup(wh, hip_lib(), sc)
    order = ["PREPARE_SOLVER_BODIES", "PREPARE_JOINTS", "PREPARE_CONTACT_CONSTRAINTS", "PRE_PROCESS_VELOCITY_INCREMENTS"]
    order += ["INTEGRATE_VELOCITIES", "WARM_START", "SOLVE_CONTACTS_BIAS", "INTEGRATE_POSITIONS", "SOLVE_CONTACTS_RELAX", "XPBD_SOLVE", "XPBD_VELOCITY_PROJECTION", "JOINT_DAMPING"] * subs
    order += ["CLEAR_VELOCITY_INCREMENTS", "SOLVE_RESTITUTION", "WRITEBACK_SOLVER_BODIES", "STORE_CONTACT_IMPULSES"]
    done = False
    for k, name in enumerate(order):
        wo.run_system(name); wh.run_system(name)
        for what, (a, b) in (("sb", (wo.solver_bodies_download(), wh.solver_bodies_download())), ("con", (wo.constraints_download(), wh.constraints_download())), ("bodies", (wo.bodies_download(), wh.bodies_download()))):
            for f in a:
                bad = ~((a[f] == b[f]) | (np.isnan(a[f]) & np.isnan(b[f])))
                if bad.any():
                    idx = tuple(np.argwhere(bad)[0])
                    print(f"substeps={subs} after[{k}] {name}: {what}.{f}: {bad.sum()} differ; first {idx}: {a[f][idx]!r} vs {b[f][idx]!r}")
                    done = True
        if done: break
    if not done: print(f"substeps={subs}: all systems identical")
