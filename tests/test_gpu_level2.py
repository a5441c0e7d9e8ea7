"""GPU: level-2 sharding on the HIP backend.  (1) ONE island split into x-slab worlds, stepped colour by colour through the C ABI with the
host transport (avn_run_color_pass / avn_halo_pack / avn_halo_unpack): bit-identical to the unsplit HIP world AND to the unsplit oracle.
(2) the library-issued exchange (avn_comm_init + avn_step: pack kernel -> grouped ncclSend / ncclRecv on the world's stream -> unpack kernel)
on the one GPU a test box has: a world that is its own peer (RCCL self send/recv) must reproduce the plain step bit for bit.  The multi-GPU
form of the same path is tools/level2_multi_gpu.py (torchrun, one rank per GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from level2_helpers import compare_with_single, global_problem, make_single, make_split, overflow_from, step_split_in_process

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bits,world_size,restitution", [(32, 2, 0.0), (32, 3, 0.3), (64, 2, 0.3)])
def test_split_island_equals_single_world_on_hip(bits, world_size, restitution):
    hip, orc = hip_lib(), oracle_lib()
    sc, pm, offs, _ = global_problem(orc, 8, 4, 5, seed=bits + world_size)
    single = make_single(hip, bits, sc, pm, offs, restitution, 3)
    ref = make_single(orc, bits, sc, pm, offs, restitution, 3)
    plan, worlds = make_split(hip, bits, sc, pm, offs, restitution, 3, world_size)
    assert sum(len(p.send_bodies) for p in plan) > 0
    for step in range(3):
        single.run_system("SOLVER")
        ref.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, restitution > 0)
        compare_with_single(single, plan, worlds)
        compare_with_single(ref, plan, worlds)


@pytest.mark.parametrize("bits,world_size,keep,restitution", [(32, 2, 6, 0.0), (32, 3, 0, 0.3), (64, 2, 3, 0.3), (32, 4, 10, 0.0)])
def test_overflow_colour_on_shared_bodies_on_hip(bits, world_size, keep, restitution):
    """Round 6: overflow-colour manifolds on shared bodies -- the global list cut into levels, one exchange slot per level (avn_halo_overflow_levels_upload).
    Split HIP worlds == the unsplit HIP world == the unsplit oracle, bit for bit."""
    hip, orc = hip_lib(), oracle_lib()
    sc, pm, offs, _ = global_problem(orc, 8, 4, 5, seed=bits + world_size + keep)
    offs = overflow_from(offs, keep)
    single = make_single(hip, bits, sc, pm, offs, restitution, 3)
    ref = make_single(orc, bits, sc, pm, offs, restitution, 3)
    plan, worlds = make_split(hip, bits, sc, pm, offs, restitution, 3, world_size)
    assert plan[0].n_overflow_levels > 1
    for step in range(3):
        single.run_system("SOLVER")
        ref.run_system("SOLVER")
        step_split_in_process(plan, worlds, 3, restitution > 0)
        compare_with_single(single, plan, worlds)
        compare_with_single(ref, plan, worlds)


@pytest.mark.parametrize("bits,world_size,damped,keep", [(32, 2, True, None), (32, 3, False, None), (64, 2, True, 6), (32, 4, True, None)])
def test_joints_on_shared_bodies_on_hip(bits, world_size, damped, keep):
    """Round 6: joints whose bodies are shared between slabs (the joint slot, avn_halo_joint_slot_set).  Split HIP worlds == the unsplit HIP world == the unsplit
    oracle: bodies, impulses and the joints' lagrange sums / forces."""
    from level2_helpers import compare_joints_with_single, make_joint_worlds, stack_joints, step_split_with_joints
    hip, orc = hip_lib(), oracle_lib()
    sc, pm, offs, _ = global_problem(orc, 8, 4, 5, seed=bits + world_size)
    if keep is not None:
        offs = overflow_from(offs, keep)
    jkw = stack_joints(sc, 8, 4, 5, seed=world_size, damped=damped)
    single, plan, worlds = make_joint_worlds(hip, bits, sc, pm, offs, 0.0, 3, world_size, jkw)
    ref = make_single(orc, bits, sc, pm, offs, 0.0, 3); ref.joints_upload(**jkw)
    assert plan[0].joint_slot
    for step in range(3):
        single.run_system("SOLVER"); ref.run_system("SOLVER")
        step_split_with_joints(plan, worlds, 3, False)
        compare_with_single(single, plan, worlds); compare_with_single(ref, plan, worlds)
        compare_joints_with_single(single, plan, worlds); compare_joints_with_single(ref, plan, worlds)
    assert float(np.abs(single.joints_download()["total_lagrange"]).max()) > 1e-4


def test_cfg5_shaped_closed_loop_manifolds_over_2_and_4_slabs_on_hip():
    """The cfg5-shaped case (50 000 cuboids f64, the HIP closed loop's own manifolds with ~10^5 in the overflow colour): HIP slabs == the unsplit HIP world == the
    unsplit oracle, every step, bodies and impulses."""
    from level2_helpers import cfg5_shaped_case
    hip, orc = hip_lib(), oracle_lib()
    cfg5_shaped_case(hip, [hip, orc], hip, (2, 4))


def test_overflow_colour_inside_a_slab_runs_in_order():
    """Manifolds of the overflow colour that touch no shared body stay legal: colour 23 runs first, on the host schedule."""
    hip, orc = hip_lib(), oracle_lib()
    sc, pm, offs, _ = global_problem(orc, 8, 2, 2, seed=3)
    # push the last colour's manifolds that are far from the slab cut into the overflow colour: order them last and move the offset
    x1 = sc.position[np.maximum(pm["body1"], pm["body2"]), 0]
    cut = np.median(sc.position[sc.rb_type == F.RB_DYNAMIC, 0])
    far = np.abs(x1 - cut) > 2.5
    last = int(np.flatnonzero(np.diff(offs[:24]) > 0)[-1])
    ids = np.arange(len(far))
    in_last = (ids >= offs[last]) & (ids < offs[last + 1])
    move = in_last & far
    if not move.any():
        pytest.skip("no movable manifold in this scene")
    keep_order = np.concatenate([ids[~move], ids[move]])
    pm2 = {k: (np.asarray(v)[keep_order] if hasattr(v, "__len__") and len(v) == len(ids) else v) for k, v in pm.items()}
    offs2 = offs.copy()
    offs2[last + 1:24] = offs[last + 1] - int(move.sum())
    offs2[24] = len(ids)
    single = make_single(hip, 32, sc, pm2, offs2, 0.0, 2)
    ref = make_single(orc, 32, sc, pm2, offs2, 0.0, 2)
    plan, worlds = make_split(hip, 32, sc, pm2, offs2, 0.0, 2, 2)
    for _ in range(2):
        single.run_system("SOLVER"); ref.run_system("SOLVER")
        step_split_in_process(plan, worlds, 2, False)
        compare_with_single(single, plan, worlds)
        compare_with_single(ref, plan, worlds)


SELF_EXCHANGE = r"""
import sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(repo)r)
import numpy as np
from helpers import F, hip_lib, oracle_lib
from level2_helpers import global_problem, make_single
hip, orc = hip_lib(), oracle_lib()
for bits in (32, 64):
    sc, pm, offs, _ = global_problem(orc, 6, 3, 4, seed=11)
    plain = make_single(hip, bits, sc, pm, offs, 0.3, 3)
    looped = make_single(hip, bits, sc, pm, offs, 0.3, 3)
    # its own peer: after every colour the bodies that colour touched travel rank 0 -> rank 0 through RCCL and are written back
    so = [0]; bodies = []
    for c in range(24):
        m = np.arange(offs[c], offs[c + 1])
        b = np.unique(np.concatenate([pm["body1"][m], pm["body2"][m]])) if len(m) else np.zeros(0, np.int64)
        b = b[sc.rb_type[b] == F.RB_DYNAMIC]
        bodies.append(b); so.append(so[-1] + len(b))
    bodies = np.concatenate(bodies).astype(np.int32)
    looped.halo_plan_upload([0], so, bodies, so, bodies)
    looped.comm_init(hip.comm_unique_id(), 1, 0)
    for _ in range(3):
        plain.step(); looped.step()
    looped.synchronize(); plain.synchronize()
    a, b = plain.bodies_download(), looped.bodies_download()
    for k in a:
        assert np.array_equal(a[k], b[k]), (bits, k)
    ia, ib = plain.impulses_download(), looped.impulses_download()
    for k in ia:
        assert np.array_equal(ia[k], ib[k]), (bits, k)
    assert float(np.abs(a["linear_velocity"]).max()) > 0.05
    # level 1's exchange through the same communicator: ncclAllGather of the device-reduced bounds == avn_dynamic_bounds, no overlaps with itself
    bounds, ov = looped.bounds_exchange()
    mn, mx = looped.dynamic_bounds()
    assert bounds.shape == (1, 6) and np.array_equal(bounds[0], np.concatenate([mn, mx])) and len(ov) == 0, (bounds, mn, mx)
    b0, ov0 = plain.bounds_exchange()     # no communicator: a world is its own only rank
    assert np.array_equal(b0[0], np.concatenate(plain.dynamic_bounds())) and len(ov0) == 0
# round 6: the overflow colour cut into levels, every level an exchange slot -- colours >= 5 moved into the overflow colour, the levels of the global list as the
# planner assigns them; the world is its own peer for every slot
from level2_helpers import overflow_from
for bits in (32, 64):
    sc, pm, offs, _ = global_problem(orc, 6, 3, 4, seed=12)
    offs = overflow_from(offs, 5)
    o0, o1 = int(offs[23]), int(offs[24])
    depth = np.zeros(sc.n, np.int64); level = np.zeros(o1 - o0, np.int64)
    for m in range(o0, o1):
        bb = [int(b) for b in (pm["body1"][m], pm["body2"][m]) if sc.rb_type[b] != F.RB_STATIC]
        level[m - o0] = max(depth[b] for b in bb)
        for b in bb:
            depth[b] = level[m - o0] + 1
    L = int(level.max()) + 1
    assert L > 3
    plain = make_single(hip, bits, sc, pm, offs, 0.3, 3)
    looped = make_single(hip, bits, sc, pm, offs, 0.3, 3)
    so = [0]; bodies = []
    for slot in range(23 + L):
        m = np.arange(offs[slot], offs[slot + 1]) if slot < 23 else o0 + np.flatnonzero(level == slot - 23)
        b = np.unique(np.concatenate([pm["body1"][m], pm["body2"][m]])) if len(m) else np.zeros(0, np.int64)
        b = b[sc.rb_type[b] == F.RB_DYNAMIC]
        bodies.append(b); so.append(so[-1] + len(b))
    bodies = np.concatenate(bodies).astype(np.int32)
    looped.halo_overflow_levels_upload(L, level)
    looped.halo_plan_upload([0], so, bodies, so, bodies)
    looped.comm_init(hip.comm_unique_id(), 1, 0)
    for _ in range(3):
        plain.step(); looped.step()
    looped.synchronize(); plain.synchronize()
    a, b = plain.bodies_download(), looped.bodies_download()
    for k in a:
        assert np.array_equal(a[k], b[k]), ("levels", bits, k)
    ia, ib = plain.impulses_download(), looped.impulses_download()
    for k in ia:
        assert np.array_equal(ia[k], ib[k]), ("levels", bits, k)
# round 6: the joint slot through the library's own exchange -- a jointed stack whose every jointed body travels rank 0 -> rank 0 after the joint systems of every substep
from level2_helpers import stack_joints
for bits in (32, 64):
    sc, pm, offs, _ = global_problem(orc, 6, 3, 4, seed=13)
    jkw = stack_joints(sc, 6, 3, 4, seed=3, damped=True)
    plain = make_single(hip, bits, sc, pm, offs, 0.0, 3); plain.joints_upload(**jkw)
    looped = make_single(hip, bits, sc, pm, offs, 0.0, 3); looped.joints_upload(**jkw)
    so = [0]; bodies = []
    for c in range(24):
        m = np.arange(offs[c], offs[c + 1])
        b = np.unique(np.concatenate([pm["body1"][m], pm["body2"][m]])) if len(m) else np.zeros(0, np.int64)
        b = b[sc.rb_type[b] == F.RB_DYNAMIC]
        bodies.append(b); so.append(so[-1] + len(b))
    jb = np.unique(np.concatenate([jkw["body1"], jkw["body2"]])); jb = jb[sc.rb_type[jb] == F.RB_DYNAMIC]
    bodies.append(jb); so.append(so[-1] + len(jb))       # slot 24: the joint slot
    bodies = np.concatenate(bodies).astype(np.int32)
    looped.halo_joint_slot_set(True, True)
    looped.halo_plan_upload([0], so, bodies, so, bodies)
    looped.comm_init(hip.comm_unique_id(), 1, 0)
    for _ in range(3):
        plain.step(); looped.step()
    looped.synchronize(); plain.synchronize()
    a, b = plain.bodies_download(), looped.bodies_download()
    for k in a:
        assert np.array_equal(a[k], b[k]), ("joint slot", bits, k)
    ja, jb_ = plain.joints_download(), looped.joints_download()
    for k in ja:
        assert np.array_equal(ja[k], jb_[k]), ("joint slot", bits, k)
    assert float(np.abs(ja["total_lagrange"]).max()) > 1e-4
print("SELF_EXCHANGE_OK")
"""


def test_library_exchange_over_rccl_self_peer():
    """In a subprocess under a timeout: a hung collective must fail the test, not the box."""
    code = SELF_EXCHANGE % {"tests": os.path.join(REPO, "tests"), "repo": REPO}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, cwd=REPO)
    assert r.returncode == 0 and "SELF_EXCHANGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_halo_plan_without_communicator_is_refused():
    hip, orc = hip_lib(), oracle_lib()
    sc, pm, offs, _ = global_problem(orc, 4, 2, 2)
    w = make_single(hip, 32, sc, pm, offs, 0.0, 2)
    w.halo_plan_upload([1], [0] * 24 + [1], [1], [0] * 24 + [1], [1])
    with pytest.raises(F.AvnError):
        w.step()
    with pytest.raises(F.AvnError):
        w.halo_plan_upload([1], [0] * 24 + [1], [10 ** 6], [0] * 24 + [1], [1])
    w.halo_plan_upload([], [], [], [], [])   # back to a plain world
    w.step()
    # a plan that names a body the next, smaller upload no longer has is dropped with the manifolds that named it (no out-of-range gather)
    last = sc.n - 1
    w.halo_plan_upload([1], [0] * 24 + [1], [last], [0] * 24 + [1], [last])
    small = {k: (np.asarray(v)[:last] if v is not None else None) for k, v in sc.body_kwargs().items()}
    w.bodies_upload(**small)
    w.step()
