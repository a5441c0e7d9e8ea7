"""GPU: island-level concurrency inside the substep loop (DW::side_group, world/joints.hpp rebuild_body_groups): islands that hold joints and
no contact manifold run their substep loop on a second stream next to the other islands' contact passes.  Islands exchange nothing inside
the solver (islands/mod.rs:1-10); the joints of one island keep the serial order of xpbd/plugin.rs:77-82,145-189.  So the split run must
equal the oracle AND the one-stream run bit for bit -- bodies, joints, impulses -- with and without the hipGraph."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, compare_dicts, hip_lib, hip_measure_lib, oracle_lib, random_joints
from test_gpu_configs import setup

pytestmark = pytest.mark.gpu


def mixed_scene(seed=7):
    """A small stack + chains: some hang clear of everything (side islands), one chain is tied to a box of the stack (its island has
    manifolds: main group), joints of all five types with damping inside the side islands."""
    sc, joints = scenes.stack_with_chains(6, 5, 6, 8, 12)
    rng = np.random.default_rng(seed)
    n0 = 6 * 5 * 6 + 1
    J = len(joints["body1"])
    g = random_joints(rng, sc.n, J)          # random frames / limits / compliances / damping of every type ...
    g["body1"] = joints["body1"]; g["body2"] = joints["body2"]   # ... on the chains' connectivity
    g["collision_disabled"] = np.ones(J, np.uint8)
    # tie chain 0's last link to a box in the top layer of the stack, and chain 1's to chain 2's (two chains, one side island)
    extra = dict(body1=np.array([n0 + 11, n0 + 12 + 11], np.int32), body2=np.array([5 * 6 + 3, n0 + 24 + 11], np.int32))
    for k, v in g.items():
        if k in extra:
            g[k] = np.concatenate([v, extra[k]])
        else:
            g[k] = np.concatenate([v, v[:2]])
    sc.linear_velocity[n0:] = rng.normal(scale=0.5, size=(sc.n - n0, 3))
    sc.rb_type[n0::12] = F.RB_KINEMATIC
    return sc, g


@pytest.mark.parametrize("bits,use_graph", [(32, 1), (32, 0), (64, 1)])
def test_side_islands_on_their_own_stream_match_oracle_and_single_stream(bits, use_graph, monkeypatch):
    sc, g = mixed_scene()
    worlds = []
    for lib, env in ((oracle_lib(), None), (hip_lib(), None), (hip_measure_lib(), "1")):   # (the switch exists in the `make measure` build only)
        if env:
            monkeypatch.setenv("AVN_NO_ISLAND_STREAMS", env)
        cfg = F.default_config(bits, substeps=4)
        cfg.use_graph = use_graph
        w = F.World(lib, cfg)
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.joints_upload(**g)
        w.existing_pairs_upload(np.zeros(0, np.uint64))
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        p = w.pairs_get().copy()
        mf = scenes.axis_aligned_manifolds(sc, np.stack([p["body1"], p["body2"]], axis=1))
        offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
        scenes.upload_manifolds(w, scenes.permute_manifolds(mf, perm), offs, sc.friction, sc.restitution)
        worlds.append(w)
        monkeypatch.delenv("AVN_NO_ISLAND_STREAMS", raising=False)
    wo, wh, w1 = worlds
    for s in range(6):
        for w in worlds:
            w.step()
        for other, name in ((wo, "oracle"), (w1, "one stream")):
            compare_dicts(other.bodies_download(), wh.bodies_download(), f"step {s} vs {name}: bodies")
            compare_dicts(other.joints_download(), wh.joints_download(), f"step {s} vs {name}: joints")
    compare_dicts(wo.impulses_download(), wh.impulses_download(), "impulses")
    tm, t1 = wh.timers(), w1.timers()
    n_links = 8 * 12
    # chains 1..7 hang clear (chain 0 is tied to the stack): 7 x 12 bodies in side islands; the one-stream world has none
    assert tm.side_island_bodies == n_links - 12 and t1.side_island_bodies == 0
    assert float(np.abs(wh.joints_download()["total_lagrange"]).max()) > 0.0


def test_cfg3_reports_its_chains_as_side_islands():
    sc, joints = scenes.stack_with_chains(50, 20, 50, 100, 100)
    joints = dict(joints, collision_disabled=np.ones(len(joints["body1"]), np.uint8))
    w = F.World(hip_lib(), F.default_config(32, substeps=4))
    setup(w, hip_lib(), sc, joints)
    w.step(); w.synchronize()
    assert w.timers().side_island_bodies == 10_000
