"""GPU: the DEVICE closed loop sharded by islands (avn_dshard_*: the bookkeeping of avn_pipeline_enable(1) replicated on every rank's device, the solver and one body
all-gather per step sharded) on the HIP backend, as several worlds on the one GPU a test box has -- against the single HIP world after every step: colour lists with
order, new pairs and their ContactIds, counters, EVERY body (own: the sharded solver; foreign: the exchange), the ranks' own contact rows.
  * tumbling piles over 2 / 3 ranks, f32 and f64 (ContactId reuse, swap_removes that move another rank's handle);
  * the reference's Many Pyramids bench scene (5 500 boxes, 100 islands) over 4 ranks by whole pyramids, 120 steps (the single world solves it in island blocks,
    the ranks in colour launches: the same bits);
  * cfg2 with two stacks (200 000 boxes) over 2 ranks through the collapse;
  * the library-issued exchange (avn_comm_init: pack -> ncclAllGather -> unpack on the world's stream inside avn_step) with a world that is its own only rank;
  * two processes over gloo (the host-mediated exchange as tensors)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import dshard_helpers as D
from avian_amd import scenes, shard
from helpers import F, REPO, hip_lib
from test_dshard_cpu import owner_by_pile
from test_sharded_closed_loop_cpu import piles

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,n_piles", [(32, 2), (64, 2), (32, 3)])
def test_piles_over_ranks_equal_the_single_world(bits, n_piles):
    bodies, colliders = piles(n_piles, 24)
    owner = owner_by_pile(bodies, n_piles, 24)
    ref, ranks = D.run(hip_lib(), bits, bodies, colliders, owner, n_piles, steps=90, rows_every=30)
    assert ref.pipeline_stats().pairs_removed > 0 and all(w.dshard_stats().own_manifolds > 0 for w in ranks)


def test_many_pyramids_over_four_ranks_120_steps():
    sc = scenes.many_pyramids(10, 10, 10)
    owner = np.full(sc.n, -1, np.int32)
    per = 55   # boxes of one pyramid (base 10); pyramid p = bodies 10 + 55 p ...
    for p in range(100):
        owner[10 + per * p:10 + per * (p + 1)] = p % 4
    ref, ranks = D.run(hip_lib(), 32, sc.body_kwargs(), sc.collider_kwargs(), owner, 4, steps=120, rows_every=60)
    assert ref.timers().island_blocks > 0, "the single world is expected to solve this scene in island blocks"
    assert all(w.timers().island_blocks == 0 and w.dshard_stats().own_bodies == 25 * 55 for w in ranks)
    assert ref.pipeline_stats().manifolds > 10000


def test_cfg2_with_two_stacks_over_two_ranks_through_the_collapse():
    sc = scenes.box_stacks(2, 50, 40, 50)
    assert sc.n == 200_001
    owner = np.full(sc.n, -1, np.int32)
    owner[1:100_001] = 0; owner[100_001:] = 1
    ref, ranks = D.make_worlds(hip_lib(), 32, sc.body_kwargs(), sc.collider_kwargs(), owner, 2)
    for s in range(30):
        ref.step()
        shard.dshard_step_in_process(ranks)
        if s % 5 == 4 or s < 3:
            D.compare(s, ref, ranks, owner)
    st = ref.pipeline_stats()
    assert st.manifolds > 400_000 and st.last_overflow_manifolds > 10_000
    d = [w.dshard_stats() for w in ranks]
    assert d[0].own_manifolds + d[1].own_manifolds == d[0].global_manifolds and min(x.own_manifolds for x in d) > 150_000
    assert d[0].bytes_sent_per_step == 100_000 * 64


def test_library_issued_exchange_with_a_world_that_is_its_own_rank():
    """avn_comm_init(1 rank) + avn_dshard_enable(1 rank): avn_step packs, all-gathers through RCCL and (having no foreign body) unpacks nothing -- the step must be the
    plain world's, and the exchange must have been issued"""
    hip = hip_lib()
    sc = scenes.box_stack(8, 8, 8)
    owner = np.where(sc.rb_type == 1, -1, 0).astype(np.int32)
    ws = []
    for sharded in (False, True):
        w = F.World(hip, F.default_config(32, substeps=4))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        if sharded:
            w.dshard_enable(1, 0, owner)
            w.comm_init(hip.comm_unique_id(), 1, 0)
        ws.append(w)
    for s in range(25):
        for w in ws: w.step()
        a, b = ws[0].bodies_download(), ws[1].bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), (s, k)
    d = ws[1].dshard_stats()
    assert d.exchanges == 25 and d.own_manifolds == d.global_manifolds == ws[0].pipeline_stats().manifolds > 500


def test_two_processes_over_gloo_on_the_hip_backend(tmp_path):
    out = str(tmp_path / "dshard.npz")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "tests", "dshard_worker.py"), out, "60"], capture_output=True, text=True, timeout=900, cwd=REPO,
                       env=dict(os.environ, OMP_NUM_THREADS="1", AVN_SHARD_BACKEND="hip"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = np.load(out)
    bodies, colliders = piles(2, 24)
    ref = F.World(hip_lib(), F.default_config(32, substeps=4))
    ref.bodies_upload(**bodies); ref.colliders_upload(**colliders); ref.existing_pairs_upload(np.zeros(0, np.uint64)); ref.collider_materials_upload(friction=0.5)
    ref.pipeline_enable()
    for _ in range(60):
        ref.step()
    off, handles = ref.pipeline_handles()
    assert np.array_equal(got["offsets"], off) and np.array_equal(got["handles"], handles)
    for k, v in ref.bodies_download().items():
        assert np.array_equal(got[k], v), f"bodies.{k} after 60 steps over gloo differ from the single world"
