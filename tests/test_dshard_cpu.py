"""CPU: the closed loop sharded by islands in its DEVICE form (include/avian_mi355x.h: avn_dshard_*) on the oracle backend -- every rank replicates the front of the
step on every body (equal inputs: equal pair sequences, ContactIds, colours and list positions without an exchange), simulates only its own bodies, solves only its
share of the colour lists, and receives the others' bodies once per step.  Two and three ranks in one process against the single world after EVERY step, through
ContactId reuse and swap_removes that move another rank's handle; the refusals; and two processes over gloo (tests/dshard_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import dshard_helpers as D
from helpers import F, REPO, oracle_lib
from test_sharded_closed_loop_cpu import piles


def owner_by_pile(bodies, n_piles, n):
    o = np.full(len(bodies["inv_mass"]), -1, np.int32)
    for k in range(n_piles):
        o[1 + k * n:1 + (k + 1) * n] = k
    return o


@pytest.mark.parametrize("bits,n_piles", [(32, 2), (64, 2), (32, 3)])
def test_device_sharded_closed_loop_equals_the_single_world(bits, n_piles):
    bodies, colliders = piles(n_piles, 24)
    owner = owner_by_pile(bodies, n_piles, 24)
    ref, ranks = D.run(oracle_lib(), bits, bodies, colliders, owner, n_piles, steps=90, rows_every=30)
    st = ref.pipeline_stats()
    assert st.pairs_removed > 0 and st.manifolds_popped > 0, "the run must reuse ContactIds and pop handles"
    assert all(w.dshard_stats().own_manifolds > 0 for w in ranks)


def test_a_manifold_between_two_ranks_fails_the_step_and_the_refusals():
    lib = oracle_lib()
    bodies, colliders = piles(2, 24, gap=0.0)   # the two piles fall onto each other: islands of different ranks meet
    owner = owner_by_pile(bodies, 2, 24)
    ref, ranks = D.make_worlds(lib, 32, bodies, colliders, owner, 2)
    with pytest.raises(F.AvnError, match="two ranks"):
        for _ in range(60):
            for w in ranks: w.step()
            recs = [w.dshard_bodies_pack() for w in ranks]
            ranks[0].dshard_bodies_unpack(1, recs[1]); ranks[1].dshard_bodies_unpack(0, recs[0])
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**bodies); w.colliders_upload(**colliders)
    with pytest.raises(F.AvnError):
        w.dshard_enable(2, 0, owner)          # needs the closed loop
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.pipeline_enable()
    bad = owner.copy(); bad[5] = -1
    with pytest.raises(F.AvnError, match="owner"):
        w.dshard_enable(2, 0, bad)            # a dynamic body nobody simulates
    with pytest.raises(F.AvnError):
        w.dshard_enable(2, 2, owner)          # rank out of range
    w.dshard_enable(2, 0, owner)
    with pytest.raises(F.AvnError):
        w.dshard_bodies_unpack(0, np.zeros((24, 16), np.float32))   # its own rank
    with pytest.raises(F.AvnError):
        w.dshard_bodies_unpack(1, np.zeros((23, 16), np.float32))   # not that rank's body count


def test_two_processes_over_gloo(tmp_path):
    """world_size 2 on gloo: each process one rank, the bodies' records through one tensor all-gather per step; the merged result equals the single world"""
    out = str(tmp_path / "dshard.npz")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "tests", "dshard_worker.py"), out, "60"], capture_output=True, text=True, timeout=600, cwd=REPO, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = np.load(out)
    bodies, colliders = piles(2, 24)
    ref = F.World(oracle_lib(), F.default_config(32, substeps=4))
    ref.bodies_upload(**bodies); ref.colliders_upload(**colliders); ref.existing_pairs_upload(np.zeros(0, np.uint64)); ref.collider_materials_upload(friction=0.5)
    ref.pipeline_enable()
    for _ in range(60):
        ref.step()
    off, handles = ref.pipeline_handles()
    assert np.array_equal(got["offsets"], off) and np.array_equal(got["handles"], handles)
    for k, v in ref.bodies_download().items():
        assert np.array_equal(got[k], v), f"bodies.{k} after 60 steps over gloo differ from the single world"
