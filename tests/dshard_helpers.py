"""Shared by tests/test_dshard_cpu.py and tests/test_gpu_dshard.py: the DEVICE closed loop sharded by islands (avn_dshard_*) against the single world."""
import numpy as np

from avian_amd import shard
from helpers import F


def make_worlds(lib, bits, bodies, colliders, owner, n_ranks, substeps=4):
    """the single world and one world per rank: every world holds every body; rank r simulates the bodies with owner == r"""
    def one():
        w = F.World(lib, F.default_config(bits, substeps=substeps))
        w.bodies_upload(**bodies); w.colliders_upload(**colliders); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        return w
    ref = one()
    ranks = []
    for r in range(n_ranks):
        w = one()
        w.dshard_enable(n_ranks, r, owner)
        ranks.append(w)
    return ref, ranks


def compare(step, ref, ranks, owner, rows=False):
    """every rank: the replicated colour lists WITH ORDER, the new pairs and their ids, the counters; the bodies it simulates; and -- the exchange -- every body"""
    off, handles = ref.pipeline_handles()
    pr, ir = ref.pairs_get(), ref.pipeline_new_pair_ids()
    edge_owner = ref.__dict__.setdefault("_edge_owner", {})   # ContactId -> the rank of its non-static body, kept up to date EVERY step (ids are reused)
    for p, i in zip(pr, ir):
        o1, o2 = owner[p["body1"]], owner[p["body2"]]
        edge_owner[int(i)] = int(o1 if o1 >= 0 else o2)
    sr = ref.pipeline_stats()
    br = ref.bodies_download()
    total = 0
    for r, w in enumerate(ranks):
        o2, h2 = w.pipeline_handles()
        assert np.array_equal(off, o2) and np.array_equal(handles, h2), f"step {step}: rank {r}'s replicated colour lists differ from the single world's (content or order)"
        assert np.array_equal(pr, w.pairs_get()) and np.array_equal(ir, w.pipeline_new_pair_ids()), f"step {step}: rank {r}: new pairs / ContactIds differ"
        s = w.pipeline_stats()
        for f in ("pairs_added", "pairs_removed", "manifolds_pushed", "manifolds_popped", "last_status_changes", "active_pairs"):
            assert getattr(s, f) == getattr(sr, f), f"step {step}: rank {r}: pipeline_stats.{f} {getattr(s, f)} != {getattr(sr, f)}"
        b = w.bodies_download()
        for k in br:
            assert np.array_equal(br[k], b[k]), f"step {step}: rank {r}: bodies.{k} differ from the single world's (own bodies: the solver; the others: the exchange)"
        d = w.dshard_stats()
        assert d.global_manifolds == len(handles) and d.own_bodies == int((owner == r).sum())
        total += d.own_manifolds
        if rows and len(handles):
            ids = np.unique(handles)
            mine = np.array([edge_owner.get(int(i)) == r for i in ids])
            ro, rw = ref.contacts_download(ids[mine]), w.contacts_download(ids[mine])
            for k in ro:
                assert np.array_equal(ro[k], rw[k]), f"step {step}: rank {r}: contact rows {k} of its own pairs differ"
    assert total == len(handles), f"step {step}: the ranks' shares of the colour lists do not add up to the single world's ({total} vs {len(handles)})"


def run(lib, bits, bodies, colliders, owner, n_ranks, steps, substeps=4, rows_every=0):
    ref, ranks = make_worlds(lib, bits, bodies, colliders, owner, n_ranks, substeps)
    for s in range(steps):
        ref.step()
        shard.dshard_step_in_process(ranks)
        compare(s, ref, ranks, owner, rows=bool(rows_every) and s % rows_every == rows_every - 1)
    return ref, ranks
