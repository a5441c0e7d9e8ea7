"""CPU: despawn inside the closed loop (avn_despawn) on the oracle, against an INDEPENDENT model of the reference's remove_collider written in
Python over the low-level ABI (avian_amd/pipeline.py: per-collider edge lists in insertion order walked backwards, pops through the host
ConstraintGraph, ids back into a heap) -- collision/narrow_phase/mod.rs:399-457, contact_types/contact_graph.rs:641-700,
data_structures/stable_graph.rs:251-315, data_structures/id_pool.rs:31-40.

World A: the library's closed loop (avn_pipeline_enable) + avn_despawn, remaining bodies RENUMBERED and re-uploaded.
World B: the Python driver; the despawned bodies stay in the arrays without a collider (they fall forever, touching nothing), so its body
indices never change.  Colour lists (ContactIds, with their order) must be equal; bodies are compared through the index map."""
import numpy as np
import pytest

from avian_amd.pipeline import ContactPipeline
from helpers import F, oracle_lib
from pipeline_scenes import dropped_boxes, stack_and_projectile


def subset(d, keep):
    return {k: (np.asarray(v)[keep] if isinstance(v, np.ndarray) and len(v) == len(keep) else v) for k, v in d.items()}


def despawn_and_reupload(w, bodies_kw, colliders_kw, gone, alive_mask):
    """avn_despawn + the uploads the header asks for: dynamic state from the world, static attributes from the scene, both compacted."""
    state = w.bodies_download()
    w.despawn(bodies=gone)
    kw = subset(bodies_kw, alive_mask)
    for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
        kw[k] = state[k][alive_mask].astype(np.float64)
    w.bodies_upload(**kw)
    new_index = np.cumsum(alive_mask) - 1
    ck = subset(colliders_kw, alive_mask)
    ck["body"] = new_index[np.asarray(colliders_kw["body"])[alive_mask]].astype(np.int32)
    w.colliders_upload(**ck)
    w.collider_materials_upload(friction=0.6, restitution=0.0)
    return new_index


@pytest.mark.parametrize("bits", [32, 64])
def test_despawn_matches_the_python_model_of_remove_collider(bits):
    lib = oracle_lib()
    bodies, colliders = dropped_boxes(seed=31, n=60)
    n = len(bodies["inv_mass"])
    wa = F.World(lib, F.default_config(bits, substeps=4))
    wa.bodies_upload(**bodies); wa.colliders_upload(**colliders); wa.existing_pairs_upload(np.zeros(0, np.uint64)); wa.collider_materials_upload(friction=0.6, restitution=0.0)
    wa.pipeline_enable()
    wb = F.World(lib, F.default_config(bits, substeps=4))
    wb.bodies_upload(**bodies); wb.colliders_upload(**colliders); wb.existing_pairs_upload(np.zeros(0, np.uint64)); wb.collider_materials_upload(friction=0.6, restitution=0.0)
    pb = ContactPipeline(wb, lib)
    alive = np.ones(n, bool)          # in ORIGINAL numbering
    bodies_a, colliders_a = dict(bodies), dict(colliders)   # what world A holds (compacted along the way)
    orig_of_a = np.arange(n)          # world A's body i is original body orig_of_a[i]
    rng = np.random.default_rng(7)

    def compare(tag):
        offa, ha = wa.pipeline_handles()
        offb, hb = pb.graph.lists()
        assert np.array_equal(offa, offb), f"{tag}: colour offsets\n{offa}\n{offb}"
        assert np.array_equal(ha, hb.astype(np.uint32)), f"{tag}: colour lists (ids or their order) differ"
        ba, bb = wa.bodies_download(), wb.bodies_download()
        for k in ba:
            assert np.array_equal(ba[k], bb[k][orig_of_a]), f"{tag}: bodies.{k}"

    total_removed = 0
    for s in range(150):
        if s in (40, 70, 100):
            # despawn ~12 % of the bodies still alive, in a scrambled order (the order decides the pops' order)
            cand = np.flatnonzero(alive)[1:]   # (never the ground)
            gone_orig = rng.permutation(cand)[:max(2, len(cand) // 8)]
            a_index_of = {int(o): i for i, o in enumerate(orig_of_a)}
            gone_a = np.array([a_index_of[int(o)] for o in gone_orig], np.uint32)
            mask_a = np.ones(len(orig_of_a), bool); mask_a[gone_a] = False
            despawn_and_reupload(wa, bodies_a, colliders_a, gone_a, mask_a)
            bodies_a, colliders_a = subset(bodies_a, mask_a), subset(colliders_a, mask_a)
            colliders_a["body"] = np.arange(int(mask_a.sum()), dtype=np.int32)
            orig_of_a = orig_of_a[mask_a]
            # world B: remove_collider per despawned body (one collider each, entity = 100 + original index), then the remaining colliders
            for o in gone_orig:
                pb.remove_collider(100 + int(o))
            alive[gone_orig] = False
            wb.colliders_upload(**subset(colliders, alive))
            wb.collider_materials_upload(friction=0.6, restitution=0.0)
            total_removed += len(gone_orig)
            compare(f"after the despawn before step {s}")
        wa.step(); pb.step()
        compare(f"step {s}")
    st = wa.pipeline_stats()
    assert total_removed >= 15 and st.pairs_removed == pb.stats["pairs_removed"] and st.pairs_added == pb.stats["pairs_added"]
    assert st.manifolds_popped == pb.stats["pops"] and st.manifolds_pushed == pb.stats["pushes"]
    assert max(pb.pairs) < pb.stats["pairs_added"], "freed ContactIds are handed out again, lowest first"


def test_despawn_protocol_errors():
    lib = oracle_lib()
    sc = stack_and_projectile(2, 2, 2, height=5.0)
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    with pytest.raises(F.AvnError):
        w.despawn(bodies=[1])            # needs the closed loop
    w.pipeline_enable()
    for _ in range(5):
        w.step()
    with pytest.raises(F.AvnError):
        w.despawn(bodies=[1, 1])         # listed twice
    with pytest.raises(F.AvnError):
        w.despawn(bodies=[sc.n + 3])     # out of range
    w.despawn(bodies=[3])
    with pytest.raises(F.AvnError):
        w.step()                         # the host owes the remaining bodies and colliders
