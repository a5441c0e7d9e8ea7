"""GPU: the x-slab sharded broad phase (avian_amd.shard.slab_*) driven through the HIP library — each slab +
halo swept by the unchanged device broad phase — against the single-world HIP run and the CPU oracle, bit for bit."""
import numpy as np
import pytest

from avian_amd import scenes, shard
from helpers import F, hip_lib, oracle_lib

pytestmark = pytest.mark.gpu


def single(lib, sc):
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get()
    mn, mx, _ = w.aabbs_download()
    w.close()
    return np.stack([p["collider1"], p["collider2"], p["flags"]], axis=1).astype(np.uint32), mn, mx


@pytest.mark.parametrize("scene", ["sparse_60k", "lattice_with_ground"])
def test_slab_sharded_broad_phase_on_device(scene):
    sc = scenes.sparse_mixed(60000, side=60.0) if scene == "sparse_60k" else scenes.box_stack(12, 8, 12)
    ref_o, mn_o, mx_o = single(oracle_lib(), sc)
    ref_h, mn, mx = single(hip_lib(), sc)
    assert np.array_equal(mn, mn_o) and np.array_equal(mx, mx_o)
    assert np.array_equal(ref_h, ref_o) and len(ref_h) > 5000
    none = np.zeros(0, np.uint64)
    for R in (2, 4, 8):
        parts = [shard.slab_broad_phase_step(hip_lib(), 32, sc.body_kwargs(), sc.collider_kwargs(), mn[:, 0], mx[:, 0], none, r, R) for r in range(R)]
        assert np.array_equal(np.concatenate(parts), ref_h), f"{scene}: {R} slabs"
