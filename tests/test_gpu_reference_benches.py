"""GPU: the reference's own 3D bench scenes (benches/src/dim3/large_pyramid.rs, many_pyramids.rs) closed loop — device
broad phase + device narrow phase + solver through avn_pipeline_enable — against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,make", [("large_pyramid_base40", lambda: scenes.large_pyramid(40)),
                                       ("many_pyramids_3x4_base8", lambda: scenes.many_pyramids(8, 3, 4))])
def test_reference_bench_scene_closed_loop_matches_oracle(name, make):
    sc = make()
    worlds = []
    for lib in (oracle_lib(), hip_lib()):
        w = F.World(lib, F.default_config(32, substeps=4))
        w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
        w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
        w.pipeline_enable()
        worlds.append(w)
    wo, wh = worlds
    for s in range(25):
        wo.step(); wh.step()
    bo, bh = wo.bodies_download(), wh.bodies_download()
    for k in bo:
        assert np.array_equal(bo[k], bh[k]), f"{name}: bodies.{k}"
    oo, oh = wo.pipeline_handles(), wh.pipeline_handles()
    assert np.array_equal(oo[0], oh[0]) and np.array_equal(oo[1], oh[1])
    n_dyn = int((sc.rb_type == F.RB_DYNAMIC).sum())
    assert wh.pipeline_stats().manifolds > n_dyn, "every box rests on something"
    assert float(np.abs(bh["linear_velocity"]).max()) < 1.0 and float(np.abs(bh["position"][:, 2]).max()) < 0.05, "the pyramid stands"
