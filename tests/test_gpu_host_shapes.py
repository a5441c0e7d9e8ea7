"""GPU: host shapes on the HIP backend (include/avian_mi355x.h "host shapes", avian_amd/csrc/world/host_shapes.hpp).  Colliders flagged AVN_SHAPE_HOST keep their
aabb / contact_manifolds on the host (callbacks; here: their real Ball / Cuboid geometry through the oracle's batch query), the rest of update_aabb / update_contacts
runs in the kernels.  The hosted HIP world == the native HIP world == the native oracle world, bit for bit, every step; the query list's grow-and-retry path; the
host-bookkeeping mode (AVN_SYS_NARROW_PHASE with the change list)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from host_shape_helpers import assert_same_closed_loop_step, make_pair
from pipeline_scenes import dropped_boxes

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bits,seed,n", [(32, 1, 48), (64, 2, 48), (32, 5, 400)])
def test_host_flagged_colliders_equal_the_native_world_on_hip(bits, seed, n):
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=seed, n=n)
    rng = np.random.default_rng(seed)
    host = rng.random(len(colliders["shape"])) < 0.35
    host[0] = seed % 2 == 0
    native, hosted, hs = make_pair(hip, orc, bits, bodies, colliders, host)
    ref, _, _ = make_pair(orc, orc, bits, bodies, colliders, np.zeros_like(host))
    for w in (native, hosted, ref):
        w.pipeline_enable()
    for step in range(30):
        native.step(); hosted.step(); ref.step()
        assert_same_closed_loop_step(native, hosted, step)
        a, b = ref.bodies_download(), hosted.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k} against the native oracle"
    st = hosted.host_shape_stats()
    assert st.host_colliders == int(host.sum()) and hs.manifold_queries > 100
    # what crossed the bus is the host colliders' queries and answers, nothing else
    aq, ab, mq, mm = (d.itemsize for d in F.host_shape_dtypes(bits))
    assert st.bytes_to_host == hs.aabb_queries * aq + hs.manifold_queries * mq
    assert st.bytes_from_host == hs.aabb_queries * ab + hs.manifold_queries * (mq + mm)


RETRY = r"""
import os, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(repo)r)
os.environ["AVN_HS_QUERY_CAP"] = "7"
import numpy as np
from helpers import F, hip_measure_lib, oracle_lib
from host_shape_helpers import assert_same_closed_loop_step, make_pair
from pipeline_scenes import dropped_boxes
hip, orc = hip_measure_lib(), oracle_lib()
bodies, colliders = dropped_boxes(seed=7, n=120)
host = np.ones(len(colliders["shape"]), bool)
native, hosted, hs = make_pair(hip, orc, 32, bodies, colliders, host)
native.pipeline_enable(); hosted.pipeline_enable()
most = 0
for step in range(25):
    native.step(); hosted.step()
    assert_same_closed_loop_step(native, hosted, step)
    most = max(most, hosted.host_shape_stats().last_manifold_queries)
assert most > 7 * 4, most   # the list started with room for 7 queries: it grew several times, each time through the host-only retry
print("RETRY_OK", most)
"""


def test_query_list_grows_through_the_host_only_retry():
    code = RETRY % {"tests": os.path.join(REPO, "tests"), "repo": REPO}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0 and "RETRY_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_host_shapes_in_the_host_bookkeeping_mode():
    """An Avian integration's mode: the host keeps ContactGraph / ConstraintGraph (avian_amd.pipeline.ContactPipeline) and drives UPDATE_AABB /
    COLLECT_COLLISION_PAIRS / NARROW_PHASE / SOLVER; the host pairs' changes arrive in the same change list."""
    from avian_amd.pipeline import ContactPipeline
    hip, orc = hip_lib(), oracle_lib()
    bodies, colliders = dropped_boxes(seed=4, n=40)
    rng = np.random.default_rng(4)
    host = rng.random(len(colliders["shape"])) < 0.4
    native, hosted, hs = make_pair(hip, orc, 32, bodies, colliders, host)
    pa, pb = ContactPipeline(native, hip), ContactPipeline(hosted, hip)
    for step in range(20):
        pa.step(); pb.step()
        assert not hosted.host_shape_errors()
        a, b = native.bodies_download(), hosted.bodies_download()
        for k in a:
            assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
        assert sorted(pa.pairs) == sorted(pb.pairs) and list(pa.active) == list(pb.active)
        ids = np.array(sorted(pa.pairs), np.uint32)
        ra, rb = native.contacts_download(ids), hosted.contacts_download(ids)
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), f"step {step}: contact rows.{k}"
    assert hs.manifold_queries > 50


def test_host_shapes_without_callbacks_fail_loudly_on_hip():
    hip = hip_lib()
    bodies, colliders = dropped_boxes(seed=3, n=8)
    cols = dict(colliders); cols["shape"] = np.full(len(colliders["shape"]), F.SHAPE_HOST, np.uint8)
    w = F.World(hip, F.default_config(32))
    w.bodies_upload(**bodies); w.colliders_upload(**cols); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.pipeline_enable()
    with pytest.raises(F.AvnError):
        w.step()


def test_capsules_on_hip_equal_the_oracle_given_the_same_host_answers():
    """The capsule callbacks (a shape neither backend has a kernel for) feed the HIP world and the oracle world: same host answers => same world, bit for bit,
    240 steps; and the capsules lie on the ground at the end."""
    from host_shape_helpers import capsule_world
    wh, ch, top = capsule_world(hip_lib(), 32)
    wo, co, _ = capsule_world(oracle_lib(), 32)
    for step in range(240):
        wh.step(); wo.step()
        if step % 8 == 7 or step < 40:
            a, b = wo.bodies_download(), wh.bodies_download()
            for k in a:
                assert np.array_equal(a[k], b[k]), f"step {step}: bodies.{k}"
    assert not wh.host_shape_errors() and ch.queries == co.queries
    y = wh.bodies_download()["position"][1:, 1]
    assert (y > top + 0.25 - 0.03).all() and np.median(y) < top + 0.25 + 0.05


def test_host_shapes_with_sleeping_and_a_despawn_on_hip():
    """Host-flagged colliders next to avn_sleeping_enable and an avn_despawn of a host-shaped body: hosted HIP world == native HIP world every step (bodies, colour
    lists, island ids, Sleeping flags, timers), and the hosted HIP world == the hosted oracle world at the end."""
    from test_host_shapes_cpu import _combo
    slept, hs = _combo(hip_lib(), oracle_lib())
    assert hs.manifold_queries > 200 and slept > 0
