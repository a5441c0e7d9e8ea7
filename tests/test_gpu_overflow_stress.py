"""GPU: the overflow colour's hand-over through TAGS in the velocity records (k_overflow_flow_tag, f32) against the TICKET form (k_overflow_flow; does not rely
on a 16-byte record being single-copy atomic across XCDs) and against the oracle, bit for bit, over REPEATED collapses of a lattice whose overflow colour is
10^4 manifolds strong and hundreds of dependent hops deep.  A torn record, or a substep kernel that does not pass the w lanes of the velocity records through,
shows up as a body that differs (k_contacts.hip: "WHAT THIS RELIES ON").  tools/stress_ovf.py is the same at any size."""
import os
import subprocess
import sys

import numpy as np
import pytest

from avian_amd import scenes
from helpers import F, hip_lib, oracle_lib

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(lib, sc):
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    return w


def test_tag_hand_over_equals_the_oracle_over_repeated_collapses():
    sc = scenes.box_stack(20, 16, 20)   # 6 400 boxes: the collapse puts ~10^4 manifolds into the overflow colour
    steps, reps = 14, 6
    wo = make(oracle_lib(), sc)
    ref, ovf = [], 0
    for _ in range(steps):
        wo.step(); ref.append(wo.bodies_download()); ovf = max(ovf, wo.pipeline_stats().last_overflow_manifolds)
    assert ovf > 5000, ovf
    for r in range(reps):
        w = make(hip_lib(), sc)
        for s in range(steps):
            w.step()
            b = w.bodies_download()
            for k in b:
                assert np.array_equal(b[k], ref[s][k]), f"run {r}, step {s}: bodies.{k} differs from the oracle (tag hand-over)"
        w.close()


def test_ticket_form_agrees_on_the_same_collapse():
    """the round-2 ticket form behind AVN_OVF_TICKETS (`make measure` build) through a slice of the closed-loop parity suite: the two hand-overs are checked against the same oracle"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_graph.py::test_medium_stack_6400_boxes_first_steps",
                        "tests/test_gpu_overflow_stress.py::test_tag_hand_over_equals_the_oracle_over_repeated_collapses"], capture_output=True, text=True, timeout=900, cwd=REPO,
                       env=dict(os.environ, AVN_LIB_PATH=os.path.join(REPO, "avian_amd", "csrc", "measure", "libavian_mi355x.so"), AVN_OVF_TICKETS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
