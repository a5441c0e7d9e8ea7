"""GPU: the A/B switches of the closed loop and of avn_step keep their (older) code paths correct.  Each runs a slice of the parity suite in a
subprocess with the switch set: round 2's one-wave replay of the colour lists (AVN_PG_REPLAY_WAVE), the narrow phase behind the broad phase
(AVN_NO_NP_OVERLAP), blocking waits (AVN_NO_SPIN_SYNC), fencing events and the broad phase enqueued first (AVN_EVENT_SYSTEM_FENCE,
AVN_BP_ENQUEUE_FIRST), joint islands on the main stream (AVN_NO_ISLAND_STREAMS), the broad phase on the solver's stream (AVN_NO_BP_OVERLAP), the
one-lane-per-body warm start in the closed loop (AVN_WS_LANE_PER_BODY; round 4's default there is four lanes per body)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLOSED_LOOP = ["tests/test_gpu_graph.py::test_churning_pile_replays_swap_remove_exactly", "tests/test_gpu_graph.py::test_pairs_removed_and_ids_reused_like_the_oracle",
               "tests/test_gpu_graph.py::test_medium_stack_6400_boxes_first_steps"]
FROZEN = ["tests/test_gpu_parity.py", "tests/test_gpu_island_streams.py"]
SLEEPING = ["tests/test_gpu_sleeping.py::test_large_stack_sleeps_and_wakes_in_thousands_of_ops", "tests/test_gpu_sleeping.py::test_a_world_entirely_asleep_steps_as_the_identity_and_can_be_woken"]


@pytest.mark.parametrize("env,targets", [
    ({"AVN_PG_REPLAY_WAVE": "1"}, CLOSED_LOOP),
    ({"AVN_NO_NP_OVERLAP": "1", "AVN_NO_SPIN_SYNC": "1"}, CLOSED_LOOP),
    ({"AVN_EVENT_SYSTEM_FENCE": "1", "AVN_BP_ENQUEUE_FIRST": "1"}, FROZEN + CLOSED_LOOP[:1]),
    ({"AVN_NO_ISLAND_STREAMS": "1", "AVN_NO_BP_OVERLAP": "1"}, ["tests/test_gpu_parity.py", "tests/test_gpu_configs.py::test_cfg3_stack_with_distance_joint_chains_matches_oracle"]),
    ({"AVN_WS_LANE_PER_BODY": "1"}, CLOSED_LOOP),
    # round 6, sleeping: the island manager's own edge lists instead of the device-built adjacency; the walk on the calling thread; the manager's digest in front of the solver
    ({"AVN_SLP_HOST_SPLIT": "1"}, SLEEPING),
    ({"AVN_SLP_SYNC_SPLIT": "1", "AVN_SLP_CHECK_ADJ": "1"}, SLEEPING),
    ({"AVN_SLP_NO_FAST": "1"}, SLEEPING),
    ({"AVN_SLP_ADJ_MIN": "1"}, SLEEPING),
], ids=["wave-replay", "serial-narrow-phase-blocking-waits", "fencing-events-bp-first", "single-stream", "lane-per-body-warm-start",
        "sleeping-host-split", "sleeping-sync-split-checked", "sleeping-digest-first", "sleeping-device-adjacency-for-every-island"])
def test_parity_slice_with_the_switch_set(env, targets):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *targets], capture_output=True, text=True, timeout=600, cwd=REPO,
                       env=dict(os.environ, AVN_LIB_PATH=os.path.join(REPO, "avian_amd", "csrc", "measure", "libavian_mi355x.so"), **env))   # (only the `make measure` build reads the environment)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout   # (AVN_SLP_CHECK_ADJ: a device neighbour list that differs from the manager's own fails the step with AVN_ERR_STATE)
