"""CPU: the groundwork experiment for the device constraint graph stays true — colouring in dependency rounds
reproduces the library's serial greedy ConstraintGraph (constraint_graph.rs:163-236 restated) exactly on a box stack's contact set."""
import os
import subprocess
import sys

from helpers import REPO


def test_round_parallel_colouring_equals_serial_greedy():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "experiments", "parallel_greedy_depth.py"), "14", "9", "14"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "round-parallel colouring == serial greedy: True" in r.stdout
