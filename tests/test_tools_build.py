"""CPU: the hardware probes under tools/ (the measurements DESIGN.md section 8 rests on) still compile for gfx950."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = ["launch_floor_probe.hip", "l2_retention_probe.hip", "neighbour_sync_probe.hip"]


@pytest.mark.parametrize("src", PROBES)
def test_probe_compiles_for_gfx950(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this container")
    out = tmp_path / (src + ".o")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-c", os.path.join(REPO, "tools", src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0
