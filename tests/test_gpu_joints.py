"""GPU: the joint invariants of tests/joint_scenes.py on the HIP path, and HIP == oracle bit for bit on the same scenes."""
import numpy as np
import pytest

from helpers import hip_lib, oracle_lib
from joint_scenes import JOINT_CASES, check_joint_case, run_joint_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("case", sorted(JOINT_CASES))
def test_hip_joint_invariants_and_oracle_identity(case, bits):
    got = run_joint_case(hip_lib(), case, bits)
    check_joint_case(case, got)
    ref = run_joint_case(oracle_lib(), case, bits)
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True), f"{case}: {k} differs from the oracle after 120 steps"
