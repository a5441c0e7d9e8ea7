"""CPU: the oracle's XPBD joints.  The reference holds no numeric test for 3D joints
("parity unpinned"), so these are the physical invariants each joint type exists to enforce, checked on the oracle
(and, in tests/test_gpu_joints.py, on the HIP path), plus the accuracy of the shared deterministic asin."""
import ctypes as C

import numpy as np
import pytest

from helpers import F, oracle_lib
from joint_scenes import JOINT_CASES, check_joint_case, run_joint_case


def test_deterministic_asin_accuracy():
    dll = oracle_lib().dll
    dll.avo_asin_f32.argtypes = [C.c_float]; dll.avo_asin_f32.restype = C.c_float
    dll.avo_asin_f64.argtypes = [C.c_double]; dll.avo_asin_f64.restype = C.c_double
    xs = np.concatenate([np.linspace(-1, 1, 4001), [1e-8, -1e-8, 0.5, -0.5, 0.4999999, 0.9999999, 0.0]])
    for x in xs:
        r32 = dll.avo_asin_f32(float(np.float32(x))); e32 = np.arcsin(np.float64(np.float32(x)))
        assert abs(r32 - e32) <= 2.5 * np.spacing(np.float32(max(abs(e32), 1e-30))), (x, r32, e32)
        r64 = dll.avo_asin_f64(float(x)); e64 = np.arcsin(x)
        assert abs(r64 - e64) <= 4 * np.spacing(max(abs(e64), 1e-300)), (x, r64, e64)
    assert np.isnan(dll.avo_asin_f32(1.5)) and np.isnan(dll.avo_asin_f64(-1.0000001))


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("case", sorted(JOINT_CASES))
def test_oracle_joint_invariants(case, bits):
    check_joint_case(case, run_joint_case(oracle_lib(), case, bits))
