"""GPU: the same physical checks as tests/test_physics_sanity.py on the HIP backend."""
import pytest

from helpers import hip_lib
from physics_scenes import CHECKS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("check", CHECKS, ids=[c.__name__[6:] for c in CHECKS])
@pytest.mark.parametrize("bits", [32, 64])
def test_hip(check, bits):
    check(hip_lib(), bits)
