"""CPU: the restated algorithm behaves like mechanics says it must.  The reference holds no test vectors for contacts, friction,
restitution or joints (rows a10-a28 are parity-unpinned by reference data); these checks pin the oracle -- and through
bit-identity the HIP product -- to answers that come from physics instead: free fall in closed form, the friction cone, the e^2 rebound,
momentum conservation, a stack that stays where it is, a rigid pendulum.  Closed loop (broad phase, narrow phase, solver) on tiny scenes."""
import pytest

from helpers import oracle_lib
from physics_scenes import CHECKS


@pytest.mark.parametrize("check", CHECKS, ids=[c.__name__[6:] for c in CHECKS])
@pytest.mark.parametrize("bits", [32, 64])
def test_oracle(check, bits):
    check(oracle_lib(), bits)
