"""CPU: pin the oracle.  (1) the reference's own known-answer tests for the path (tests/golden/reference_kats.json,
transcribed with file:line), within the reference's own epsilons; (2) the frozen oracle vectors, bit-exact (drift guard)."""
import pytest

import golden_checks as G
from helpers import oracle_lib


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("case", G.load_kats(), ids=lambda c: c["name"])
def test_oracle_meets_reference_kat(case, bits):
    G.check_kat(G.run_kat(oracle_lib(), case, bits), case)


@pytest.mark.parametrize("bits", [32, 64])
def test_oracle_matches_frozen_vectors(bits):
    got = G.solver_vectors(oracle_lib(), bits, oracle_lib())
    got.update(G.broadphase_vectors(oracle_lib(), bits))
    got.update(G.joints_vectors(oracle_lib(), bits, oracle_lib()))
    got.update(G.narrow_vectors(oracle_lib(), bits))
    assert G.check_vectors(got, bits) == 57


def test_oracle_is_deterministic_run_to_run():
    """src/tests/mod.rs:151-183 `cubes_simulation_is_locally_deterministic`: 4 runs must be identical (self-consistency)."""
    import numpy as np
    runs = [G.solver_vectors(oracle_lib(), 32, oracle_lib()) for _ in range(4)]
    for r in runs[1:]:
        for k in runs[0]:
            assert np.array_equal(runs[0][k], r[k], equal_nan=True), k


@pytest.mark.parametrize("bits", [32, 64])
def test_oracle_meets_reference_coefficient_combine_and_solver_body_kats(bits):
    G.check_coefficient_combine(oracle_lib(), bits)
    G.check_solver_body_membership(oracle_lib(), bits)
