"""GPU: the HIP path against the COMMITTED golden fixtures, through the C ABI, without executing the oracle for the
numbers: (1) reference KATs within the reference's epsilons, (2) frozen vectors bit-exact, (3) run-to-run determinism."""
import numpy as np
import pytest

import golden_checks as G
from helpers import hip_lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("case", G.load_kats(), ids=lambda c: c["name"])
def test_hip_meets_reference_kat(case, bits):
    G.check_kat(G.run_kat(hip_lib(), case, bits), case)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_hip_matches_frozen_vectors(bits, use_graph):
    got = G.solver_vectors(hip_lib(), bits, hip_lib(), use_graph=use_graph)   # product colouring, product solver
    got.update(G.broadphase_vectors(hip_lib(), bits))
    got.update(G.joints_vectors(hip_lib(), bits, hip_lib(), use_graph=use_graph))
    got.update(G.narrow_vectors(hip_lib(), bits))
    assert G.check_vectors(got, bits) == 57


def test_hip_is_deterministic_run_to_run():
    """src/tests/mod.rs:151-183: the reference requires 4 identical runs; so do we (no atomics on the float path)."""
    runs = [G.solver_vectors(hip_lib(), 32, hip_lib()) for _ in range(4)]
    for r in runs[1:]:
        for k in runs[0]:
            assert np.array_equal(runs[0][k], r[k], equal_nan=True), k


@pytest.mark.parametrize("bits", [32, 64])
def test_hip_meets_reference_coefficient_combine_and_solver_body_kats(bits):
    """physics_material.rs:398-446 and solver_body/plugin.rs:318-353 (see golden_checks)."""
    G.check_coefficient_combine(hip_lib(), bits)
    G.check_solver_body_membership(hip_lib(), bits)
