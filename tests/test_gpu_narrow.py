"""GPU: the HIP contact_query::contact_manifolds (avn_contact_manifolds, k_narrow.hip) against the CPU oracle bit for
bit, plus the geometric invariants on the product's own output."""
import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib
from narrow_checks import check_invariants, check_swap_symmetry
from narrow_scenes import random_pairs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("seed,n", [(1, 2000), (7, 20000)])
def test_contact_manifolds_match_oracle_bit_for_bit(bits, seed, n):
    pairs = random_pairs(seed, n)
    wo, wh = F.World(oracle_lib(), F.default_config(bits)), F.World(hip_lib(), F.default_config(bits))
    oo, oh = wo.contact_manifolds(**pairs), wh.contact_manifolds(**pairs)
    assert int((oh["point_count"] > 0).sum()) > n // 4
    for k in oo:
        assert oo[k].dtype == oh[k].dtype and np.array_equal(oo[k], oh[k], equal_nan=True), f"{k} differs from the oracle"
    tol = 2e-5 if bits == 32 else 1e-10
    check_invariants(pairs, oh, tol)
    check_swap_symmetry(wh, pairs, oh, tol)


def test_contact_manifolds_empty_and_bad_arguments():
    wh = F.World(hip_lib(), F.default_config(32))
    out = wh.contact_manifolds(np.zeros(0, np.uint8), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 4)), np.zeros(0, np.uint8),
                               np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 4)), np.zeros(0))
    assert out["point_count"].shape == (0,)
    with pytest.raises(F.AvnError):
        wh.contact_manifolds([7], [[1, 1, 1]], [[0, 0, 0]], [[0, 0, 0, 1]], [0], [[1, 1, 1]], [[0, 0, 0]], [[0, 0, 0, 1]], [0.0])


@pytest.mark.parametrize("bits", [32, 64])
def test_hip_manifolds_against_the_independent_second_opinion(bits):
    """The device contact_manifolds query against tests/narrow_bruteforce.py (projected support-face intersection, 15-axis SAT optimality):
    the same check the oracle passes on the CPU, run on the product directly."""
    from test_narrow_second_opinion import check_projected_polygon_intersection
    w = F.World(hip_lib(), F.default_config(bits))
    for seed in (1, 2):
        check_projected_polygon_intersection(w, seed)
