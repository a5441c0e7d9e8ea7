"""CPU: the f32 tolerance of "identical inputs through the reference", MEASURED (SURVEY.md §8 / VERDICT N1).

The HIP kernels match the baseline oracle bit for bit; what that baseline is worth against a real Avian build depends on three things the
reference leaves to its platform: sin/cos from the platform libm (crates/avian3d/Cargo.toml:38-44, `enhanced-determinism` is opt-in and
tests/mod.rs:151-183 only asserts run-to-run equality on ONE machine), the association of glam's f32 quaternion product (SSE2 pairwise vs
scalar left-to-right), and FMA contraction.  tools/measure_tolerance.py runs the closed loop through each variant; this test asserts the
envelope on two small scenes (the full-size numbers are in DESIGN.md §2, profiles/r02_tolerance_*.json)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from measure_tolerance import VARIANTS, measure  # noqa: E402

# (max |dx| m, |dv| m/s, |dw| rad/s, |dq|) allowed against the baseline after N steps: one rounding difference is ~1e-9 after the first
# step; contacts that open or close one step earlier amplify it to millimetres within ten steps (the solver is a chaotic map: the same
# order of magnitude whichever of the three sources perturbs it), and the piles stay piles (centimetres after 100 steps).
BOUNDS = {1: (1e-7, 1e-5, 1e-5, 1e-7), 10: (2e-2, 0.25, 0.4, 1e-2), 100: (0.15, 1.0, 2.0, 0.1)}


@pytest.mark.parametrize("spec", ["large_pyramid:20", "stack:8,6,8"])
def test_variant_drift_stays_inside_the_stated_tolerance(spec):
    res = measure(spec, sorted(BOUNDS))
    for v in VARIANTS:
        for steps, bound in BOUNDS.items():
            got = res[v][steps]
            assert all(g <= b for g, b in zip(got, bound)), f"{spec} {v} after {steps} steps: {got} exceeds {bound}"
    # the perturbations are real: the FMA build and the scalar quaternion product do change bits in the very first step ...
    assert max(res["fma"][1]) > 0.0 and max(res["scalar_quat"][1]) > 0.0
    # ... while sin/cos only sees the tiny per-substep rotation angles: polynomial and host libm agree to the last bit there
    assert res["libm_trig"][10][0] <= 1e-6
