#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors_*.npz: input/output vectors of the CPU oracle on small seeded worlds.

The reference (pure Rust) cannot be built or imported in this image, so these vectors are NOT reference outputs:
they freeze the oracle's own results (after it passed the reference KATs in reference_kats.json) so that
  * `-m "not gpu"` tests detect any drift of the oracle, and
  * `-m gpu` tests can check the HIP path against committed numbers without executing the oracle.
Regenerate with:  python tests/golden/make_oracle_vectors.py   (deterministic: numpy PCG64 seeds below).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import F, color_and_upload, oracle_lib, random_joints, random_world  # noqa: E402

from avian_amd import scenes  # noqa: E402


def solver_case(bits, seed):
    wd = random_world(seed=seed, n_bodies=96, n_manifolds=260, n_joints=40, hub_degree=26)
    w = F.World(oracle_lib(), F.default_config(bits, substeps=3))
    color_and_upload(w, oracle_lib(), wd)
    for _ in range(3):
        w.step()
    out = {}
    for name, d in (("bodies", w.bodies_download()), ("impulses", w.impulses_download()), ("joints", w.joints_download())):
        for k, v in d.items():
            out[f"{name}.{k}"] = v
    return out


def joints_case(bits):
    """All five XPBD joint types (fixed / revolute / spherical / prismatic / distance) + contacts, 2 steps."""
    wd = random_world(seed=77, n_bodies=80, n_manifolds=150, n_joints=0)
    wd["joints_generic"] = random_joints(np.random.default_rng(77), 80, 90)
    w = F.World(oracle_lib(), F.default_config(bits, substeps=3))
    color_and_upload(w, oracle_lib(), wd)
    for _ in range(2):
        w.step()
    out = {}
    for name, d in (("gj.bodies", w.bodies_download()), ("gj.joints", w.joints_download())):
        for k, v in d.items():
            out[f"{name}.{k}"] = v
    return out


def broadphase_case(bits):
    sc = scenes.box_stack(6, 4, 5)
    w = F.World(oracle_lib(), F.default_config(bits, substeps=2))
    w.bodies_upload(**sc.body_kwargs())
    w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB")
    w.run_system("COLLECT_COLLISION_PAIRS")
    mn, mx, ents = w.aabbs_download()
    p = w.pairs_get()
    return {"bp.aabb_min": mn, "bp.aabb_max": mx, "bp.interval_entities": ents,
            "bp.pairs": np.stack([p["collider1"], p["collider2"], p["body1"].astype(np.uint32), p["body2"].astype(np.uint32), p["flags"]], axis=1)}


def main():
    for bits in (32, 64):
        out = {}
        out.update(solver_case(bits, seed=2026))
        out.update(broadphase_case(bits))
        out.update(joints_case(bits))
        sys.path.insert(0, os.path.dirname(HERE))
        import golden_checks  # the narrow-phase / closed-loop case lives there (one definition for generator and checkers)
        out.update(golden_checks.narrow_vectors(oracle_lib(), bits))
        path = os.path.join(HERE, f"oracle_vectors_f{bits}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes", len(out), "arrays")


if __name__ == "__main__":
    main()
