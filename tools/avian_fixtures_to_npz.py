#!/usr/bin/env python3
"""integration/rust/avian_fixtures writes <scene>.avf chunk streams; this turns them into tests/golden/avian/<scene>.npz, what tests/test_reference_fixtures.py reads.
usage: python tools/avian_fixtures_to_npz.py <dir with .avf files> [tests/golden/avian]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from avian_fixture_format import avf_to_npz  # noqa: E402


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "tests", "golden", "avian")
    os.makedirs(dst, exist_ok=True)
    for name in sorted(os.listdir(src)):
        if name.endswith(".avf"):
            out = os.path.join(dst, name[:-4] + ".npz")
            avf_to_npz(os.path.join(src, name), out)
            print(f"{name} -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
