#!/usr/bin/env python3
"""Debugging aid for the device ConstraintGraph (k_graph.hip): runs a scene through the HIP closed loop with AVN_PG_DUMP set and
through the oracle, replays every step's dumped ops serially in Python from the ORACLE's previous lists and reports the first op
whose colour or whose list effect differs.  usage: python tools/debug_pg.py [n_boxes] [steps] [seed]"""
import os
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
dump = tempfile.mkdtemp(prefix="pgdump")
os.environ["AVN_PG_DUMP"] = dump
import avian_amd  # noqa: E402
from avian_amd import _ffi as F  # noqa: E402
from helpers import oracle_lib  # noqa: E402
from pipeline_scenes import dropped_boxes  # noqa: E402
from test_pipeline_cpu import make  # noqa: E402


def read_dump(step):
    p = os.path.join(dump, "step_%04d.bin" % step)
    if not os.path.exists(p):
        return None
    raw = np.fromfile(p, np.uint32)
    n = int(raw[0]); o = 4
    cid = raw[o:o + n]; o += n
    info = raw[o:o + n]; o += n
    bodies = raw[o:o + 2 * n].view(np.int32).reshape(n, 2); o += 2 * n
    order = raw[o:o + n]; o += n
    cnt = raw[o:o + 32]
    return cid, info, bodies, order, cnt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 21
    bodies, colliders = dropped_boxes(seed=seed, n=n)
    wo, _ = make(oracle_lib(), 32, bodies, colliders); wo.pipeline_enable()
    wh, _ = make(avian_amd.load_library(), 32, bodies, colliders); wh.pipeline_enable()
    nb = len(bodies["position"])
    masks = np.zeros(nb, np.uint32)     # per-body colour masks implied by the oracle's lists
    prev = (np.zeros(25, np.uint32), np.zeros(0, np.uint32))
    pair_bodies = {}
    for s in range(steps):
        wo.step(); wh.step()
        oo, oh = wo.pipeline_handles(), wh.pipeline_handles()
        d = read_dump(s + 1)
        if d is not None:
            cid, info, bod, order, cnt = d
            kind = info & 3; s1 = (info >> 2) & 1; s2 = (info >> 3) & 1; col = (info >> 8) & 0xFF
            assert np.all(np.diff(cid.astype(np.int64)) > 0), "ops not in ascending contact id"
            lists = [list(prev[1][prev[0][c]:prev[0][c + 1]]) for c in range(24)]
            where = {int(x): c for c in range(24) for x in lists[c]}
            for k in range(len(cid)):
                x = int(cid[k]); b1, b2 = int(bod[k, 0]), int(bod[k, 1])
                pair_bodies[x] = (b1, b2, int(s1[k]), int(s2[k]))
                if kind[k] == 1:
                    if not s1[k] and not s2[k]:
                        free = ~(masks[b1] | masks[b2]) & 0xFFFFF
                        c = (int(free) & -int(free)).bit_length() - 1 if free else 23
                    else:
                        m = masks[b2] if s1[k] else masks[b1]
                        free = int(~m) & 0x7FFFFE
                        c = free.bit_length() - 1 if free else 23
                    if c != col[k]:
                        print(f"step {s}: op {k} cid {x} bodies {b1},{b2} static {s1[k]}{s2[k]}: device colour {col[k]}, serial {c}; masks {masks[b1]:06x} {masks[b2]:06x}")
                        return 1
                    if c < 23:
                        if not s1[k]: masks[b1] |= np.uint32(1 << c)
                        if not s2[k]: masks[b2] |= np.uint32(1 << c)
                    lists[c].append(x); where[x] = c
                elif kind[k] == 2:
                    c = where.pop(x)
                    if c != col[k]:
                        print(f"step {s}: pop op {k} cid {x}: device colour {col[k]}, serial {c}"); return 1
                    if c < 23:
                        masks[b1] &= ~np.uint32(1 << c); masks[b2] &= ~np.uint32(1 << c)
                    i = lists[c].index(x); lists[c][i] = lists[c][-1]; lists[c].pop()
            for c in range(24):
                dv = list(oh[1][oh[0][c]:oh[0][c + 1]]); orc = list(oo[1][oo[0][c]:oo[0][c + 1]])
                if lists[c] != orc:
                    print(f"step {s}: python replay of the DEVICE's ops != oracle list, colour {c}: ops differ from the oracle's ({len(lists[c])} vs {len(orc)})"); return 1
                if dv != orc:
                    ops_c = [(int(kind[k]), int(cid[k])) for k in range(len(cid)) if col[k] == c and kind[k]]
                    print(f"step {s}: colour {c}: device list != oracle list although the ops agree\n prev   {list(prev[1][prev[0][c]:prev[0][c + 1]])}\n ops    {ops_c}\n oracle {orc}\n device {dv}")
                    print(" order bucket:", [int(v) for v in order[:len(cid)]][:80], " counts", cnt[:25])
                    return 1
        elif not (np.array_equal(oo[0], oh[0]) and np.array_equal(oo[1], oh[1])):
            print(f"step {s}: lists differ and no dump"); return 1
        prev = oo
        bo, bh = wo.bodies_download(), wh.bodies_download()
        for k in bo:
            if not np.array_equal(bo[k], bh[k]):
                print(f"step {s}: bodies.{k} differs although the lists agree (solver side: overflow pass / warm start CSR?) overflow manifolds {oo[0][24] - oo[0][23]}"); return 1
    print("all", steps, "steps agree")
    return 0


if __name__ == "__main__":
    sys.exit(main())
