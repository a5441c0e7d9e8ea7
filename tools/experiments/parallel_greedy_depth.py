#!/usr/bin/env python3
"""Feasibility number for the next round's device constraint graph.

The reference colours manifolds with a SERIAL greedy pass in processing order (constraint_graph.rs:163-236: lowest colour of 0..19 whose body bitsets
contain neither body for dynamic pairs, highest free colour of 22..1 for a pair with a static body).  A parallel pass reproduces it exactly if a manifold is coloured only after every
EARLIER manifold that shares a dynamic body with it: rounds of "all manifolds whose predecessors are done".  The number of
rounds is the longest chain in that dependency DAG.  This script measures it (CPU, numpy) for the bench's cfg2 contact set in
broad-phase emission order, checks that the round-parallel colouring equals the serial one, and prints the round sizes.

usage: python tools/experiments/parallel_greedy_depth.py [nx ny nz]      (default 50 40 50 = cfg2)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from avian_amd import _ffi as F, scenes
from helpers import oracle_lib   # CPU run: the oracle's broad phase provides the emission order, its graph the serial colours

COLORS, OVERFLOW, DYNAMIC_COLORS = 24, 23, 20


def main():
    nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 40, 50)
    lib = oracle_lib()
    sc = scenes.box_stack(nx, ny, nz)
    w = F.World(lib, F.default_config(32))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get()
    b1, b2 = p["body1"].astype(np.int64), p["body2"].astype(np.int64)
    M, N = len(b1), sc.n
    dyn = np.asarray(sc.rb_type) == F.RB_DYNAMIC
    print(f"{N - 1} boxes, {M} pairs in emission order")
    # serial greedy (the library's host ConstraintGraph)
    t0 = time.time()
    offs, perm = scenes.color_manifolds(lib, {"body1": b1.astype(np.int32), "body2": b2.astype(np.int32)}, sc.rb_type)
    serial = np.empty(M, np.int64)
    for c in range(COLORS):
        serial[perm[offs[c]:offs[c + 1]]] = c
    print(f"serial greedy: {time.time() - t0:.2f} s, colours used {int((np.diff(offs) > 0).sum())}, overflow {int(offs[24] - offs[23])}")
    # dependency levels: level(m) = 1 + max(level of the previous manifold on body1, on body2), dynamic bodies only
    t0 = time.time()
    level = np.zeros(M, np.int64)
    last_level = np.zeros(N, np.int64)
    for m in range(M):                      # (a scan over the emission order; the DEVICE version iterates rounds instead)
        a, b = b1[m], b2[m]
        l = 1 + max(last_level[a] if dyn[a] else 0, last_level[b] if dyn[b] else 0)
        level[m] = l
        if dyn[a]: last_level[a] = l
        if dyn[b]: last_level[b] = l
    rounds = int(level.max())
    sizes = np.bincount(level)[1:]
    print(f"dependency DAG: {rounds} rounds ({time.time() - t0:.1f} s); round sizes: first 5 {sizes[:5].tolist()}, median {int(np.median(sizes))}, last 5 {sizes[-5:].tolist()}")
    # round-parallel colouring: inside a round no two manifolds share a dynamic body, so each one only reads its bodies' masks
    used = np.zeros(N, np.uint32)           # bit c = body already has a manifold of colour c (colours 0..22; overflow is not recorded)
    par = np.empty(M, np.int64)
    order = np.argsort(level, kind="stable")
    starts = np.concatenate([[0], np.cumsum(sizes)])
    for r in range(rounds):
        ms = order[starts[r]:starts[r + 1]]
        a, b = b1[ms], b2[ms]
        both = dyn[a] & dyn[b]
        # dynamic-vs-dynamic: lowest free colour of 0..19; with one static body: highest free colour of 22..1, only the non-static body
        # is looked at and marked (constraint_graph.rs:163-236); static-vs-static: overflow
        mask = np.where(dyn[a], used[a], np.uint32(0)) | np.where(dyn[b], used[b], np.uint32(0))
        free_dd = ~mask & np.uint32((1 << DYNAMIC_COLORS) - 1)
        free_st = ~mask & np.uint32(((1 << OVERFLOW) - 1) & ~1)
        low = np.zeros(len(ms), np.int64); high = np.zeros(len(ms), np.int64)
        nz = free_dd != 0
        low[nz] = np.log2((free_dd[nz] & (~free_dd[nz] + np.uint32(1))).astype(np.float64)).astype(np.int64)
        nz2 = free_st != 0
        high[nz2] = np.floor(np.log2(free_st[nz2].astype(np.float64))).astype(np.int64)
        col = np.where(both, np.where(nz, low, OVERFLOW), np.where(dyn[a] | dyn[b], np.where(nz2, high, OVERFLOW), OVERFLOW))
        par[ms] = col
        bit = np.where(col < OVERFLOW, np.uint32(1) << col.astype(np.uint32), np.uint32(0))
        np.bitwise_or.at(used, a[dyn[a]], bit[dyn[a]])
        np.bitwise_or.at(used, b[dyn[b]], bit[dyn[b]])
    same = bool(np.array_equal(par, serial))
    print(f"round-parallel colouring == serial greedy: {same}")
    if not same:
        bad = np.flatnonzero(par != serial)
        print("first mismatches:", bad[:5], par[bad[:5]], serial[bad[:5]])
        sys.exit(1)


if __name__ == "__main__":
    main()
