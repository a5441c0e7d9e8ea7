#!/usr/bin/env python3
"""CPU model of k_pg_replay_wide: one colour's push / swap_remove sequence replayed B ops at a time by a whole workgroup.

Per batch (ops t = 0 .. n-1 of the colour, list length L at its start, H_t = length after op t):
  * a pop at t vacates level q = H_t.  Its filler is what sits at level q at that time: the element of the push at s = 1 + (largest
    u < t with H_u <= q) (heights move by +-1: that op is the push that last went up through q); if no such u: op 0 if L <= q, else
    the list entry at q as of the batch start.  (min-tree over H, one descent per pop.)
  * a filler taken from the list may be popped LATER IN THE SAME BATCH (at B = 1024 this happens a dozen times per batch): that pop
    finds it in the hole it was moved to.  pred[j] = the pop whose filler op j pops; P_j = P_pred (pointer jumping through chains), and the
    placement of a filler that is popped later in the batch is never written (its slot is overwritten by the later pop's filler).
  * the batch is exact while every pop's (resolved) position is below the lowest level the batch touches up to and including that op; the
    first op that breaks this runs alone, serially, and the batch restarts behind it.
Checked against the plain serial semantics on random and adversarial (tiny lists) sequences."""
import random
import sys


def serial(lst, ops):
    lst = list(lst)
    pos = {x: i for i, x in enumerate(lst)}
    for kind, x in ops:
        if kind == 1:
            pos[x] = len(lst); lst.append(x)
        else:
            p = pos.pop(x); last = lst[-1]
            lst[p] = last
            lst.pop()
            if last != x:
                pos[last] = p
    return lst


def wide(lst, ops, B=1024, stats=None):
    mem = dict(enumerate(lst))   # global image of the list (entries beyond the length are stale, as on the device)
    pos = {x: i for i, x in enumerate(lst)}
    L = len(lst)
    cur = 0
    while cur < len(ops):
        n = min(B, len(ops) - cur)
        while True:   # (redo with a shorter n after a cut)
            batch = ops[cur:cur + n]
            H = []
            hh = L
            for kind, _ in batch:
                hh += 1 if kind == 1 else -1
                H.append(hh)
            h = [H[t] - (1 if batch[t][0] == 1 else -1) for t in range(n)]   # length before op t
            P0 = [pos[x] if kind == 0 else None for kind, x in batch]          # staged at batch start
            # fillers
            match = [None] * n; y = [None] * n
            for t, (kind, x) in enumerate(batch):
                if kind != 0: continue
                q = H[t]
                u = next((u for u in range(t - 1, -1, -1) if H[u] <= q), None)
                if u is not None: s = u + 1
                elif L <= q: s = 0
                else: s = None
                if s is not None:
                    assert batch[s][0] == 1 and h[s] == q, (s, t)
                    match[t] = s; y[t] = batch[s][1]
                else:
                    y[t] = mem[q]
            # fillers from the list that are popped later in this batch
            where = {}
            for t in range(n - 1, -1, -1):   # (the FIRST pop that moves a value wins: lanes behind a cut may hold garbage duplicates)
                if batch[t][0] == 0 and match[t] is None: where[y[t]] = t
            pred = [None] * n
            for t, (kind, x) in enumerate(batch):
                if kind == 0 and x in where and where[x] < t: pred[t] = where[x]
            P = list(P0)
            for t in range(n):   # (pointer jumping on the device)
                if pred[t] is not None:
                    r = t
                    while pred[r] is not None: r = pred[r]
                    P[t] = P0[r]
            dead = [False] * n
            for t in range(n):
                if pred[t] is not None: dead[pred[t]] = True
            # conflict-free prefix (with the RESOLVED positions; a pop whose element is a filler moved in this batch is judged by the hole)
            f = n
            lo = 1 << 60; pmax = -1
            for t, (kind, x) in enumerate(batch):
                lo = min(lo, h[t] if kind == 1 else h[t] - 1)
                if kind == 0: pmax = max(pmax, P[t] + 1)
                if pmax > lo:
                    f = t; break
            if f == n: break
            if f == 0: break
            n = f
            if stats is not None: stats["cuts"] = stats.get("cuts", 0) + 1
        if f == 0:   # the first op alone, serially
            kind, x = batch[0]
            if kind == 1:
                mem[L] = x; pos[x] = L; L += 1
            else:
                p = pos.pop(x); last = mem[L - 1]
                if p != L - 1:
                    mem[p] = last; pos[last] = p
                L -= 1
            cur += 1
            if stats is not None: stats["serial"] = stats.get("serial", 0) + 1
            continue
        consumed = [False] * n
        for t in range(n):
            if batch[t][0] == 0 and match[t] is not None: consumed[match[t]] = True
        writes = {}
        for t, (kind, x) in enumerate(batch):
            if kind == 0:
                pos.pop(x, None)
                if not dead[t]:
                    assert P[t] not in writes
                    writes[P[t]] = y[t]; pos[y[t]] = P[t]
            elif not consumed[t]:
                assert h[t] not in writes
                writes[h[t]] = x; pos[x] = h[t]
        mem.update(writes)
        L = H[n - 1]
        cur += n
        if stats is not None: stats["batches"] = stats.get("batches", 0) + 1
    return [mem[i] for i in range(L)]


def random_case(rng, n0, n_ops, p_push):
    lst = list(range(n0))
    live = set(lst); nxt = n0
    ops = []
    touched = set()
    for _ in range(n_ops):
        cand = [x for x in live if x not in touched] if rng.random() >= p_push else None
        if cand:
            x = rng.choice(cand); live.discard(x); touched.add(x); ops.append((0, x))
        else:
            x = nxt; nxt += 1; live.add(x); touched.add(x); ops.append((1, x))
    return lst, ops


def main():
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    stats = {}
    cases = 0
    for n0, n_ops, pp, B in [(0, 40, 0.7, 16), (3, 60, 0.5, 16), (10, 200, 0.5, 64), (40, 300, 0.45, 64), (200, 600, 0.5, 256), (2000, 3000, 0.5, 1024),
                             (2000, 1500, 0.3, 1024), (300, 1200, 0.6, 1024), (5, 30, 0.4, 8), (64, 64, 0.2, 32)]:
        for _ in range(300 if n0 < 500 else 30):
            lst, ops = random_case(rng, n0, n_ops, pp)
            a = serial(lst, ops); b = wide(lst, ops, B, stats)
            assert a == b, (n0, n_ops, pp, B)
            cases += 1
    print("ok", cases, "cases", stats)


if __name__ == "__main__":
    main()
