#!/usr/bin/env python3
"""CPU model of k_pg_replay (device ConstraintGraph, exact swap_remove replay of one colour's op sequence).

The reference's GraphColor::manifold_handles is a Vec with push (append) and swap_remove (constraint_graph.rs:163-296):
inherently serial.  The device replays a colour's ops (ascending ContactId) with ONE wave, 64 ops at a time:
  * heights h_t by prefix sum; lo = lowest list position the batch's pushes / vacates touch;
  * a pop whose element sits at a position >= lo may interact with the moving tail -> the batch is cut in front of it and
    that op runs alone, serially;
  * otherwise every pop's filler is either the list entry at h_t - 1 as of the batch start, or the element of the latest
    earlier push of the batch at that height (stack matching), and all lanes apply their op independently.
This script checks that rule against the plain serial semantics on random sequences (also adversarial: tiny lists)."""
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


def batched(lst, ops, B=64, stats=None):
    mem = dict(enumerate(lst))          # global memory image of the list (entries beyond len are stale garbage, as on device)
    pos = {x: i for i, x in enumerate(lst)}
    L = len(lst)
    cur = 0
    while cur < len(ops):
        batch = ops[cur:cur + B]
        n = len(batch)
        # heights before each op
        h = []
        hh = L
        for kind, _ in batch:
            h.append(hh); hh += 1 if kind == 1 else -1
        P = [pos[x] if kind == 0 else None for kind, x in batch]       # loaded at batch start
        # conflict-free prefix: f = first pop whose position >= lo of the prefix that includes it
        f = n
        lo = 1 << 60
        for t, (kind, x) in enumerate(batch):
            lo = min(lo, h[t] if kind == 1 else h[t] - 1)
            # all pops of the prefix [0..t] must sit below lo(prefix)
            if any(batch[s][0] == 0 and P[s] >= lo for s in range(t + 1)):
                f = t
                break
        if f == 0:   # single op, serially
            kind, x = batch[0]
            if kind == 1:
                mem[L] = x; pos[x] = L; L += 1
            else:
                p = pos.pop(x); last = mem[L - 1]
                if p != L - 1:
                    mem[p] = last; pos[last] = p
                L -= 1
            cur += 1
            if stats is not None: stats["serial"] += 1
            continue
        n = f
        batch = batch[:n]
        if stats is not None: stats["fast"] += 1
        # fast path: every lane independently
        match = [None] * n
        for t in range(n):
            if batch[t][0] == 0:
                q = h[t] - 1
                for s in range(t):
                    if batch[s][0] == 1 and h[s] == q:
                        match[t] = s
        consumed = [False] * n
        for t in range(n):
            if match[t] is not None: consumed[match[t]] = True
        writes_mem = {}
        writes_pos = {}
        dels = []
        for t, (kind, x) in enumerate(batch):
            if kind == 0:
                y = batch[match[t]][1] if match[t] is not None else mem[h[t] - 1]
                assert y != x
                writes_mem[P[t]] = y; writes_pos[y] = P[t]; dels.append(x)
            elif not consumed[t]:
                assert h[t] not in writes_mem
                writes_mem[h[t]] = x; writes_pos.setdefault(x, h[t])
        # a pushed-and-consumed element got its position from the pop lane; unconsumed pushes from their own lane
        for t, (kind, x) in enumerate(batch):
            if kind == 1 and not consumed[t]: writes_pos[x] = h[t]
        mem.update(writes_mem); pos.update(writes_pos)
        for x in dels: pos.pop(x, None)
        L = h[n - 1] + (1 if batch[n - 1][0] == 1 else -1)
        cur += n
    return [mem[i] for i in range(L)]


def trial(rng, n, k, p_push):
    lst = list(range(n))
    alive = set(lst)
    nxt = n
    ops = []
    pushed = set()
    for _ in range(k):
        cand = alive - pushed
        if cand and (rng.random() > p_push):
            x = rng.choice(tuple(cand)); alive.discard(x); ops.append((0, x))
        else:
            ops.append((1, nxt)); alive.add(nxt); pushed.add(nxt); nxt += 1
    return lst, ops


def main():
    rng = random.Random(1)
    stats = {"fast": 0, "serial": 0}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
        n = rng.choice([1, 2, 5, 40, 300, 3000]); k = rng.choice([1, 10, 64, 65, 200, 1000]); pp = rng.choice([0.0, 0.2, 0.5, 0.8])
        k = min(k, 4 * n + 50)
        lst, ops = trial(rng, n, k, pp)
        a = serial(lst, ops); b = batched(lst, ops, 64, stats)
        assert a == b, (n, k, pp)
    print("ok", stats)


if __name__ == "__main__":
    main()
