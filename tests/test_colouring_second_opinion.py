"""CPU: ConstraintGraph::push_manifold / pop_manifold (reference dynamics/solver/constraint_graph.rs:163-296) restated a third time, in
plain Python sets and lists straight from the reference text, against the product's host graph and the oracle's: colour of every push, and
the colour lists INCLUDING their order after random pushes and swap_remove pops (the overflow colour is solved in list order)."""
import numpy as np
import pytest

from helpers import F, hip_lib, oracle_lib

COLORS, OVERFLOW, DYNAMIC = 24, 23, 20


class PyGraph:
    def __init__(self):
        self.body_set = [set() for _ in range(COLORS)]
        self.lists = [[] for _ in range(COLORS)]
        self.where = {}          # handle -> (colour, body1, body2)

    def push(self, h, b1, b2, s1, s2):
        c = OVERFLOW
        if not s1 and not s2:
            for i in range(DYNAMIC):
                if b1 in self.body_set[i] or b2 in self.body_set[i]:
                    continue
                self.body_set[i].update((b1, b2)); c = i
                break
        else:
            b = b1 if not s1 else b2
            for i in reversed(range(1, OVERFLOW)):
                if b in self.body_set[i]:
                    continue
                self.body_set[i].add(b); c = i
                break
        self.lists[c].append(h)
        self.where[h] = (c, b1, b2)
        return c

    def pop(self, h):
        c, b1, b2 = self.where.pop(h)
        if c != OVERFLOW:
            self.body_set[c].discard(b1); self.body_set[c].discard(b2)
        lst = self.lists[c]
        i = lst.index(h)
        lst[i] = lst[-1]      # swap_remove
        lst.pop()

    def flat(self):
        offs = np.cumsum([0] + [len(l) for l in self.lists])
        return offs, np.array([h for l in self.lists for h in l], np.uint64)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_three_statements_of_the_graph_agree(seed):
    rng = np.random.default_rng(seed)
    n_bodies = 60                      # dense enough to reach the overflow colour
    static = rng.random(n_bodies) < 0.1
    py, gh, go = PyGraph(), F.ConstraintGraph(hip_lib()), F.ConstraintGraph(oracle_lib())
    live = []
    overflowed = 0
    for h in range(4000):
        if live and rng.random() < 0.35:
            k = int(rng.integers(0, len(live)))
            victim = live[k]; live[k] = live[-1]; live.pop()
            py.pop(victim); gh.pop(victim); go.pop(victim)
            continue
        a = int(rng.integers(0, n_bodies)); b = int((a + 1 + rng.integers(0, n_bodies - 1)) % n_bodies)
        if static[a] and static[b]:
            continue
        c = py.push(h, a, b, bool(static[a]), bool(static[b]))
        assert gh.push(h, a, b, bool(static[a]), bool(static[b])) == c == go.push(h, a, b, bool(static[a]), bool(static[b])), f"colour of push {h}"
        overflowed += c == OVERFLOW
        live.append(h)
        if h % 500 == 499:
            offs, handles = py.flat()
            for g in (gh, go):
                o2, h2 = g.lists()
                assert np.array_equal(o2, offs) and np.array_equal(h2, handles), f"lists after {h + 1} operations"
    assert overflowed > 20, "the scene must exercise the overflow colour"
