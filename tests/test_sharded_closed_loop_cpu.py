"""CPU: the closed loop SHARDED BY ISLANDS with replicated integer bookkeeping (avian_amd/shard.py: ShardedClosedLoop) on the oracle backend:
two (and three) ranks, each running only its own islands' broad phase / narrow phase / solver, stay bit-identical to the single world through
ContactId reuse (lowest free id first, data_structures/id_pool.rs:31-40) and swap_removes that move ANOTHER rank's handle
(dynamics/solver/constraint_graph.rs:245-296) -- in one process (payloads handed around) and over gloo with world_size 2."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from avian_amd import shard
from helpers import F, REPO, oracle_lib
from pipeline_scenes import dropped_boxes


def piles(n_piles=2, n=24, gap=30.0, seed=50):
    """`n_piles` tumbling piles (boxes and balls) on ONE static ground, `gap` metres apart: every pile is a set of islands of its own."""
    parts = [dropped_boxes(seed=seed + k, n=n) for k in range(n_piles)]
    bodies = {k: [np.asarray(parts[0][0][k])[:1]] for k in parts[0][0]}
    colliders = {k: [np.asarray(parts[0][1][k])[:1]] for k in parts[0][1] if k not in ("entity_index", "body")}
    for k, (b, c) in enumerate(parts):
        pos = np.asarray(b["position"])[1:].copy(); pos[:, 0] += (k - (n_piles - 1) / 2) * gap
        for key in b:
            bodies[key].append(pos if key == "position" else np.asarray(b[key])[1:])
        for key in colliders:
            colliders[key].append(np.asarray(c[key])[1:])
    bodies = {k: np.concatenate(v) for k, v in bodies.items()}
    colliders = {k: np.concatenate(v) for k, v in colliders.items()}
    m = len(bodies["inv_mass"])
    colliders["entity_index"] = (np.arange(m, dtype=np.uint32) + 100)
    colliders["body"] = np.arange(m, dtype=np.int32)
    return bodies, colliders


def single_world(lib, bits, bodies, colliders):
    w = F.World(lib, F.default_config(bits, substeps=4))
    w.bodies_upload(**bodies); w.colliders_upload(**colliders); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable()
    return w


def plan_by_pile(bodies, n_piles, n):
    """rank r owns pile r (the partitioner's job in a real run: shard.plan; here the piles ARE the ranks' islands)."""
    m = len(bodies["inv_mass"])
    rank = np.full(m, -1, np.int32)
    for k in range(n_piles):
        rank[1 + k * n:1 + (k + 1) * n] = k
    return shard.ShardPlan(n_piles, rank.copy(), rank, n_piles)


def compare(step, ref, ranks):
    off, handles = ref.pipeline_handles()
    for w, loop, loc in ranks:
        goff, gh = loop.global_lists
        assert np.array_equal(goff, off) and np.array_equal(gh, handles), f"step {step}: rank {loop.rank}'s replicated colour lists differ from the single world's"
        br, bl = ref.bodies_download(), w.bodies_download()
        mine = loop.plan.rank_of_body[loc] == loop.rank
        for k in br:
            assert np.array_equal(br[k][loc[mine]], bl[k][mine]), f"step {step}: rank {loop.rank}: bodies.{k}"


@pytest.mark.parametrize("bits,n_piles", [(32, 2), (64, 2), (32, 3)])
def test_sharded_closed_loop_in_one_process_equals_the_single_world(bits, n_piles):
    lib = oracle_lib()
    n = 24
    bodies, colliders = piles(n_piles, n)
    ref = single_world(lib, bits, bodies, colliders)
    p = plan_by_pile(bodies, n_piles, n)
    ranks = shard.sharded_closed_loop_worlds(lib, bits, bodies, colliders, p)
    loops = [r[1] for r in ranks]
    for s in range(90):
        ref.step()
        shard.step_in_process(loops)
        compare(s, ref, ranks)
    st = ref.pipeline_stats()
    l0 = loops[0]
    assert st.pairs_added == l0.stats["pairs_added"] and st.pairs_removed == l0.stats["pairs_removed"] > 0
    assert st.manifolds_pushed == l0.stats["pushes"] and st.manifolds_popped == l0.stats["pops"] > 0
    assert max(l0.pairs) < st.pairs_added, "ContactIds were freed and handed out again"
    # the ranks' colour lists interleave: a pop on one rank moves handles of the other (what a per-rank ConstraintGraph could not reproduce)
    off, handles = ref.pipeline_handles()
    owner = np.array([l0.pairs[int(h)][4] for h in handles])
    mixed = sum(1 for c in range(F.GRAPH_COLOR_COUNT) if len(set(owner[off[c]:off[c + 1]].tolist())) > 1)
    assert mixed >= 3


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_sharded_closed_loop_over_gloo_world_size_2(tmp_path):
    out = str(tmp_path / "sharded_cl.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "tests", "sharded_closed_loop_worker.py"), out, "60"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, AVN_SHARD_BACKEND="oracle", OMP_NUM_THREADS="1"), cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    lib = oracle_lib()
    bodies, colliders = piles(2, 24)
    ref = single_world(lib, 32, bodies, colliders)
    for s in range(60):
        ref.step()
    b = ref.bodies_download()
    for k in b:
        assert np.array_equal(got[k], b[k]), f"{k}: the two-rank run differs from the single world after 60 steps"
    off, handles = ref.pipeline_handles()
    assert np.array_equal(got["offsets"], off) and np.array_equal(got["handles"], handles)


# ---- round 5: the replicated bookkeeping in the library (avn_shard_*, host C++) ---------------------------------------------------------------------
@pytest.mark.parametrize("bits,n_piles", [(32, 2), (64, 3)])
def test_native_sharded_closed_loop_in_one_process_equals_the_single_world(bits, n_piles):
    """ShardedClosedLoopNative: flat-array payloads, the integer work in avo_shard_* (the oracle's restatement: ordered sets and maps, a literal insertion
    sort) -- bit-identical to the single world through id reuse and foreign swap_removes, like the Python model above."""
    lib = oracle_lib()
    n = 24
    bodies, colliders = piles(n_piles, n)
    ref = single_world(lib, bits, bodies, colliders)
    p = plan_by_pile(bodies, n_piles, n)
    ranks = shard.sharded_closed_loop_worlds(lib, bits, bodies, colliders, p, native=True)
    loops = [r[1] for r in ranks]
    for s in range(90):
        ref.step()
        shard.step_in_process_native(loops)
        compare(s, ref, ranks)
    st, sh = ref.pipeline_stats(), loops[0].shard.stats()
    assert (st.pairs_added, st.pairs_removed, st.manifolds_pushed, st.manifolds_popped) == (sh.pairs_added, sh.pairs_removed, sh.pushes, sh.pops)
    assert sh.pairs_removed > 0 and sh.next_id < sh.pairs_added, "ContactIds were freed and handed out again"


def test_product_shard_bookkeeping_equals_the_oracles_on_the_same_streams():
    """avn_shard_* (avian_amd/csrc/avn_shard.cpp: vectors, a heap, std::stable_sort; host only -- it runs without a device) against avo_shard_* on the
    payload streams of a three-rank run: after every phase the ids handed out, the active lists, the removals and both handle lists must be equal."""
    import avian_amd
    lib = oracle_lib()
    plib = avian_amd.load_library()          # (only the host-side bookkeeping entry points are called: no world is created on it)
    n_piles, n = 3, 24
    bodies, colliders = piles(n_piles, n)
    p = plan_by_pile(bodies, n_piles, n)
    ranks = shard.sharded_closed_loop_worlds(lib, 32, bodies, colliders, p, native=True)
    loops = [r[1] for r in ranks]
    mirrors = [F.Shard(plib, colliders["entity_index"], r) for r in range(n_piles)]
    total_removed = 0
    for s in range(70):
        p1 = [l.phase1() for l in loops]
        kc, kx, pr = (np.concatenate([x[i] for x in p1]) for i in range(3))
        ch = []
        for l, m in zip(loops, mirrors):
            got = m.phase2(kc, kx, pr)
            ch.append(l.phase2(kc, kx, pr))
            # (l.phase2 ran avo_shard_phase2 inside: its last results are still readable)
            a, b, c, d, cnt = F.vp(), F.vp(), F.vp(), F.vp(), F.C.c_size_t()
            assert lib.fn("shard_new_local_pairs")(l.shard.handle, F.C.byref(a), F.C.byref(b), F.C.byref(c), F.C.byref(d), F.C.byref(cnt)) == 0
            want = tuple(F.Shard._u32(x, cnt.value) for x in (a, b, c, d))
            for x, y in zip(got, want):
                assert np.array_equal(x, y), f"step {s}: rank {l.rank}: new local pairs differ"
            assert np.array_equal(m.active(), l.shard.active()), f"step {s}: rank {l.rank}: active lists differ"
        ch = np.concatenate(ch)
        for l, m in zip(loops, mirrors):
            rem = m.phase3(ch)
            l.phase3(ch)
            ra, cnt = F.vp(), F.C.c_size_t()
            assert lib.fn("shard_removed_local")(l.shard.handle, F.C.byref(ra), F.C.byref(cnt)) == 0
            assert np.array_equal(rem, F.Shard._u32(ra, cnt.value)), f"step {s}: rank {l.rank}: removals differ"
            total_removed += len(rem)
            for g in (False, True):
                (o1, h1), (o2, h2) = m.handles(g), l.shard.handles(g)
                assert np.array_equal(o1, o2) and np.array_equal(h1, h2), f"step {s}: rank {l.rank}: {'global' if g else 'local'} handle lists differ"
    assert total_removed > 10


def test_native_sharded_closed_loop_many_pyramids_over_2_and_4_ranks():
    """The reference's Many Pyramids scene at a size the CPU runs in seconds (4 ground plates x 4 pyramids of base 5 = 240 boxes, 16 islands) over
    2 and 4 ranks by whole pyramids, 40 steps each, against the single world every step."""
    from avian_amd import scenes
    lib = oracle_lib()
    base, rows, cols = 5, 4, 4
    sc = scenes.many_pyramids(base, rows, cols)
    bodies, colliders = sc.body_kwargs(), sc.collider_kwargs()
    per = base * (base + 1) // 2
    pyramid = (np.arange(sc.n) - rows) // per          # (bodies 0..rows-1 are the static ground plates)
    for world in (2, 4):
        rank = np.where(np.arange(sc.n) < rows, -1, pyramid * world // (rows * cols)).astype(np.int32)
        p = shard.ShardPlan(world, rank.copy(), rank, rows * cols)
        single = single_world(lib, 32, bodies, colliders)
        ranks = shard.sharded_closed_loop_worlds(lib, 32, bodies, colliders, p, native=True)
        loops = [r[1] for r in ranks]
        for s in range(40):
            single.step()
            shard.step_in_process_native(loops)
            compare(s, single, ranks)
        assert single.pipeline_stats().manifolds > 300


def test_native_sharded_closed_loop_over_gloo_with_flat_tensors(tmp_path):
    out = str(tmp_path / "sharded_cl_native.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(REPO, "tests", "sharded_closed_loop_worker.py"), out, "60", "native"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, AVN_SHARD_BACKEND="oracle", OMP_NUM_THREADS="1"), cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    lib = oracle_lib()
    bodies, colliders = piles(2, 24)
    ref = single_world(lib, 32, bodies, colliders)
    for s in range(60):
        ref.step()
    b = ref.bodies_download()
    for k in b:
        assert np.array_equal(got[k], b[k]), f"{k}: the two-rank run (library bookkeeping, tensor all-gathers) differs from the single world after 60 steps"
    off, handles = ref.pipeline_handles()
    assert np.array_equal(got["offsets"], off) and np.array_equal(got["handles"], handles)
