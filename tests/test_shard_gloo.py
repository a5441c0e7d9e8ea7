"""N > 1 host path on CPU: world_size-2 `gloo` run of the island-sharded flow (each rank drives only its own
sub-world through the C ABI, oracle backend) must reproduce the single-world run BIT FOR BIT, and the per-step bounds
exchange must flag islands of different ranks coming into AABB contact."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import F, REPO, hip_lib, oracle_lib

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shard_worker as SW  # noqa: E402

from avian_amd import scenes, shard  # noqa: E402


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch(case, out, steps, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(REPO, "tests", "shard_worker.py"), case, out, str(steps)]
    env = dict(os.environ, AVN_SHARD_BACKEND="oracle", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]


def test_islands_partition_product_matches_checker_and_scipy():
    """Integer host work, no GPU needed: product union-find == oracle flood fill == scipy connected components."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(3)
    n, e = 4000, 1200   # below the percolation threshold: many small islands
    rb = (rng.random(n) < 0.05).astype(np.uint8) * F.RB_STATIC
    x = rng.uniform(-100, 100, n)
    e1 = rng.integers(0, n, e).astype(np.int32); e2 = rng.integers(0, n, e).astype(np.int32)
    for R in (1, 2, 4, 8):
        ip, rp, kp = hip_lib().islands_partition(rb, x, e1, e2, R)
        io, ro, ko = oracle_lib().islands_partition(rb, x, e1, e2, R)
        assert kp == ko and np.array_equal(ip, io) and np.array_equal(rp, ro)
        dyn = (rb[e1] != F.RB_STATIC) & (rb[e2] != F.RB_STATIC)
        g = coo_matrix((np.ones(dyn.sum()), (e1[dyn], e2[dyn])), shape=(n, n))
        ncomp, lab = connected_components(g, directed=False)
        nonstatic = rb != F.RB_STATIC
        assert kp == len(np.unique(lab[nonstatic]))
        # same partition: bodies share an island iff they share a scipy component
        assert len(set(zip(ip[nonstatic].tolist(), lab[nonstatic].tolist()))) == kp
        assert (ip[~nonstatic] == -1).all() and (rp[~nonstatic] == -1).all()
        # every island on exactly one rank, ranks are contiguous slabs in mean-x order, load roughly balanced
        assert rp[nonstatic].min() == 0 and rp[nonstatic].max() == R - 1
        w = np.bincount(rp[nonstatic], minlength=R)
        assert w.max() <= 1.5 * w.mean() + 50
    # empty and degenerate inputs
    i0, r0, k0 = hip_lib().islands_partition(np.zeros(0, np.uint8), np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32), 4)
    assert k0 == 0 and len(i0) == 0
    with pytest.raises(F.AvnError):
        hip_lib().islands_partition(np.zeros(3, np.uint8), np.zeros(3), np.array([7], np.int32), np.array([0], np.int32), 2)


def test_sharded_two_ranks_bit_identical_to_single_world(tmp_path):
    out = str(tmp_path / "stacks.npz")
    steps = 4
    launch("stacks", out, steps)
    got = np.load(out)
    sc, joints = SW.build_case("stacks")
    ref, ref_pairs, _ = SW.run_world(oracle_lib(), sc.body_kwargs(), sc.collider_kwargs(), joints, sc.friction, sc.restitution, steps, 2)
    assert int(got["n_islands"]) == 3 and sorted(got["owned"].tolist()) == [27, 54]
    for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
        assert np.array_equal(got[k], ref[k]), f"{k}: sharded run differs from the single world"
    assert float(np.abs(ref["linear_velocity"]).max()) > 0.05
    # pair lists: the union over ranks equals the single world's list, and each rank's list is a sub-SEQUENCE of it
    for s in range(steps):
        glob = [tuple(p) for p in ref_pairs[s]]
        seen = []
        for r in range(2):
            sub = [tuple(p) for p in got[f"pairs_r{r}_s{s}"]]
            it = iter(glob)
            assert all(p in it for p in sub), f"step {s} rank {r}: emission order is not the global order"
            seen += sub
        assert sorted(seen) == sorted(glob)
    assert int(got["first_overlap"]) == -1, "independent stacks must never trigger the proximity exchange"


def test_bounds_exchange_detects_cross_rank_approach(tmp_path):
    out = str(tmp_path / "approach.npz")
    launch("approach", out, 6)
    got = np.load(out)
    assert sorted(got["owned"].tolist()) == [8, 8]
    assert 0 <= int(got["first_overlap"]) <= 5, "the thrown body reaches the other rank's stack: bounds must overlap"


def test_thrown_body_merges_two_ranks_islands_and_the_run_stays_the_single_world(tmp_path):
    """Level 1 completed: the bounds exchange triggers shard.repartition, the two islands land on one rank together with their known pairs
    and the merged interval order, and from then on the run equals the single world bit for bit (bodies after 12 steps, through the impact)."""
    out = str(tmp_path / "merge.npz")
    steps = 12
    launch("merge", out, steps)
    got = np.load(out)
    sc, _ = SW.build_case("approach")
    ref, _, _ = SW.run_world(oracle_lib(), sc.body_kwargs(), sc.collider_kwargs(), None, sc.friction, sc.restitution, steps, 2)
    assert np.all(got["holders"][1:] == 1), "every dynamic body lives on exactly one rank at the end"
    assert len(got["changed"]) >= 1, "the thrown body must have forced a re-partition"
    oh = got["owned_hist"]
    assert oh[:, 0].tolist() == [8, 8] and sorted(oh[:, -1].tolist()) == [0, 16], "two islands of 8 became one island of 16 on one rank"
    for k in ("position", "rotation", "linear_velocity", "angular_velocity"):
        assert np.array_equal(got[k], ref[k]), f"{k}: the re-partitioned run differs from the single world"
    # the impact really happened: bodies of the right stack were moved by the thrown one
    rest, _, _ = SW.run_world(oracle_lib(), *_without_throw(sc), None, sc.friction, sc.restitution, steps, 2)
    assert float(np.abs(ref["linear_velocity"][9:] - rest["linear_velocity"][9:]).max()) > 0.1


def _without_throw(sc):
    b = sc.body_kwargs()
    b["linear_velocity"] = np.zeros_like(b["linear_velocity"])
    return b, sc.collider_kwargs()


def test_merge_interval_orders_and_cross_pairs():
    """Unit level: the k-way merge keeps each rank's own order, sorts across ranks by (key, entity) and keeps replicated colliders once."""
    mk = lambda order, key: shard.RankState(np.zeros(0, np.int64), {}, {}, np.zeros((0, 2), np.int64), np.array(order, np.int64), np.array(key, float))
    a = mk([0, 5, 3, 7], [-9.0, 1.0, 1.0, 4.0])       # 5 before 3 although tied: history, must survive
    b = mk([0, 4, 2, 6], [-9.0, 1.0, 2.0, 4.0])
    assert shard.merge_interval_orders([a, b]).tolist() == [0, 4, 5, 3, 2, 6, 7]
    fresh = mk([0, 2, 1], [np.nan] * 3)
    assert shard.merge_interval_orders([fresh, mk([0, 3], [np.nan] * 2)]).tolist() == [0, 2, 1, 3]


def test_bounds_overlap_predicate():
    mn = np.array([[0, 0, 0], [1, 1, 1], [5, 5, 5.0]]); mx = np.array([[1, 1, 1], [2, 2, 2], [6, 6, 6.0]])
    assert shard.bounds_overlap(mn, mx) == [(0, 1)]          # touching counts, like ColliderAabb::intersects
    inf = np.inf
    assert shard.bounds_overlap(np.array([[inf] * 3, [0, 0, 0.0]]), np.array([[-inf] * 3, [1, 1, 1.0]])) == []


# ---- x-slab sharding of the broad phase (cfg4's multi-GPU shape) ----------------------------------------

def single_world_pairs(lib, sc, bodies_per_step):
    """One persistent world over the frames: the reference flow (pairs already in the contact graph are not re-emitted)."""
    w = F.World(lib, F.default_config(32, substeps=1))
    w.bodies_upload(**bodies_per_step[0]); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    out = []
    for b in bodies_per_step:
        w.bodies_upload(**b)
        w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
        p = w.pairs_get()
        out.append(np.stack([p["collider1"], p["collider2"], p["flags"]], axis=1).astype(np.uint32))
    mn, mx, _ = w.aabbs_download()
    w.close()
    return out, mn, mx


@pytest.mark.parametrize("scene", ["lattice_with_ground", "sparse"])
def test_slab_sharded_broad_phase_equals_single_world(scene):
    """Every rank sweeps its x-slab + halo with the unchanged broad phase; the concatenation of the ranks' owned pairs in slab
    order must BE the single-world pair list (same pairs, same order, same flags) -- including lattices full of equal min.x
    keys and a ground slab that spans every slab."""
    lib = oracle_lib()
    sc = scenes.box_stack(6, 6, 6) if scene == "lattice_with_ground" else scenes.sparse_mixed(3000, side=20.0)
    ref, mn, mx = single_world_pairs(lib, sc, [sc.body_kwargs()])
    assert len(ref[0]) > 1000
    none = np.zeros(0, np.uint64)
    for R in (1, 2, 3, 8):
        parts = [shard.slab_broad_phase_step(lib, 32, sc.body_kwargs(), sc.collider_kwargs(), mn[:, 0], mx[:, 0], none, r, R) for r in range(R)]
        assert np.array_equal(np.concatenate(parts), ref[0]), f"{scene}: {R} slabs"
        if scene == "sparse" and R > 1:
            assert max(len(p) for p in parts) < 0.8 * len(ref[0]), "the work is actually split"
    # the plan: ties never straddle a boundary, slabs are contiguous in min.x, every collider has exactly one owner
    pl = shard.slab_plan(mn[:, 0], 4)
    assert pl.slab_of_collider.min() == 0 and pl.slab_of_collider.max() <= 3
    order = np.argsort(mn[:, 0], kind="stable")
    assert np.all(np.diff(pl.slab_of_collider[order]) >= 0)
    for v in np.unique(mn[:, 0]):
        assert len(np.unique(pl.slab_of_collider[mn[:, 0] == v])) == 1
    # degenerate inputs: no colliders, more ranks than distinct keys
    e = shard.slab_plan(np.zeros(0), 4)
    assert len(e.slab_of_collider) == 0
    loc, own = shard.slab_colliders(shard.slab_plan(np.zeros(5), 4), 2, np.zeros(5), np.ones(5))
    assert len(loc) == 0 or own.all()


def test_slab_sharded_broad_phase_keeps_the_persistent_tie_order_over_frames():
    """A lattice (many equal min.x) whose columns trade places over frames: in the single persistent world equal keys keep the
    PREVIOUS frame's relative order (stable insertion sort over the kept AabbIntervals), so collider1 / collider2 of a new pair
    depend on history.  Slab sub-worlds seeded with that order (slab_next_order) reproduce it; seeding by upload index does not
    (ADVICE round 1)."""
    lib = oracle_lib()
    sc = scenes.box_stack(6, 3, 6)
    n = sc.n
    rng = np.random.default_rng(5)
    frames = []
    pos = sc.position.copy()
    K = 12   # boxes 1 .. 2K are taken out of the lattice: K couples (A = 1 + 2k, B = 2 + 2k) far above it, 40 apart in y
    for f in range(5):
        b = sc.body_kwargs()
        if f:   # whole x-columns of the lattice jump to other lattice columns: orders change, new keys tie with columns already there
            ux, inv = np.unique(pos[1 + 2 * K:, 0], return_inverse=True)
            pos[1 + 2 * K:, 0] = rng.permutation(ux)[inv]
        for k in range(K):
            A, B = 1 + 2 * k, 2 + 2 * k
            y0 = 100.0 + 100.0 * k
            # frame 0: A right of B (sorted order B, A); frame 1+: the SAME min.x -- the persistent world keeps B before A, a fresh
            # world would put A (lower upload index) first; frame 3+: they touch, and the new pair is (B, A) only with the kept order
            xa, xb = (50.0 + k, 49.0 + k) if f == 0 else (49.5 + k, 49.5 + k)
            pos[A] = [xa, y0, 0.0]
            pos[B] = [xb, y0 + (40.0 if f < 3 else 0.9), 0.0]
        b["position"] = pos.copy()
        frames.append(b)
    ref, _, _ = single_world_pairs(lib, sc, frames)
    assert sum(len(r) for r in ref[1:]) > 50 and len(ref[3]) >= K, "later frames must create new pairs"
    cols = sc.collider_kwargs()
    full = F.World(lib, F.default_config(32, substeps=1))
    full.bodies_upload(**frames[0]); full.colliders_upload(**cols)
    for seeded in (True, False):
        known = np.zeros(0, np.uint64); order = None; same = True
        for f, b in enumerate(frames):
            full.bodies_upload(**b); full.run_system("UPDATE_AABB")
            mn, mx, _ = full.aabbs_download()
            parts = [shard.slab_broad_phase_step(lib, 32, b, cols, mn[:, 0], mx[:, 0], known, r, 3, prev_order=order if seeded else None) for r in range(3)]
            rec = np.concatenate(parts)
            same = same and np.array_equal(rec, ref[f])
            known = np.concatenate([known, shard.pair_keys(ref[f])])
            order = shard.slab_next_order(order, mn[:, 0], len(mn))
        if seeded:
            assert same, "slabs seeded with the persistent order must equal the single world on every frame"
        else:
            assert not same, "the scene must actually distinguish the persistent order from the upload order"
    full.close()


def test_slab_sharded_broad_phase_gloo_world_size_2_over_frames(tmp_path):
    """world_size 2 on gloo, three frames of colliders in free flight (they cross the slab boundary): the all-gathered pair
    records of every frame equal the single persistent world's new pairs, bit for bit and in order."""
    out = str(tmp_path / "slabs.npz")
    launch("slabs", out, 3)
    got = np.load(out)
    sc = SW.slab_scene()
    ref, _, _ = single_world_pairs(oracle_lib(), sc, [SW.moved(sc, s) for s in range(3)])
    for s in range(3):
        assert np.array_equal(got[f"pairs_s{s}"], ref[s]), f"frame {s}"
    assert len(ref[0]) > 500 and len(ref[1]) > 0 and len(ref[2]) > 0


def test_slab_sharded_broad_phase_f64_world():
    """Same property in the f64 build of the path (Scalar = f64: 64-bit sort keys, 4-candidate sweep batches)."""
    lib = oracle_lib()
    sc = scenes.sparse_mixed(2500, side=18.0)
    w = F.World(lib, F.default_config(64, substeps=1))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs()); w.existing_pairs_upload(np.zeros(0, np.uint64))
    w.run_system("UPDATE_AABB"); w.run_system("COLLECT_COLLISION_PAIRS")
    p = w.pairs_get(); mn, mx, _ = w.aabbs_download()
    ref = np.stack([p["collider1"], p["collider2"], p["flags"]], axis=1).astype(np.uint32)
    assert mn.dtype == np.float64 and len(ref) > 1000
    parts = [shard.slab_broad_phase_step(lib, 64, sc.body_kwargs(), sc.collider_kwargs(), mn[:, 0], mx[:, 0], np.zeros(0, np.uint64), r, 3) for r in range(3)]
    assert np.array_equal(np.concatenate(parts), ref)
