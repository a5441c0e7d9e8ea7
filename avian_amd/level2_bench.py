"""Level-2 sharding over real GPUs: ONE contact island (a box stack) split into x-slab worlds, one rank per GPU, the halo exchange issued by
the library (avn_comm_init: RCCL ncclSend / ncclRecv on the world's stream after every colour launch).  Every rank also steps the UNSPLIT
island on its own GPU and compares its slab against it: the split must be bit-identical.  Used by `bench.py --gpus N` (N > 1, the `level2`
object of the JSON line) and by tools/level2_multi_gpu.py.  Strong scaling: the island's size is fixed as N grows."""
from __future__ import annotations

import time

import numpy as np


def global_island(lib, F, scenes, nx, ny, nz, device, seed=5, bits=32):
    """The whole island, identical on every rank: bodies, the broad phase's pairs (found on this rank's GPU), face manifolds, colouring."""
    sc = scenes.box_stack(nx, ny, nz)
    probe = F.World(lib, F.default_config(bits, substeps=1, device=device))
    probe.bodies_upload(**sc.body_kwargs())
    probe.colliders_upload(**sc.collider_kwargs())
    probe.existing_pairs_upload(np.zeros(0, np.uint64))
    probe.run_system("UPDATE_AABB"); probe.run_system("COLLECT_COLLISION_PAIRS")
    pairs = probe.pairs_get()
    probe.close()
    rng = np.random.default_rng(seed)     # (after the pair search: the same pairs as bench.py's cfg2 world, but no resting lattice)
    sc.linear_velocity[1:] += rng.normal(scale=0.2, size=(sc.n - 1, 3))
    mf = scenes.axis_aligned_manifolds(sc, np.stack([pairs["body1"], pairs["body2"]], axis=1))
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    return sc, scenes.permute_manifolds(mf, perm), offs


def run(lib, rank, world_size, device, broadcast_bytes, all_reduce_max, barrier, dims=(50, 40, 50), substeps=4, steps=10, warmup=3, check_steps=3, bits=32):
    """broadcast_bytes(b: bytes | None) -> bytes (from rank 0); all_reduce_max(x: float) -> float; barrier()."""
    from avian_amd import _ffi as F, scenes, shard
    t_plan = time.perf_counter()
    sc, pm, offs = global_island(lib, F, scenes, *dims, device=device, bits=bits)
    plan = shard.level2_plan_lib(lib, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world_size)   # the library's planner (C ABI)
    mine = plan[rank]
    t_plan = time.perf_counter() - t_plan

    def make(split):
        w = F.World(lib, F.default_config(bits, substeps=substeps, device=device, use_graph=0 if split else 1))
        if split:
            w.bodies_upload(**{k: (np.asarray(v)[mine.bodies] if v is not None else None) for k, v in sc.body_kwargs().items()})
            scenes.upload_manifolds(w, shard.level2_local_manifolds(mine, pm), mine.color_offsets, sc.friction, sc.restitution)
            w.halo_plan_upload(mine.peers, mine.send_offsets, mine.send_bodies, mine.recv_offsets, mine.recv_bodies)
        else:
            w.bodies_upload(**sc.body_kwargs())
            scenes.upload_manifolds(w, pm, offs, sc.friction, sc.restitution)
        return w

    uid = broadcast_bytes(lib.comm_unique_id() if rank == 0 else None)
    split = make(True)
    split.comm_init(uid, world_size, rank)
    single = make(False)
    # (1) parity: the slab against the unsplit island stepped on the same GPU
    ok = 1.0
    for _ in range(check_steps):
        split.step(); single.step()
    split.synchronize(); single.synchronize()
    a, b = single.bodies_download(), split.bodies_download()
    for k in a:
        if not np.array_equal(a[k][mine.bodies], b[k]):
            ok = 0.0
    ia, ib = single.impulses_download(), split.impulses_download()
    for k in ia:
        if not np.array_equal(ia[k][mine.manifolds], ib[k]):
            ok = 0.0
    all_ok = all_reduce_max(1.0 - ok) == 0.0
    # (2) time: split island over all ranks, and the unsplit island on one GPU for the ratio
    for _ in range(warmup):
        split.step()
    split.synchronize(); barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        split.step()
    split.synchronize(); barrier()
    t_split = all_reduce_max(time.perf_counter() - t0)
    for _ in range(warmup):
        single.step()
    single.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        single.step()
    single.synchronize()
    t_single = time.perf_counter() - t0
    n_send = int(len(mine.send_bodies)); n_recv = int(len(mine.recv_bodies))
    lists = int(np.count_nonzero(np.diff(mine.send_offsets.astype(np.int64)))) if len(mine.peers) else 0
    split.close(); single.close()
    return {"status": "ok", "scalar_bits": bits, "substeps": substeps, "island": f"box stack {dims[0]}x{dims[1]}x{dims[2]} ({sc.n - 1} bodies, {len(pm['body1'])} manifolds), ONE island over {world_size} x-slabs",
            "scaling": "strong", "bit_identical_to_unsplit_island": bool(all_ok), "checked_steps": check_steps,
            "ms_per_step_split": round(t_split / steps * 1e3, 4), "ms_per_step_unsplit_one_gpu": round(t_single / steps * 1e3, 4),
            "substeps_per_s_split": round(steps * substeps / t_split, 2),
            "rank0": {"bodies": int(len(mine.bodies)), "manifolds": int(len(mine.manifolds)), "peers": [int(p) for p in mine.peers],
                      "halo_bodies_sent_per_pass": n_send, "halo_bodies_received_per_pass": n_recv, "non_empty_send_lists_per_pass": lists},
            "exchange": "avn_comm_init: grouped ncclSend / ncclRecv (RCCL) on the world's stream after every colour launch; no host code inside the step",
            "plan_seconds": round(t_plan, 3)}
