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


def closed_loop_island(lib, F, scenes, bits, dims, steps, substeps=4, friction=0.5, device=0):
    """A cfg5-SHAPED manifold set: the device closed loop's own manifolds (the library's narrow phase, ContactIds and 24-colour ConstraintGraph) after `steps` steps of a
    box stack whose lattice is collapsing -- a deep pile overflows the graph's 23 colours, so the set carries overflow-colour manifolds on most bodies -- turned into
    the host-uploaded form level 2 works on: colour-major arrays + offsets, the overflow colour in the graph's list order.  Returns (scene with the stepped body
    state, manifolds, offsets, warm-start impulses).  Deterministic: every rank that runs it on its own GPU holds the same set."""
    sc = scenes.box_stack(*dims)
    w = F.World(lib, F.default_config(bits, substeps=substeps, device=device))
    w.bodies_upload(**sc.body_kwargs()); w.colliders_upload(**sc.collider_kwargs())
    w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=friction)
    w.pipeline_enable()
    b1_of = np.zeros(0, np.int64); b2_of = np.zeros(0, np.int64)
    for _ in range(steps):
        w.step(); w.synchronize()
        ids = w.pipeline_new_pair_ids().astype(np.int64); pr = w.pairs_get()
        if len(ids):
            top = int(ids.max()) + 1
            if top > len(b1_of):
                b1_of = np.concatenate([b1_of, np.zeros(top - len(b1_of), np.int64)]); b2_of = np.concatenate([b2_of, np.zeros(top - len(b2_of), np.int64)])
            b1_of[ids] = pr["body1"]; b2_of[ids] = pr["body2"]   # (a recycled ContactId names its newest pair)
    offs, handles = w.pipeline_handles()
    rows = w.contacts_download(handles)
    for k, v in w.bodies_download().items():
        setattr(sc, k, v)
    w.close()
    mf = dict(body1=b1_of[handles].astype(np.int32), body2=b2_of[handles].astype(np.int32), normal=rows["normal"], point_count=rows["point_count"], anchor1=rows["anchor1"],
              anchor2=rows["anchor2"], penetration=rows["penetration"], normal_speed=rows["normal_speed"], friction=rows["friction"], restitution=rows["restitution"])
    return sc, mf, offs.astype(np.int64), (rows["warm_start_normal_impulse"], rows["warm_start_tangent_impulse"])


def run(lib, rank, world_size, device, broadcast_bytes, all_reduce_max, barrier, dims=(50, 40, 50), substeps=4, steps=10, warmup=3, check_steps=3, bits=32, closed_loop_steps=0):
    """broadcast_bytes(b: bytes | None) -> bytes (from rank 0); all_reduce_max(x: float) -> float; barrier().  closed_loop_steps > 0: the island's manifolds are the
    device closed loop's own after that many steps (closed_loop_island: overflow-colour manifolds on shared bodies, exchanged level by level) instead of the
    synthetic face manifolds."""
    from avian_amd import _ffi as F, scenes, shard
    t_plan = time.perf_counter()
    warm = (None, None)
    if closed_loop_steps:
        sc, pm, offs, warm = closed_loop_island(lib, F, scenes, bits, dims, closed_loop_steps, substeps=substeps, device=device)
        fr, re = pm["friction"], pm["restitution"]
    else:
        sc, pm, offs = global_island(lib, F, scenes, *dims, device=device, bits=bits)
        fr, re = sc.friction, sc.restitution
    plan = shard.level2_plan_lib(lib, sc.position, sc.rb_type, pm["body1"], pm["body2"], offs, world_size)   # the library's planner (C ABI)
    mine = plan[rank]
    t_plan = time.perf_counter() - t_plan

    def make(split):
        w = F.World(lib, F.default_config(bits, substeps=substeps, device=device, use_graph=0 if split else 1))
        if split:
            w.bodies_upload(**{k: (np.asarray(v)[mine.bodies] if v is not None else None) for k, v in sc.body_kwargs().items()})
            lm = shard.level2_local_manifolds(mine, pm)
            pick = (lambda a: a[mine.manifolds] if isinstance(a, np.ndarray) else a)
            scenes.upload_manifolds(w, lm, mine.color_offsets, pick(fr), pick(re), pick(warm[0]), pick(warm[1]))
            mine.upload(w)
        else:
            w.bodies_upload(**sc.body_kwargs())
            scenes.upload_manifolds(w, pm, offs, fr, re, warm[0], warm[1])
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
            "manifold_source": f"the device closed loop's own after {closed_loop_steps} steps (narrow phase + ConstraintGraph of the library)" if closed_loop_steps else "synthetic face manifolds",
            "overflow_manifolds": int(offs[24] - offs[23]), "overflow_levels_as_exchange_slots": int(mine.n_overflow_levels) if mine.n_overflow_levels > 1 else 0,
            "rank0": {"bodies": int(len(mine.bodies)), "manifolds": int(len(mine.manifolds)), "peers": [int(p) for p in mine.peers],
                      "halo_bodies_sent_per_pass": n_send, "halo_bodies_received_per_pass": n_recv, "non_empty_send_lists_per_pass": lists},
            "exchange": "avn_comm_init: grouped ncclSend / ncclRecv (RCCL) on the world's stream after every colour launch; no host code inside the step",
            "plan_seconds": round(t_plan, 3)}
