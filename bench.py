#!/usr/bin/env python3
"""bench.py — physics substeps/second of the MI355X hot path on BASELINE.json's cfg2 (100k-cuboid box stack).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run with
one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over the device-resident world (`avn_step`): AABB update + sweep-and-prune
broad phase + solver-body/constraint preparation + S substeps (integrate, warm start, biased solve, integrate
positions, relax, XPBD) + restitution + write-back + impulse store — everything SURVEY.md §8(d) counts in the
"whole step".  Inputs are resident in HBM before the timed region; the narrow phase (parry, out of scope) is not
part of the path, so the manifold set is fixed during the timed steps while body state evolves.  Inside avn_step the
broad phase runs on a second stream next to the solver (it only reads what the solver rewrites at the very end), so
`device_ms.broad_phase` is its own duration and the four device_ms figures do not add up to ms_per_step.

  value      = N_gpus * K * substeps / max-over-ranks wall seconds   (physics substeps / second, whole step)
  roofline   = the dominant kernel (k_color_pass<SOLVE_BIAS>: TGS-Soft biased contact solve, one launch per graph
               colour): ALGORITHMIC bytes (248 + 88 P per manifold) / duration measured with HIP
               events on the library's own stream (avn_profile_system), against the 8 TB/s HBM3E peak.
  cpu_baseline = the CPU oracle (C++ restatement of the reference) on the same inputs, bounded sample: 1 thread and
                 min(64, host cores) threads running the reference's own parallel loops; `value` is the better of the two.

Multi-GPU (weak scaling): the path shards by interaction islands (avian_amd/shard.py).  With N ranks the global
scene is N 100k-stacks side by side on one static slab; `avn_islands_partition` assigns one island (stack) to each
rank, every rank uploads and steps ONLY its sub-world, and the only collective in the timed loop is the per-step
all-gather of one AABB per rank (RCCL, 48 bytes) that detects islands of different ranks coming into AABB contact.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

SCENES = {
    # name: (nx, ny, nz, substeps)   — cfg2 is the configuration BASELINE.json's metric is quoted on
    "cfg2_box_stack_100k": (50, 40, 50, 4),
    "box_stack_12k": (25, 20, 25, 4),
    "box_stack_1k": (10, 10, 10, 4),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def build_inputs(lib, scene_name, rank=0, world_size=1):
    """This rank's (sub-)scene.  N = 1: the whole cfg2 stack.  N > 1: the global N-stack scene is planned with
    avn_islands_partition and only the rank's own islands (+ the static slab) are kept."""
    from avian_amd import _ffi as F, scenes, shard
    nx, ny, nz, substeps = SCENES[scene_name]
    if world_size == 1:
        return scenes.box_stack(nx, ny, nz), substeps, None
    glob = scenes.box_stacks(world_size, nx, ny, nz)
    per = nx * ny * nz
    edges = np.concatenate([scenes.lattice_edges(1 + s * per, nx, ny, nz) for s in range(world_size)])
    pl = shard.plan(lib, glob.rb_type, glob.position, edges, world_size)
    assert pl.n_islands == world_size and np.bincount(pl.rank_of_body[1:], minlength=world_size).tolist() == [per] * world_size
    return glob.subset(pl.local_bodies(rank)), substeps, pl


def setup_world(world, lib, sc, pairs_from=None):
    """Upload bodies + colliders, run the broad phase once to obtain the pair list, generate the synthetic face
    manifolds for those pairs, colour them with the host ConstraintGraph and upload them.  Returns metadata."""
    from avian_amd import scenes
    world.bodies_upload(**sc.body_kwargs())
    world.colliders_upload(**sc.collider_kwargs())
    world.existing_pairs_upload(np.zeros(0, np.uint64))
    world.run_system("UPDATE_AABB")
    world.run_system("COLLECT_COLLISION_PAIRS")
    pairs = world.pairs_get()
    mf = scenes.axis_aligned_manifolds(sc, np.stack([pairs["body1"], pairs["body2"]], axis=1))
    offs, perm = scenes.color_manifolds(lib, mf, sc.rb_type)
    pm = scenes.permute_manifolds(mf, perm)
    scenes.upload_manifolds(world, pm, offs, sc.friction, sc.restitution)
    counts = np.diff(offs.astype(np.int64))
    return dict(n_pairs=int(len(pairs)), n_manifolds=int(len(perm)), points=int(pm["point_count"].sum()),
                colors_used=int((counts > 0).sum()), color_counts=[int(c) for c in counts if c > 0], manifolds=pm,
                offsets=offs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="cfg2_box_stack_100k", choices=sorted(SCENES))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop (device narrow phase) leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU oracle sample (split between the 1-thread and the multi-thread run)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-thread CPU sample (default min(64, host cores))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    # validation-only overrides (NOT used by the driver): run the N > 1 code path on a 1-GPU box with gloo
    backend = os.environ.get("AVN_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("AVN_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"

    import avian_amd
    from avian_amd import _ffi as F
    lib = avian_amd.load_library()
    from avian_amd import shard
    sc, substeps, plan = build_inputs(lib, args.scene, rank, world_size)
    cfg = F.default_config(32, substeps=substeps, device=local_rank, use_graph=0 if args.no_graph else 1)
    w = F.World(lib, cfg)
    meta = setup_world(w, lib, sc)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the per-step exchange of the sharded path — all-gather of each rank's dynamic-body bounds over RCCL.
    # The gather of step s is checked at step s + 1 (bounds are swept AABBs, i.e. already one step conservative),
    # so the collective overlaps the next step instead of stalling it.
    pending = [None]
    overlaps_seen = [0]

    def exchange():
        if world_size == 1:
            return
        if pending[0] is not None:
            work, outs = pending[0]
            work.wait()
            a = torch.stack(outs).cpu().numpy()
            overlaps_seen[0] += len(shard.bounds_overlap(a[:, :3], a[:, 3:]))
        mn, mx = w.dynamic_bounds()
        t = torch.tensor(np.concatenate([mn, mx]), dtype=torch.float64, device=coll_device)
        outs = [torch.empty_like(t) for _ in range(world_size)]
        pending[0] = (dist.all_gather(outs, t, async_op=True), outs)

    for _ in range(args.warmup):
        w.step()
        exchange()
    w.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
        exchange()
    w.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        assert overlaps_seen[0] == 0, "independent stacks must not trigger a re-partition"
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # per-phase device times: the MEDIAN of 7 further steps, each read back on its own (one step's events are a noisy sample:
    # a step that finds new pairs runs the emit pass, a pair-set rebuild lands on another); not part of the timed region
    samples = []
    for _ in range(7):
        w.step()
        samples.append(w.timers())

    class _Med:
        pass
    tm = _Med()
    for f in ("broad_phase_ms", "prepare_ms", "substeps_ms", "finalize_ms", "step_ms"):
        setattr(tm, f, float(np.median([getattr(x, f) for x in samples])))
    tm.kernel_launches = samples[-1].kernel_launches
    tm.contact_constraint_count = samples[-1].contact_constraint_count
    tm.pair_count = samples[-1].pair_count
    tm.island_blocks = samples[-1].island_blocks

    # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream -----------------------
    # (1) IN whole steps: the library brackets the biased-solve pass of every substep with events on its own
    #     stream (avn_timers.bias_pass_ms / bias_pass_launches).  Events captured into a hipGraph cannot be read back on this
    #     runtime, so the same world is switched to direct launches (use_graph = 0) for 8 extra whole steps — same kernels,
    #     same overlapped broad phase — and the last step's figure is taken;
    # (2) isolated: 20 back-to-back passes outside a step (nothing running next to them), for comparison.
    cfg.use_graph = 0
    w.config_set(cfg)
    for _ in range(8):
        w.step()
    tm_direct = w.timers()
    cfg.use_graph = 0 if args.no_graph else 1
    w.config_set(cfg)
    pts = meta["points"]
    algo_bytes_per_pass = 248 * meta["n_manifolds"] + 88 * pts  # SURVEY.md §8d, biased solve pass
    reps = 20
    w.profile_system("SOLVE_CONTACTS_BIAS", 2)
    ms_iso, launches_iso = w.profile_system("SOLVE_CONTACTS_BIAS", reps)
    iso_launch_s = (ms_iso / 1e3) / max(launches_iso, 1)
    launches_per_pass = max(launches_iso // reps, 1)
    avg_launch_s = iso_launch_s
    achieved_gbs = (algo_bytes_per_pass / launches_per_pass) / avg_launch_s / 1e9
    in_step = None
    if tm_direct.bias_pass_launches and tm_direct.bias_pass_ms > 0:
        t_in = (tm_direct.bias_pass_ms / 1e3) / int(tm_direct.bias_pass_launches)
        in_step = {"avg_launch_us": round(t_in * 1e6, 3), "frac": round((algo_bytes_per_pass / int(tm_direct.bias_pass_launches)) / t_in / 1e9 / HBM_PEAK_GBS, 5),
                   "note": "pass time / launches inside whole steps (direct launches): includes inter-launch gaps and the contention of the broad phase "
                           "running on its own stream next to the first substep"}
    traffic = None
    pmc = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_color_pass<float, SOLVE_BIAS>", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_us": round(avg_launch_s * 1e6, 3), "launches_per_pass": launches_per_pass,
                "algorithmic_bytes_per_launch": int(algo_bytes_per_pass / launches_per_pass),
                "measured": "20 back-to-back biased-solve passes (300 launches) on the world's stream, HIP events on that stream, nothing else running",
                "in_step": in_step}

    # ---- PCIe-inclusive step (reported next to `value`, never as `value`): what a host-resident ECS pays when the boundary
    # hands over host buffers every step — re-upload bodies + the colour-major manifold set, step, download bodies + impulses
    pcie = None
    if rank == 0 and world_size == 1 and not args.no_pcie:
        from avian_amd import scenes
        n_p = 3
        w.synchronize()
        c0 = time.perf_counter()
        for _ in range(n_p):
            w.bodies_upload(**sc.body_kwargs())
            scenes.upload_manifolds(w, meta["manifolds"], meta["offsets"], sc.friction, sc.restitution)
            w.step()
            w.bodies_download(); w.impulses_download()
        w.synchronize()
        ms_p = (time.perf_counter() - c0) / n_p * 1e3
        host_bytes = sum(int(np.asarray(v).nbytes) for v in sc.body_kwargs().values() if v is not None) + \
            sum(int(np.asarray(v).nbytes) for v in meta["manifolds"].values() if hasattr(v, "nbytes"))
        pcie = {"ms_per_step": round(ms_p, 3), "substeps_per_s": round(substeps / (ms_p / 1e3), 2), "host_bytes_up_per_step": host_bytes,
                "note": "pageable host arrays through avn_bodies_upload / avn_manifolds_upload / *_download every step (incidence CSR rebuilt on the host); "
                        "the device-resident path above keeps everything in HBM"}

    # ---- closed loop (secondary figure, never `value`): the same bodies with the DEVICE narrow phase instead of the fixed
    # manifold set — broad phase -> narrow phase -> status changes -> ConstraintGraph -> solver, all behind avn_step
    closed = None
    if rank == 0 and world_size == 1 and not args.no_closed_loop:
        wc = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
        wc.bodies_upload(**sc.body_kwargs()); wc.colliders_upload(**sc.collider_kwargs())
        wc.existing_pairs_upload(np.zeros(0, np.uint64))
        wc.collider_materials_upload(friction=sc.friction, restitution=sc.restitution)
        wc.pipeline_enable()
        for _ in range(4):
            wc.step()
        wc.synchronize()
        n_c = 10
        c0 = time.perf_counter()
        host_ms = changes = 0.0
        for _ in range(n_c):
            wc.step()
            ps = wc.pipeline_stats(); host_ms += ps.last_host_ms; changes += ps.last_status_changes
        wc.synchronize()
        ms_c = (time.perf_counter() - c0) / n_c * 1e3
        ps = wc.pipeline_stats()
        closed = {"ms_per_step": round(ms_c, 3), "substeps_per_s": round(substeps / (ms_c / 1e3), 2), "active_pairs": ps.active_pairs,
                  "manifolds": ps.manifolds, "overflow_manifolds": ps.last_overflow_manifolds, "status_changes_per_step": round(changes / n_c, 1),
                  "host_bookkeeping_ms": round(host_ms / n_c, 3),
                  "note": "avn_pipeline_enable: Ball/Cuboid narrow phase on device (parry part parity-unpinned), host status processing in the library"}
        del wc

    # ---- CPU baseline: the oracle on the same inputs, rank 0 at N=1 only, bounded sample -------------------------
    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from helpers import oracle_lib  # cpu_baseline leg: the oracle is the thing timed here, by contract
        from avian_amd import scenes
        def cpu_sample(threads, seconds):
            """One bounded sample of the oracle with `threads` pool threads (oracle/avo_parallel.hpp: the reference's own
            par_for_each / par_iter_mut loops); protocol of benches/src/cli.rs:358-405: one un-timed step, then the mean."""
            os.environ["AVO_THREADS"] = str(threads)   # read once, at world creation
            try:
                wo = F.World(oracle_lib(), F.default_config(32, substeps=substeps))
            finally:
                del os.environ["AVO_THREADS"]
            wo.bodies_upload(**sc.body_kwargs())
            wo.colliders_upload(**sc.collider_kwargs())
            wo.existing_pairs_upload(np.zeros(0, np.uint64))
            wo.run_system("UPDATE_AABB")
            wo.run_system("COLLECT_COLLISION_PAIRS")
            scenes.upload_manifolds(wo, meta["manifolds"], meta["offsets"], sc.friction, sc.restitution)
            c0 = time.perf_counter(); wo.step(); first = time.perf_counter() - c0
            n_cpu = int(max(1, min(args.steps, seconds / max(first, 1e-6))))
            c0 = time.perf_counter()
            for _ in range(n_cpu):
                wo.step()
            cpu_s = time.perf_counter() - c0
            sm, _ = wo.profile_system("SUBSTEP", 1)
            wo.close()
            return {"value": round(n_cpu * substeps / cpu_s, 4), "unit": "substeps/s", "cores": threads, "kind": "port",
                    "sample": f"{n_cpu} whole steps ({n_cpu * substeps} substeps) of the same {args.scene} inputs after 1 warm-up step, "
                              f"C++ oracle (g++ -O2 -ffp-contract=off) with {threads} thread(s) on {os.cpu_count()} host cores"
                              + ("" if threads == 1 else "; threads follow the reference's own parallel loops (par_for_each over a colour's constraints, "
                                                          "chunk = len / threads, min_len 64; par_iter_mut over bodies); the broad phase is serial as in the reference"),
                    "ms_per_step": round(cpu_s / n_cpu * 1e3, 2), "substep_loop_only_ms": round(sm, 2)}

        threads = max(1, min(args.cpu_threads or 64, os.cpu_count() or 1))
        single = cpu_sample(1, args.cpu_seconds / 2 if threads > 1 else args.cpu_seconds)
        cpu = single
        if threads > 1:
            multi = cpu_sample(threads, args.cpu_seconds / 2)
            cpu = multi if multi["value"] >= single["value"] else single    # the baseline is the CPU's best
            cpu = dict(cpu, single_thread={k: single[k] for k in ("value", "ms_per_step", "substep_loop_only_ms")},
                       multi_thread={k: multi[k] for k in ("value", "cores", "ms_per_step", "substep_loop_only_ms")})

    if rank == 0:
        total_substeps = world_size * args.steps * substeps
        out = {
            "metric": "physics substeps/sec at N dynamic bodies (3D)",
            "value": round(total_substeps / elapsed, 3),
            "unit": "substeps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.scene, "dynamic_bodies_per_gpu": sc.n - 1, "manifolds_per_gpu": meta["n_manifolds"],
                       "contact_points_per_gpu": pts, "broadphase_pairs": meta["n_pairs"], "substeps": substeps,
                       "solver_iterations": 1, "dt": 1.0 / 60.0, "colors_used": meta["colors_used"],
                       "hip_graph": not args.no_graph, "sharding": ("single world" if world_size == 1 else
                                    "avn_islands_partition: one interaction island (100k stack) per rank; per-step RCCL all-gather of rank bounds only"),
                       "narrow_phase": "out of path: fixed synthetic face manifolds"},
            "device_ms": {"broad_phase": round(tm.broad_phase_ms, 4), "prepare": round(tm.prepare_ms, 4),
                          "substeps": round(tm.substeps_ms, 4), "finalize": round(tm.finalize_ms, 4),
                          "kernel_launches_per_step": tm.kernel_launches},
            "substep_loop_only_substeps_per_s": round(substeps / (tm.substeps_ms / 1e3), 2) if tm.substeps_ms > 0 else None,
            "roofline": roofline,
            "pcie_inclusive": pcie,
            "closed_loop": closed,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
