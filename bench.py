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
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc side pass that measures roofline.traffic in this run")
    ap.add_argument("--no-iters8", action="store_true", help="skip the solver_iterations = 8 extension leg")
    ap.add_argument("--no-level2", action="store_true", help="N > 1: skip the one-island-over-all-GPUs leg (level-2 sharding over RCCL)")
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
    # HBM traffic of the dominant kernel: measured IN THIS RUN by a rocprofv3 side pass (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of
    # this script, kernel-trace only; corrections as MI355X_MICROARCH.md prescribes: tools/pmc_traffic.py).  When the side pass is skipped
    # or fails, `traffic` is null and `traffic_from_profile` names the committed file the last measured figure came from.
    traffic = None
    traffic_src = None
    if rank == 0 and world_size == 1 and not args.no_traffic:
        import subprocess
        outp = os.path.join(REPO, "gpurun_out", "bench_pmc_traffic.json")
        os.makedirs(os.path.dirname(outp), exist_ok=True)
        try:
            r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_traffic.py"), outp, "--steps", "2", "--warmup", "1", "--scene", args.scene],
                               capture_output=True, text=True, timeout=240)
            if r.returncode == 0 and os.path.exists(outp):
                traffic = json.load(open(outp)).get("hbm_bytes_per_launch")
                traffic_src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE side passes of this run (tools/pmc_traffic.py)"
        except Exception:
            traffic = None
    if traffic is None:
        for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            pmc = os.path.join(REPO, "profiles", name)
            if os.path.exists(pmc):
                traffic_src = {"traffic_from_profile": "profiles/" + name, "hbm_bytes_per_launch": json.load(open(pmc)).get("hbm_bytes_per_launch"),
                               "note": "NOT measured in this run"}
                break
    # whole-step figure of SURVEY.md §8(d): (228 N + (672 + 212 P) M) bytes per substep against the whole step's wall time
    n_dyn = sc.n - 1
    algo_bytes_per_substep = 228 * n_dyn + 672 * meta["n_manifolds"] + 212 * pts
    step_s = elapsed / args.steps
    whole_step = {"algorithmic_bytes_per_step": int(algo_bytes_per_substep * substeps), "achieved": round(algo_bytes_per_substep * substeps / step_s / 1e9, 2),
                  "frac": round(algo_bytes_per_substep * substeps / step_s / 1e9 / HBM_PEAK_GBS, 5),
                  "substep_loop_frac": round(algo_bytes_per_substep * substeps / (tm.substeps_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if tm.substeps_ms > 0 else None}
    iso = {"avg_launch_us": round(avg_launch_s * 1e6, 3), "achieved": round(achieved_gbs, 2), "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
           "measured": "20 back-to-back biased-solve passes on the world's stream, HIP events on that stream, nothing else running"}
    if in_step is not None:   # the PRIMARY figure: the launches as they run inside whole steps
        t_in = in_step["avg_launch_us"] * 1e-6
        prim_gbs = (algo_bytes_per_pass / int(tm_direct.bias_pass_launches)) / t_in / 1e9
        prim = {"achieved": round(prim_gbs, 2), "frac": round(prim_gbs / HBM_PEAK_GBS, 5), "avg_launch_us": in_step["avg_launch_us"],
                "measured": "HIP events around the biased-solve pass of every substep inside whole steps (direct launches, broad phase overlapped on its own stream): pass time / launches"}
    else:
        prim = {"achieved": iso["achieved"], "frac": iso["frac"], "avg_launch_us": iso["avg_launch_us"], "measured": iso["measured"]}
    roofline = {"bound": "hbm", "kernel": "k_color_pass<float, SOLVE_BIAS>", "achieved": prim["achieved"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": prim["frac"], "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": prim["avg_launch_us"], "launches_per_pass": launches_per_pass,
                "algorithmic_bytes_per_launch": int(algo_bytes_per_pass / launches_per_pass),
                "measured": prim["measured"], "isolated": iso, "whole_step": whole_step}
    #  `frac` above divides the PASS's bytes by the pass's launches -- one of which is a 2 500-manifold colour that runs in k_color_pass_oct, not in
    # the dominant kernel.  The dominant kernel's OWN launches, so that its fraction can be reproduced from profiles/*kernel_stats_cfg2.csv (its average duration there):
    big = [c for c in meta["color_counts"] if c >= 16384]   # the per-colour switch of world/contacts.hpp: lane form from ~16 k manifolds (hysteresis 12 288 / 20 480)
    if big:
        ppm = pts / max(meta["n_manifolds"], 1)
        b_launch = (248 + 88 * ppm) * (sum(big) / len(big))
        roofline["dominant_kernel_own_launches"] = {"launches_per_pass": len(big), "mean_manifolds_per_launch": round(sum(big) / len(big), 1), "algorithmic_bytes_per_launch": int(b_launch),
                                                    "colour_sizes_of_the_pass": meta["color_counts"],
                                                    "frac_at_the_isolated_launch_time": round(b_launch / (iso["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                                    "note": "bytes of one lane-form launch (600 B x its manifolds at P = 4); divide by the kernel's average duration in the rocprofv3 stats of the same command "
                                                            "(profiles/) for its own fraction; `frac_at_the_isolated_launch_time` uses this run's isolated pass time / all launches of the pass (an upper bound of the lane-form launch's time)"}

    # ---- declared extension: solver_iterations = 8 (BASELINE.json config 2 says "4 substeps x 8 XPBD iters"; the reference has no such knob --
    # SURVEY.md header note 2 -- so this is NOT the parity configuration and never `value`): 8 outer repeats of the biased solve, the relax pass
    # and the joint pass per substep
    iters8 = None
    if rank == 0 and world_size == 1 and not args.no_iters8:
        cfg.solver_iterations = 8
        w.config_set(cfg)
        for _ in range(2):
            w.step()
        w.synchronize()
        n8 = max(3, min(10, args.steps))
        c0 = time.perf_counter()
        for _ in range(n8):
            w.step()
        w.synchronize()
        ms8 = (time.perf_counter() - c0) / n8 * 1e3
        iters8 = {"solver_iterations": 8, "ms_per_step": round(ms8, 4), "substeps_per_s": round(substeps / (ms8 / 1e3), 2),
                  "contact_passes_per_s": round(substeps * 16 / (ms8 / 1e3), 1),
                  "note": "declared extension of avn_config (not reference behaviour): each substep repeats biased solve, relax and the joint pass 8 times"}
        cfg.solver_iterations = 1
        w.config_set(cfg)

    # ---- PCIe-inclusive step (reported next to `value`, never as `value`): what a host-resident ECS pays when the boundary
    # hands over host buffers every step — re-upload bodies + the colour-major manifold set, step, download bodies + impulses.
    # The host arrays are PINNED (page-locked) once, as a Bevy integration would keep its staging buffers, so that the figure is the bus,
    # not the pageable-copy path (which swung 14 -> 29 ms between boxes in round 1).
    pcie = None
    if rank == 0 and world_size == 1 and not args.no_pcie:
        from avian_amd import scenes

        def pinned(a, dtype=None):
            """A page-locked copy IN THE TYPE THE ABI TAKES (the world's scalar for the float arrays): what a host keeps as its staging buffers.
            (Round 2 pinned the float64 masters: every call then converted 200 MB to f32 into pageable memory first -- the leg measured numpy.)"""
            if a is None or not hasattr(a, "nbytes") or a.nbytes == 0:
                return a
            a = np.ascontiguousarray(a)
            if dtype is not None and a.dtype.kind == "f":
                a = a.astype(dtype)
            t = torch.from_numpy(a).pin_memory()
            keep.append(t)
            return t.numpy()
        keep = []
        int_types = {"body1": np.int32, "body2": np.int32, "point_count": np.uint8, "manifold_flags": np.uint8, "rb_type": np.uint8, "locked_axes": np.uint8,
                     "body_flags": np.uint8, "dominance": np.int8}
        bk = {k: pinned(np.asarray(v).astype(int_types[k]) if k in int_types and v is not None else v, w.dtype) for k, v in sc.body_kwargs().items()}
        mfp = {k: pinned(np.asarray(v).astype(int_types[k]) if k in int_types else v, w.dtype) for k, v in meta["manifolds"].items()}
        fr = pinned(np.full(meta["n_manifolds"], sc.friction, w.dtype)); re_ = pinned(np.full(meta["n_manifolds"], sc.restitution, w.dtype))
        n_b, n_m = sc.n, meta["n_manifolds"]
        bout = {k: pinned(np.zeros(sh, w.dtype)) for k, sh in (("position", (n_b, 3)), ("rotation", (n_b, 4)), ("linear_velocity", (n_b, 3)), ("angular_velocity", (n_b, 3)))}
        iout = {k: pinned(np.zeros(sh, w.dtype)) for k, sh in (("warm_start_normal_impulse", (n_m, 4)), ("warm_start_tangent_impulse", (n_m, 4, 2)), ("normal_impulse", (n_m, 4)))}
        n_p = 20

        def pcie_step():
            w.bodies_upload(**bk)
            scenes.upload_manifolds(w, mfp, meta["offsets"], fr, re_)
            w.step()
            w.bodies_download(out=bout); w.impulses_download(out=iout)
            w.synchronize()
        for _ in range(3):   # (the first transfers out of freshly page-locked buffers map them into the device's address space: 10-40 ms each)
            pcie_step()
        per_step = []
        for _ in range(n_p):
            c0 = time.perf_counter()
            pcie_step()
            per_step.append((time.perf_counter() - c0) * 1e3)
        # the MEDIAN step: on some boxes of the pool single transfers stall for 10-40 ms (one step in ten); the mean and the worst step are reported next to it
        ms_p = float(np.median(per_step))
        up_bytes = sum(int(np.asarray(v).nbytes) for v in bk.values() if v is not None) + sum(int(np.asarray(v).nbytes) for v in mfp.values() if hasattr(v, "nbytes")) + 2 * fr.nbytes
        down_bytes = sum(int(v.nbytes) for v in bout.values()) + sum(int(v.nbytes) for v in iout.values())
        pcie = {"ms_per_step": round(ms_p, 3), "substeps_per_s": round(substeps / (ms_p / 1e3), 2), "steps": n_p, "statistic": "median over the steps",
                "mean_ms_per_step": round(float(np.mean(per_step)), 3), "max_ms_per_step": round(float(np.max(per_step)), 3),
                "host_bytes_up_per_step": up_bytes, "host_bytes_down_per_step": down_bytes,
                "note": "page-locked host arrays in the ABI's own types through avn_bodies_upload / avn_manifolds_upload / avn_bodies_download / avn_impulses_download every step "
                        "(everything re-sent, changed or not); the device-resident path above keeps everything in HBM"}

    # ---- closed loop (secondary figure, never `value`): the same bodies with the DEVICE narrow phase instead of the fixed
    # manifold set — broad phase -> narrow phase -> status changes -> ConstraintGraph -> solver, all behind avn_step
    closed = None
    if rank == 0 and world_size == 1 and not args.no_closed_loop:
        # the frozen-manifold world (and the PCIe leg's page-locked staging buffers) are not needed any more: a second live world of this size in the
        # process costs the closed loop 2-6 % (measured: tools/time_closed_loop.py with AVN_TOOL_EXTRA_WORLD=1; bench vs the stand-alone tool on one box)
        w.close()
        try:
            keep.clear()
        except NameError:
            pass
        def closed_windows(sleeping, hooked=0.0):
            wc = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
            ck = sc.collider_kwargs()
            if hooked:   # a fraction of the boxes carries ActiveCollisionHooks (both bits); the callbacks are vectorised numpy: the leg prices the round trips, not Python loops
                fl = np.zeros(sc.n, np.uint8)
                fl[1:][np.random.default_rng(0).random(sc.n - 1) < hooked] = F.COLLIDER_FILTER_PAIRS | F.COLLIDER_MODIFY_CONTACTS
                ck["collider_flags"] = fl
                def _modify(recs):
                    recs["friction"] *= recs["friction"].dtype.type(0.9)
                wc.collision_hooks_set(lambda pairs, keep: None, _modify)
            wc.bodies_upload(**sc.body_kwargs()); wc.colliders_upload(**ck)
            wc.existing_pairs_upload(np.zeros(0, np.uint64))
            wc.collider_materials_upload(friction=sc.friction, restitution=sc.restitution)
            wc.pipeline_enable()    # ContactGraph / IdPool / ConstraintGraph bookkeeping on the device (k_graph.hip)
            if sleeping:
                wc.sleeping_enable()   # Avian's default: IslandPlugin + IslandSleepingPlugin are in SolverPlugins (dynamics/solver/mod.rs:61-84)

            def window(n):
                wc.synchronize()
                c0 = time.perf_counter()
                host = ch = byt = 0.0; ovf = 0; isl_host = awake = 0.0
                for _ in range(n):
                    wc.step()
                    wc.synchronize()   # (a frame: the host reads the step's results before it starts the next one; back-to-back avn_step calls measure 2 % slower)
                    ps = wc.pipeline_stats(); host += ps.last_host_ms; ch += ps.last_status_changes; ovf = max(ovf, ps.last_overflow_manifolds)
                    byt += substeps * (228 * (sc.n - 1) + 1520 * ps.manifolds)     # SURVEY.md §8(d) with P = 4 (an upper bound: piles hold 1-4 points)
                    if sleeping:
                        st = wc.sleeping_stats(); isl_host += st.last_host_ms; awake += st.n_awake_bodies
                wc.synchronize()
                ms = (time.perf_counter() - c0) / n * 1e3
                ps = wc.pipeline_stats()
                out = {"ms_per_step": round(ms, 3), "substeps_per_s": round(substeps / (ms / 1e3), 2), "host_bookkeeping_ms": round(host / n, 3),
                       "status_changes_per_step": round(ch / n, 1), "manifolds_at_end": ps.manifolds, "max_overflow_manifolds": int(ovf),
                       "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(byt / n / (ms / 1e3) / 1e9, 2),
                                    "frac": round(byt / n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "note": "whole closed-loop step: solver bytes of SURVEY.md §8(d) at P = 4 (upper bound) / wall time; the narrow phase's and the bookkeeping's own bytes are not counted"}}
                if sleeping:
                    st = wc.sleeping_stats()
                    out.update(island_manager_host_ms=round(isl_host / n, 3), mean_awake_bodies=round(awake / n, 1), islands_at_end=int(st.islands.n_islands), splits_total=int(st.islands.splits))
                return out
            for _ in range(4):
                wc.step()
            transient = window(20)   # steps 4..23: the pile is still compacting (~2e5 status changes per step, an overflow colour 10^5 strong and hundreds of levels deep)
            settled = window(20)     # steps 24..43
            for _ in range(56):
                wc.step()
            steady = window(20)      # steps 100..119: the pile has stopped compacting (2-3e4 status changes per step, a few hundred overflow manifolds)
            ps = wc.pipeline_stats()
            if hooked:
                hk = wc.collision_hook_stats()
                steady["hooks"] = {"hooked_colliders": int((ck["collider_flags"] != 0).sum()), "filter_queries_last_step": int(hk.last_filter_queries), "modify_records_last_step": int(hk.last_modify_queries),
                                   "bytes_each_way_last_step": int(hk.last_modify_queries) * F.hook_contact_dtype(32).itemsize + int(hk.last_filter_queries) * F.HOOK_PAIR_DTYPE.itemsize,
                                   "callback_ms_last_step": round(hk.last_callback_ms, 3), "python_errors": len(wc.host_shape_errors())}
            wc.close()
            return transient, settled, steady, ps
        transient, settled, steady, ps = closed_windows(False)
        # headline of the leg = the steady window (the sustained rate); the two earlier windows are the transient of the initial condition
        # (a perfect lattice of 100 000 touching boxes that collapses into a pile) and are reported next to it
        closed = {"ms_per_step": steady["ms_per_step"], "substeps_per_s": steady["substeps_per_s"], "host_bookkeeping_ms": steady["host_bookkeeping_ms"],
                  "window": "steps 100..119 after avn_pipeline_enable (20 steps, steady)", "active_pairs": ps.active_pairs, "roofline": steady["roofline"],
                  "steady_steps_100_119": steady, "steps_24_43": settled, "transient_steps_4_23": transient,
                  "note": "avn_pipeline_enable(1): broad phase -> Ball/Cuboid narrow phase (parry part parity-unpinned) -> status-change loop, greedy colouring, handle lists "
                          "and the overflow colour's order ALL on the device; per step the host reads three counter blocks"}
        # the same three windows with avn_sleeping_enable (round 6): persistent islands, the deferred split every other settled step, the Sleeping set at the end of
        # every step.  Nothing of cfg2's pile falls asleep inside these windows -- the figure is what sleeping COSTS a pile that is still awake.
        try:
            t2, s2, st2, _ = closed_windows(True)
            closed["sleeping_enabled"] = {"steady_steps_100_119": st2, "steps_24_43": s2, "transient_steps_4_23": t2,
                                          "settled_ratio_to_sleeping_off": round(st2["ms_per_step"] / steady["ms_per_step"], 3),
                                          "note": "avn_sleeping_enable on the same bodies: the island manager (host C++) digests the step's pairs and status changes UNDER the solver's kernels when no "
                                                  "change names a Sleeping body; split_island's walk reads a device-built CSR and, for an island that is still one piece, runs on a worker thread"}
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the line down
            closed["sleeping_enabled"] = {"status": "error: " + str(e)[:300]}
        # the same windows with 1 % of the boxes carrying ActiveCollisionHooks (round 6): filter_pairs / modify_contacts called back from inside avn_step,
        # only the hooked pairs on the bus -- the scene that used to force the PCIe-inclusive HostNarrowPhase mode below
        try:
            t3, s3, st3, _ = closed_windows(False, hooked=0.01)
            closed["collision_hooks_1pct"] = {"steady_steps_100_119": st3, "steps_24_43": s3, "transient_steps_4_23": t3,
                                              "settled_ratio_to_unhooked": round(st3["ms_per_step"] / steady["ms_per_step"], 3),
                                              "note": "avn_collision_hooks_set: 1 % of the boxes flagged FILTER_PAIRS | MODIFY_CONTACTS; the filter accepts everything, modify_contacts scales the friction "
                                                      "(vectorised numpy callbacks through ctypes); two extra round trips per step (the pending count, the records)"}
        except Exception as e:  # noqa: BLE001
            closed["collision_hooks_1pct"] = {"status": "error: " + str(e)[:300]}
        # measured HBM-side traffic of the settled closed-loop step next to the algorithmic bytes: two rocprofv3 --pmc passes of
        # tools/time_closed_loop.py (FETCH_SIZE, WRITE_SIZE separately, kernel-trace only), per-kernel means over the last 10 of 120 steps
        if not args.no_traffic:
            try:
                import subprocess
                outp = os.path.join(REPO, "gpurun_out", "bench_pmc_closed_loop.json")
                os.makedirs(os.path.dirname(outp), exist_ok=True)
                subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_closed_loop_tail.py"), outp, "120", "10"], capture_output=True, text=True, timeout=420, stdin=subprocess.DEVNULL)
                pm = json.load(open(outp))
                tot = sum((k["fetch_MB_per_launch"] + k["write_MB_per_launch"]) * k["launches_per_step"] for k in pm["kernels"].values())
                solver = sum((k["fetch_MB_per_launch"] + k["write_MB_per_launch"]) * k["launches_per_step"] for n, k in pm["kernels"].items()
                             if any(t in n for t in ("k_color_pass", "k_overflow_flow", "k_body_warm_start", "k_integrate_positions")))
                alg = substeps * (228 * (sc.n - 1) + 1520 * closed["steady_steps_100_119"]["manifolds_at_end"]) / 1e6
                closed["roofline"]["traffic"] = {"unit": "MB per step", "all_kernels": round(tot, 1), "substep_loop_kernels": round(solver, 1), "algorithmic_substep_loop": round(alg, 1),
                                                 "substep_loop_over_algorithmic": round(solver / alg, 3), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in this run (FETCH x 2 on gfx950), steps 110..119"}
            except Exception as e:  # noqa: BLE001 -- a secondary measurement must not take the line down
                closed["roofline"]["traffic"] = None
                closed["roofline"]["traffic_source"] = "profiles/r05_pmc_closed_loop_settled.json (the in-run pass failed: " + str(e)[:120] + ")"
        # ---- the closed loop with persistent islands + sleeping (avn_sleeping_enable): N is not constant any more -- every
        # window reports the awake body count next to its time.  Scene: the reference's own "Many Pyramids 3D" bench (10 x 10 pyramids of base 10:
        # 5 500 boxes in 100 islands, benches/src/dim3/many_pyramids.rs:15-64) plus one box dropped from 35 m onto one pyramid (lands at ~2.7 s):
        # the islands settle and fall asleep one by one, the impact wakes exactly one of them.  (cfg2's 100 k lattice never rests long enough:
        # it is split and put to sleep, and woken again by a still-active pair that starts touching in the next narrow phase.)
        try:
            from avian_amd import scenes as _sc
            base = _sc.many_pyramids(10, 10, 10)
            top = base.position[10:][np.argmax(base.position[10:, 1])]
            extra = np.array([[top[0] + 0.2, top[1] + 35.0, top[2] + 0.1]])
            scs = _sc.Scene(np.vstack([base.position, extra]), np.vstack([base.rotation, [[0, 0, 0, 1.0]]]), np.vstack([base.linear_velocity, [[0, 0, 0.0]]]),
                            np.vstack([base.angular_velocity, [[0, 0, 0.0]]]), np.append(base.inv_mass, base.inv_mass[-1]), np.vstack([base.inv_inertia_local, base.inv_inertia_local[-1:]]),
                            np.append(base.rb_type, 0).astype(np.uint8), np.vstack([base.half_extents, [[0.5, 0.5, 0.5]]]), np.append(base.shape, 0).astype(np.uint8))
            ws = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
            ws.bodies_upload(**scs.body_kwargs()); ws.colliders_upload(**scs.collider_kwargs())
            ws.existing_pairs_upload(np.zeros(0, np.uint64)); ws.collider_materials_upload(friction=scs.friction, restitution=scs.restitution)
            ws.pipeline_enable(); ws.sleeping_enable()
            wins = []
            budget_t0 = time.perf_counter()
            for wi in range(10):   # (until the scene has gone to sleep entirely, if it does: the last windows then time a world that costs nothing)
                c0 = time.perf_counter(); awake = host = 0.0; slept = woken = 0
                n_w = 50
                for _ in range(n_w):
                    ws.step()
                    st = ws.sleeping_stats(); awake += st.n_awake_bodies; host += st.last_host_ms; slept += st.last_islands_slept; woken += st.last_islands_woken
                ws.synchronize()
                ms = (time.perf_counter() - c0) / n_w * 1e3
                st = ws.sleeping_stats()
                wins.append({"steps": f"{wi * n_w}..{wi * n_w + n_w - 1}", "ms_per_step": round(ms, 3), "mean_awake_bodies": round(awake / n_w, 1), "islands_slept": int(slept),
                             "islands_woken_by_the_sleeping_set": int(woken), "island_host_ms_per_step": round(host / n_w, 3), "islands_at_end": int(st.islands.n_islands),
                             "sleeping_islands_at_end": int(st.islands.n_sleeping_islands), "splits_total": int(st.islands.splits), "manifolds_at_end": int(ws.pipeline_stats().manifolds),
                             "kernel_launches_last_step": int(ws.timers().kernel_launches)})
                if time.perf_counter() - budget_t0 > 60 or (wi >= 4 and wins[-1]["mean_awake_bodies"] == 0.0):
                    break
            closed["sleeping"] = {"scene": "Many Pyramids 3D (5 500 boxes, 100 islands, 10 static grounds) + one box dropped from 35 m, f32, 4 substeps, avn_sleeping_enable "
                                           "(thresholds 0.15 / 0.15, time_to_sleep 0.5 s)",
                                  "dynamic_bodies": int((scs.rb_type == 0).sum()), "windows": wins,
                                  "note": "persistent islands, deferred split, SleepIslands / WakeIslands with the reference's pop / push order (host island manager + device op pipeline); "
                                          "avn_step synchronises at its end in this mode"}
            del ws
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the line down
            closed["sleeping"] = {"status": "error: " + str(e)[:300]}

    # ---- the PCIe-inclusive step on the manifolds a closed loop really holds ("measure it on the closed loop's own step-110 manifolds"): the frozen
    # benchmark set above carries 678 k manifolds (strip contacts between diagonal neighbours included); the same 100 000 boxes hold ~204 k once settled.  HostNarrowPhase
    # mode as a host would run it: bodies + every manifold WITH its warm-start impulses up, step, bodies + impulses down, page-locked staging buffers.
    if pcie is not None and closed is not None and rank == 0 and world_size == 1:
        try:
            from avian_amd import level2_bench, scenes
            nx_c, ny_c, nz_c, _ = SCENES[args.scene]
            sc2, mf2, offs2, warm2 = level2_bench.closed_loop_island(lib, F, scenes, 32, (nx_c, ny_c, nz_c), 110, substeps=substeps, device=local_rank)
            w2 = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
            keep2 = []

            def pin2(a, dtype=None):
                a = np.ascontiguousarray(a)
                if dtype is not None and a.dtype.kind == "f":
                    a = a.astype(dtype)
                t = torch.from_numpy(a).pin_memory(); keep2.append(t)
                return t.numpy()
            it2 = {"body1": np.int32, "body2": np.int32, "point_count": np.uint8, "rb_type": np.uint8, "locked_axes": np.uint8, "body_flags": np.uint8, "dominance": np.int8}
            bk2 = {k: (pin2(np.asarray(v).astype(it2[k]) if k in it2 else v, w2.dtype) if v is not None else None) for k, v in sc2.body_kwargs().items()}
            fr2, re2 = pin2(mf2.pop("friction"), w2.dtype), pin2(mf2.pop("restitution"), w2.dtype)
            mfp2 = {k: pin2(np.asarray(v).astype(it2[k]) if k in it2 else v, w2.dtype) for k, v in mf2.items()}
            wn2, wt2 = pin2(warm2[0], w2.dtype), pin2(warm2[1], w2.dtype)
            n_b2, n_m2 = sc2.n, len(mfp2["body1"])
            bout2 = {k: pin2(np.zeros(sh, w2.dtype)) for k, sh in (("position", (n_b2, 3)), ("rotation", (n_b2, 4)), ("linear_velocity", (n_b2, 3)), ("angular_velocity", (n_b2, 3)))}
            iout2 = {k: pin2(np.zeros(sh, w2.dtype)) for k, sh in (("warm_start_normal_impulse", (n_m2, 4)), ("warm_start_tangent_impulse", (n_m2, 4, 2)), ("normal_impulse", (n_m2, 4)))}

            def pcie_step2():
                w2.bodies_upload(**bk2)
                scenes.upload_manifolds(w2, mfp2, offs2, fr2, re2, wn2, wt2)
                w2.step()
                w2.bodies_download(out=bout2); w2.impulses_download(out=iout2)
                w2.synchronize()
            for _ in range(3):
                pcie_step2()
            per2 = []
            for _ in range(20):
                c0 = time.perf_counter(); pcie_step2(); per2.append((time.perf_counter() - c0) * 1e3)
            up2 = sum(int(v.nbytes) for v in bk2.values() if v is not None) + sum(int(v.nbytes) for v in mfp2.values()) + fr2.nbytes + re2.nbytes + wn2.nbytes + wt2.nbytes
            down2 = sum(int(v.nbytes) for v in bout2.values()) + sum(int(v.nbytes) for v in iout2.values())
            pcie["closed_loop_manifolds"] = {
                "ms_per_step": round(float(np.median(per2)), 3), "mean_ms_per_step": round(float(np.mean(per2)), 3), "max_ms_per_step": round(float(np.max(per2)), 3), "steps": 20,
                "statistic": "median over the steps", "manifolds": int(n_m2), "overflow_manifolds": int(offs2[24] - offs2[23]), "host_bytes_up_per_step": int(up2), "host_bytes_down_per_step": int(down2),
                "note": "the same flow on the manifold set the device closed loop holds after 110 steps of this scene (its own narrow phase and ConstraintGraph), warm-start impulses uploaded "
                        "with the manifolds as a host ContactGraph would: what HostNarrowPhase mode costs on a settled pile of this size; the set is re-sent unchanged every step "
                        "(the flow's cost does not depend on whether it changed)"}
            w2.close(); keep2.clear()
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the line down
            pcie["closed_loop_manifolds"] = {"status": "error: " + str(e)[:300]}

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
                              f"C++ oracle (g++ -O3 -ffp-contract=off) with {threads} thread(s) on {os.cpu_count()} host cores"
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
        # the like-for-like figure: the SUBSTEP LOOP of the restatement (no glam SIMD) against the device's substep loop on the same manifolds.  The whole-step
        # rates are not divided: 97 % of the CPU step is the serial insertion sort / sweep / constraint generation outside the substeps, a quotient of them says nothing.
        best_loop = min(x for x in (cpu.get("substep_loop_only_ms"), (cpu.get("single_thread") or {}).get("substep_loop_only_ms"), (cpu.get("multi_thread") or {}).get("substep_loop_only_ms")) if x)
        # (`substep_loop_only_ms` is ONE substep -- avn_profile_system(SUBSTEP) -- of the step's `substeps`; the device figure is the whole loop of a step)
        cpu["substep_loop_speedup"] = {"cpu_one_substep_ms": best_loop, "cpu_substep_loop_ms": round(best_loop * substeps, 2), "gpu_substep_loop_ms": round(tm.substeps_ms, 4),
                                       "ratio": round(best_loop * substeps / tm.substeps_ms, 1) if tm.substeps_ms > 0 else None}

    # ---- N > 1: level-2 sharding leg -- ONE cfg2 island over all N GPUs (strong scaling), exchange issued by the library over RCCL ----------
    # Not part of `value`.  A watchdog guards the leg: should the exchange hang on some fabric, the JSON line is still printed (with
    # level2.status = "timeout") and the process exits 0 -- the headline measurement above must not depend on it.
    level2 = None
    if world_size > 1 and os.environ.get("AVN_BENCH_LEVEL2", "1") != "0" and backend == "nccl" and not args.no_level2:
        import threading
        from avian_amd import level2_bench
        done = threading.Event()
        line_holder = {}

        def watchdog():
            if not done.wait(float(os.environ.get("AVN_BENCH_LEVEL2_TIMEOUT", "180"))):
                if rank == 0 and "make" in line_holder:
                    part = dict(line_holder.get("partial") or {}, status="timeout" if "partial" not in line_holder else "ok (cfg5 leg timed out)")
                    print(json.dumps(line_holder["make"](part)), flush=True)
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()

        def bcast(b):
            o = [b]
            dist.broadcast_object_list(o, src=0)
            return o[0]

        def armax(x):
            tt = torch.tensor([x], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        nx_, ny_, nz_, _ = SCENES[args.scene]
    else:
        done = None

    # ---- N > 1: the closed loop SHARDED BY ISLANDS (round 5): the reference's Many Pyramids scene (5 500 boxes, 100 islands), whole pyramids per rank, the
    # integer bookkeeping replicated in the library (avn_shard_*), three flat-tensor all-gathers per step.  Every rank checks that its replicated colour lists
    # hash to rank 0's.  Not part of `value`; measured for the first time on the driver's node (development boxes have one GPU: tests/test_gpu_sharded_closed_loop.py
    # runs the same code as four worlds on one device).
    sharded_holder = {}

    def run_sharded_leg():
        import zlib
        from avian_amd import scenes as _sc
        base, rows, cols = 10, 10, 10
        scs = _sc.many_pyramids(base, rows, cols)
        bodies_s, colliders_s = scs.body_kwargs(), scs.collider_kwargs()
        per = base * (base + 1) // 2
        pyr = (np.arange(scs.n) - rows) // per
        rk = np.where(np.arange(scs.n) < rows, -1, pyr * world_size // (rows * cols)).astype(np.int32)
        pl = shard.ShardPlan(world_size, rk.copy(), rk, rows * cols)
        b_, loc_, g2l_ = shard.split_bodies(pl, rank, bodies_s)
        c_ = shard.split_colliders(g2l_, colliders_s)
        ws = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
        ws.bodies_upload(**b_); ws.colliders_upload(**c_); ws.existing_pairs_upload(np.zeros(0, np.uint64)); ws.collider_materials_upload(friction=0.5)
        loop = shard.ShardedClosedLoopNative(lib, ws, pl, rank, colliders_s)
        gather = shard.tensor_gather(dist, torch)
        for _ in range(10):
            loop.step(gather)
        barrier(); c0 = time.perf_counter()
        n_st = 30
        changes = 0
        for _ in range(n_st):
            changes += loop.step(gather)
        ws.synchronize(); barrier()
        dt_local = time.perf_counter() - c0
        tt = torch.tensor([dt_local], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        off, handles = loop.global_lists
        h = zlib.crc32(handles.tobytes(), zlib.crc32(off.tobytes()))
        hs = gather(np.array([h], np.uint32))
        st = loop.shard.stats()
        return {"status": "ok", "scene": "Many Pyramids 3D (5 500 boxes, 100 islands), whole pyramids per rank", "n_ranks": world_size, "steps": n_st,
                "ms_per_step": round(float(tt.item()) / n_st * 1e3, 3), "substeps_per_s": round(n_st * substeps / float(tt.item()), 2),
                "status_changes_per_step": round(changes / n_st, 1), "contact_ids_handed_out": int(st.next_id), "pairs_removed": int(st.pairs_removed),
                "replicated_colour_lists_equal_on_all_ranks": bool((hs == hs[0]).all()),
                "bookkeeping": "avn_shard_* (host C++), payloads as flat tensors over torch.distributed (" + backend + "); physics through the low-level ABI (host pipeline mode)"}

    # ---- N > 1 (round 6): the same scene through the DEVICE-sharded closed loop (avn_dshard_*): every rank replicates the integer / geometry front on its own device, simulates its
    # pyramids, and avn_step itself issues the one all-gather of the step (ncclAllGather on the world's stream: avn_comm_init).  The host reads counters only.
    def run_dsharded_leg():
        import zlib
        from avian_amd import scenes as _sc
        base, rows, cols = 10, 10, 10
        scs = _sc.many_pyramids(base, rows, cols)
        per = base * (base + 1) // 2
        pyr = (np.arange(scs.n) - rows) // per
        owner = np.where(np.arange(scs.n) < rows, -1, pyr * world_size // (rows * cols)).astype(np.int32)
        wd = F.World(lib, F.default_config(32, substeps=substeps, device=local_rank))
        wd.bodies_upload(**scs.body_kwargs()); wd.colliders_upload(**scs.collider_kwargs()); wd.existing_pairs_upload(np.zeros(0, np.uint64)); wd.collider_materials_upload(friction=0.5)
        wd.pipeline_enable(); wd.dshard_enable(world_size, rank, owner)
        box = [lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        wd.comm_init(box[0], world_size, rank)
        for _ in range(10):
            wd.step()
        wd.synchronize(); barrier(); c0 = time.perf_counter()
        n_st = 30
        for _ in range(n_st):
            wd.step(); wd.synchronize()
        barrier()
        tt = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        off, handles = wd.pipeline_handles()
        bd = wd.bodies_download()
        h = zlib.crc32(bd["position"].tobytes(), zlib.crc32(handles.tobytes(), zlib.crc32(off.tobytes())))
        hs = shard.tensor_gather(dist, torch)(np.array([h], np.uint32))
        d = wd.dshard_stats()
        wd.close()
        return {"status": "ok", "scene": "Many Pyramids 3D (5 500 boxes, 100 islands), whole pyramids per rank", "n_ranks": world_size, "steps": n_st,
                "ms_per_step": round(float(tt.item()) / n_st * 1e3, 3), "substeps_per_s": round(n_st * substeps / float(tt.item()), 2),
                "own_bodies": int(d.own_bodies), "own_manifolds": int(d.own_manifolds), "global_manifolds": int(d.global_manifolds), "all_gathers_issued_by_the_library": int(d.exchanges),
                "bytes_sent_per_step": int(d.bytes_sent_per_step), "host_bytes_read_per_step": "three counter blocks (4 + 36 + 352 bytes)",
                "colour_lists_and_ALL_bodies_equal_on_all_ranks": bool((hs == hs[0]).all()),
                "bookkeeping": "replicated on every rank's device (k_graph.hip); the solver and one ncclAllGather of 64 B per own body sharded"}

    def make_line(level2_obj):
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
            # the same bodies in the device closed loop (real contacts: broad phase -> narrow phase -> bookkeeping -> solver), settled window; `value` above stays the
            # metric BASELINE.json defines (fixed manifold set), this is the number a user of the closed loop gets
            "value_closed_loop": (closed or {}).get("substeps_per_s"),
            "value_frozen": round(total_substeps / elapsed, 3),
            "value_of_record": "value_closed_loop (steps 100..119 of the device closed loop on the same 100 000 boxes: the rate a user of the drop-in gets); `value` is kept on BASELINE.json's "
                               "fixed-manifold definition (SURVEY.md section 8d) because the driver's contract times exactly `steps` steps of that workload",
            "solver_iterations_8": iters8,
            "pcie_inclusive": pcie,
            "closed_loop": closed,
            "cpu_baseline": cpu,
        }
        if sharded_holder:
            out["closed_loop_sharded"] = sharded_holder.get("result")
            if "device" in sharded_holder: out["closed_loop_sharded_device"] = sharded_holder["device"]
        if level2_obj is not None:
            out["level2"] = level2_obj
            # The STRONG-scaling figure of an N > 1 run, at the top level (`value` / `scaling` above stay the contract's weak-scaling figure:
            # N independent stacks, no data-path collective).  One island -- the same total work -- over N GPUs with the per-colour RCCL
            # exchange issued by the library: the unsplit one-GPU step time next to the split N-GPU step time, parity checked against the unsplit island.
            def strong(l2):
                if not isinstance(l2, dict) or l2.get("status") != "ok" or not l2.get("ms_per_step_split") or not l2.get("ms_per_step_unsplit_one_gpu"):
                    return {"status": (l2 or {}).get("status", "not run") if isinstance(l2, dict) else "not run"}
                try:   # (the two step times only: the driver computes efficiencies itself)
                    return {"status": "ok", "workload": l2.get("island"), "n_gpus": world_size, "ms_per_step_one_gpu": l2["ms_per_step_unsplit_one_gpu"],
                            "ms_per_step_n_gpus": l2["ms_per_step_split"], "substeps_per_s_n_gpus": l2.get("substeps_per_s_split"),
                            "bit_identical_to_the_unsplit_island": l2.get("bit_identical_to_unsplit_island")}
                except Exception as e:  # noqa: BLE001 -- a secondary object must not take the line down
                    return {"status": "error: " + str(e)[:200]}
            out["strong_scaling"] = {"cfg2_one_island": strong(level2_obj), "cfg5_one_island_f64": strong(level2_obj.get("cfg5_500k_f64") if isinstance(level2_obj, dict) else None),
                                     "note": "level-2 sharding (DESIGN.md section 6): measured for the first time on the driver's multi-GPU node; no curve exists from development (one GPU per box)"}
        return out

    if world_size > 1 and os.environ.get("AVN_BENCH_SHARDED", "1") != "0":
        if done is not None:
            line_holder["make"] = make_line   # (under the level-2 watchdog from here on: a hang prints the line without this leg)
        try:
            sharded_holder["result"] = run_sharded_leg()
        except Exception as e:  # noqa: BLE001 -- never fatal for the headline figure
            sharded_holder["result"] = {"status": "error: " + str(e)[:300]}
        if backend == "nccl":   # (the library's own exchange needs RCCL: one rank per device)
            try:
                sharded_holder["device"] = run_dsharded_leg()
            except Exception as e:  # noqa: BLE001
                sharded_holder["device"] = {"status": "error: " + str(e)[:300]}
    if done is not None:
        line_holder["make"] = make_line
        try:
            w.close()   # (the slab world + the unsplit island need the memory headroom of a fresh device, not this world's buffers)
            level2 = level2_bench.run(lib, rank, world_size, local_rank, bcast, armax, dist.barrier, dims=(nx_, ny_, nz_), substeps=substeps)
            line_holder["partial"] = level2
            # cfg5 (BASELINE.json: 500 k cuboids, f64, 8 substeps, ONE island): the configuration level 2 is for -- a colour launch of the
            # unsplit island takes 60-90 us, so the per-colour exchange has room to pay off
            if level2.get("status") == "ok" and os.environ.get("AVN_BENCH_LEVEL2_CFG5", "1") != "0":
                try:
                    level2["cfg5_500k_f64"] = level2_bench.run(lib, rank, world_size, local_rank, bcast, armax, dist.barrier, dims=(100, 50, 100), substeps=8,
                                                               steps=5, warmup=2, check_steps=2, bits=64)
                except Exception as e:  # noqa: BLE001
                    level2["cfg5_500k_f64"] = {"status": "error: " + str(e)[:300]}
                # round 6: the REAL cfg5 -- the closed loop's own manifolds 6 steps into the collapse (~ 1.7 * 10^5 of them in the overflow colour, on bodies shared
                # between slabs): the overflow colour travels level by level (hundreds of exchange slots per pass: a latency chain, reported as measured)
                if os.environ.get("AVN_BENCH_LEVEL2_CFG5_CLOSED_LOOP", "1") != "0":
                    try:
                        level2["cfg5_500k_f64_closed_loop_manifolds"] = level2_bench.run(lib, rank, world_size, local_rank, bcast, armax, dist.barrier, dims=(100, 50, 100), substeps=8,
                                                                                         steps=3, warmup=1, check_steps=1, bits=64, closed_loop_steps=6)
                    except Exception as e:  # noqa: BLE001
                        level2["cfg5_500k_f64_closed_loop_manifolds"] = {"status": "error: " + str(e)[:300]}
        except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal for the headline figure
            level2 = {"status": "error: " + str(e)[:300]}
        done.set()
    if rank == 0:
        print(json.dumps(make_line(level2)), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
