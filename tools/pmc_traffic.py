#!/usr/bin/env python3
"""Collect the memory-side traffic of the hot kernels with rocprofv3 PMC counters (run on the GPU box).

Follows MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass, so they are
collected in SEPARATE `--pmc` runs (kernel-trace only, no other trace domain); both are reported in KiB by rocprofv3;
on gfx950 FETCH_SIZE counts 128-byte requests at 64 B, i.e. exactly half of a wide coalesced streaming read => x2.
WRITE_SIZE is uncalibrated on gfx950 (reported as is).  Output: one JSON with per-kernel averages per launch.

usage: python tools/pmc_traffic.py OUT.json [bench args...]
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(counter, tag, bench_args):
    d = os.path.join(REPO, "gpurun_out", f"pmc_{tag}")
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag, "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-pcie", "--no-closed-loop", "--no-traffic", "--no-iters8"] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    open(os.path.join(d, "stdout.log"), "w").write(r.stdout + "\n---\n" + r.stderr[-4000:])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            per[k][0] += float(row["Counter_Value"])
            per[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in per.items() if v[1]}, r.returncode


def main():
    out_path = sys.argv[1]
    bench_args = sys.argv[2:] or ["--steps", "3", "--warmup", "1"]
    fetch, rc1 = collect("FETCH_SIZE", "fetch", bench_args)
    write, rc2 = collect("WRITE_SIZE", "write", bench_args)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0)); w = write.get(k, (0.0, 0))
        kernels[k] = {"launches": f[1] or w[1], "fetch_size_kib_raw": round(f[0], 3), "write_size_kib_raw": round(w[0], 3),
                      "fetch_bytes_corrected": round(f[0] * 1024 * 2), "write_bytes": round(w[0] * 1024),
                      "hbm_bytes_per_launch": round(f[0] * 1024 * 2 + w[0] * 1024)}
    dom = [k for k in kernels if "k_color_pass<float, 1>" in k or "k_color_pass<float, (int)1>" in k]
    res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (+ --kernel-trace); KiB -> bytes; "
                     "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted",
           "bench_args": bench_args, "returncodes": [rc1, rc2], "kernels": kernels,
           "dominant_kernel": dom[0] if dom else None,
           "hbm_bytes_per_launch": kernels[dom[0]]["hbm_bytes_per_launch"] if dom else None}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{v['launches']:6d} x {v['hbm_bytes_per_launch']/1e6:9.3f} MB  {k[:110]}")


if __name__ == "__main__":
    main()
