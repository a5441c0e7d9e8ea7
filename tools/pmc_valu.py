#!/usr/bin/env python3
"""VALU-side PMC counters of the closed loop's kernels (run on the GPU box): `rocprofv3 --pmc <set> --kernel-trace` over
tools/time_closed_loop.py, one pass per counter set (SQ_INSTS_VALU SQ_WAVES | VALUBusy VALUUtilization), per-kernel means per launch.
usage: python tools/pmc_valu.py OUT.json [steps=44]"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(counters, tag, steps):
    d = os.path.join(REPO, "gpurun_out", f"pmc_{tag}")
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag, "--",
           sys.executable, os.path.join(REPO, "tools", "time_closed_loop.py"), "50", "40", "50", str(steps)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")
            c = per[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
    return {k: {c: (v[0] / v[1], v[1]) for c, v in cs.items()} for k, cs in per.items()}, r.returncode


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 44
    a, rc1 = collect(["SQ_INSTS_VALU", "SQ_WAVES"], "valu_a", steps)
    b, rc2 = collect(["VALUBusy", "VALUUtilization"], "valu_b", steps)
    kernels = {}
    for k in sorted(set(a) | set(b)):
        e = {}
        for src in (a.get(k, {}), b.get(k, {})):
            for c, (mean, n) in src.items():
                e[c] = round(mean, 3); e["launches"] = n
        if "SQ_INSTS_VALU" in e and e.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)
        kernels[k] = e
    json.dump({"method": "rocprofv3 --pmc, two passes, kernel-trace only; means per launch over the whole run (all steps)", "steps": steps,
               "returncodes": [rc1, rc2], "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0) * kv[1].get("launches", 0))[:14]:
        print(f"{k[:44]:44s} {e}")


if __name__ == "__main__":
    main()
