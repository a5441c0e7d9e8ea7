#!/usr/bin/env python3
"""Per-kernel PMC means of the closed loop (run on the GPU box): one `rocprofv3 --pmc <set> --kernel-trace` pass over
tools/time_closed_loop.py per counter set.  usage: python tools/pmc_any.py OUT.json STEPS FILTER "C1 C2 C3" "C4 C5" ..."""
import csv, glob, json, os, subprocess, sys
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
    subprocess.run(["rm", "-rf", d])
    return {k: {c: (v[0] / v[1], v[1]) for c, v in cs.items()} for k, cs in per.items()}, r.returncode


def main():
    out_path, steps, flt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    kernels = defaultdict(dict)
    rcs = []
    for i, s in enumerate(sys.argv[4:]):
        a, rc = collect(s.split(), f"any{i}", steps)
        rcs.append(rc)
        for k, cs in a.items():
            for c, (mean, n) in cs.items():
                kernels[k][c] = round(mean, 1); kernels[k]["launches"] = n
    json.dump({"steps": steps, "returncodes": rcs, "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, e in sorted(kernels.items()):
        if flt in k:
            print(k[:60], json.dumps(e))


if __name__ == "__main__":
    main()
