#!/usr/bin/env python3
"""HBM-side traffic of the closed loop's kernels in its SETTLED steps (run on the GPU box): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
kernel-trace only) over tools/time_closed_loop.py, per-kernel means over the launches of the last `tail` steps only (the collapse of the first
steps moves five times the data).  FETCH_SIZE x 2 on gfx950 as in tools/pmc_traffic.py.  usage: python tools/pmc_closed_loop_tail.py OUT.json [steps=120] [tail=20]"""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(counter, steps, tail):
    d = os.path.join(REPO, "gpurun_out", f"pmc_tail_{counter}")
    subprocess.run(["rm", "-rf", d]); os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--",
           sys.executable, os.path.join(REPO, "tools", "time_closed_loop.py"), "50", "40", "50", str(steps)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, stdin=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [x for x in csv.DictReader(open(f)) if x.get("Counter_Name") == counter]
    rows.sort(key=lambda x: int(x["Dispatch_Id"]))
    starts = [i for i, x in enumerate(rows) if "k_update_aabb" in x["Kernel_Name"]]
    per = defaultdict(lambda: [0.0, 0])
    if len(starts) > tail:
        for x in rows[starts[-tail - 1]:starts[-1]]:
            k = x["Kernel_Name"].split("(")[0].replace("void avn::", "").replace("avn::", "")
            per[k][0] += float(x["Counter_Value"]); per[k][1] += 1
    subprocess.run(["rm", "-rf", d])
    return {k: (v[0] / v[1], v[1] / tail) for k, v in per.items()}, r.returncode, len(starts)


def main():
    out = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120; tail = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    f, rc1, n1 = collect("FETCH_SIZE", steps, tail)
    w, rc2, n2 = collect("WRITE_SIZE", steps, tail)
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0.0, 0.0)), w.get(k, (0.0, 0.0))
        res[k] = {"launches_per_step": round(fk[1] or wk[1], 2), "fetch_MB_per_launch": round(fk[0] * 1024 * 2 / 1e6, 3), "write_MB_per_launch": round(wk[0] * 1024 / 1e6, 3)}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x 2 (gfx950); means over the launches of the last %d of %d closed-loop steps of cfg2" % (tail, steps),
               "returncodes": [rc1, rc2], "steps_seen": [n1, n2], "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_MB_per_launch"] + kv[1]["write_MB_per_launch"]) * kv[1]["launches_per_step"])[:14]:
        print(f"{v['launches_per_step']:7.2f}/step  fetch {v['fetch_MB_per_launch']:9.3f} MB  write {v['write_MB_per_launch']:9.3f} MB  {k[:70]}")


if __name__ == "__main__":
    main()
