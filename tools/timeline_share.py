#!/usr/bin/env python3
"""Who owns the GPU's time when several lanes run concurrently?  Reads a rocprofv3 --kernel-trace CSV, takes the steps
between consecutive nms_kernel dispatches, cuts the timeline at every kernel start / end and shares each slice equally among the
kernels running in it.  Prints, per (kernel, grid size), launches per step, mean duration, attributed ms per step and the share;
plus the time in which nothing runs.   usage: tools/timeline_share.py <kernel_trace.csv> [first_step=2]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(conv_\w+)<(.*)>", name)
    if not m:
        return name.split("(")[0][:40]
    return f"{m.group(1)}<{m.group(2)[:60]}>"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    marks = [r["e"] for r in rows if "nms_kernel" in r["Kernel_Name"]]
    if len(marks) < first + 2:
        sys.exit("not enough steps in the trace")
    t0, t1 = marks[first], marks[-1]
    nsteps = len(marks) - 1 - first
    ks = [r for r in rows if r["s"] >= t0 and r["e"] <= t1 + 1]
    ev = []
    for i, r in enumerate(ks):
        ev.append((r["s"], 1, i))
        ev.append((r["e"], 0, i))
    ev.sort()
    live = set()
    share = defaultdict(float)
    cnt = defaultdict(int)
    dur = defaultdict(float)
    idle = 0.0
    conc = defaultdict(float)
    last = t0
    for t, kind, i in ev:
        if t > last:
            if live:
                for j in live:
                    share[j] += (t - last) / len(live)
                conc[len(live)] += t - last
            else:
                idle += t - last
            last = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    agg = defaultdict(float)
    for i, r in enumerate(ks):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        agg[key] += share[i]
        cnt[key] += 1
        dur[key] += r["e"] - r["s"]
    span = (t1 - t0) / 1e6
    print(f"# {nsteps} steps, {span / nsteps:.3f} ms per step; nothing running {idle / 1e6 / nsteps:.3f} ms per step; "
          + ", ".join(f"{k} kernels live {v / 1e6 / nsteps:.3f} ms" for k, v in sorted(conc.items())))
    print(f"{'kernel':70s} {'grid':>9s} {'n/step':>7s} {'mean_us':>9s} {'ms/step':>8s} {'share':>6s}")
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{key[0]:70s} {key[1]:>9s} {cnt[key] / nsteps:7.1f} {dur[key] / cnt[key] / 1e3:9.1f} {v / 1e6 / nsteps:8.3f} {100 * v / 1e6 / span:5.1f}%")


if __name__ == "__main__":
    main()
