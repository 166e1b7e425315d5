#!/usr/bin/env python3
"""What happens between two steps: from a rocprofv3 --kernel-trace CSV, for every nms_kernel launch (one per step) the microseconds from the end of the step's last
conv kernel to the NMS start, the NMS and pack durations, and from the end of pack to the start of the next step's first kernel; plus the share of the traced
steps' span in which no kernel runs at all.   usage: tools/step_boundary.py <kernel_trace.csv> [first_step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nms = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
print("step   last conv end -> nms start   nms     nms end -> pack start   pack    pack end -> next kernel start   step period")
for k in range(first, len(nms) - 1):
    i = nms[k]
    prev_end = max(r["e"] for r in rows[max(0, i - 12):i])
    pk = next(j for j in range(i + 1, len(rows)) if "pack_kernel" in rows[j]["Kernel_Name"])
    nxt = next(j for j in range(pk + 1, len(rows)) if "conv" in rows[j]["Kernel_Name"])
    n, p = rows[i], rows[pk]
    print(f"{k:4d}   {(n['s'] - prev_end) / 1e3:10.1f} us            {(n['e'] - n['s']) / 1e3:6.1f}   {(p['s'] - n['e']) / 1e3:10.1f} us          {(p['e'] - p['s']) / 1e3:5.1f}   "
          f"{(rows[nxt]['s'] - p['e']) / 1e3:10.1f} us                 {(rows[nms[k + 1]]['s'] - n['s']) / 1e3:9.1f} us")
# idle share over the traced steps
i0, i1 = nms[first], nms[-1]
ev = []
for r in rows[i0:i1]:
    ev.append((r["s"], 1))
    ev.append((r["e"], -1))
ev.sort()
depth, last, idle = 0, rows[i0]["s"], 0
for t, d in ev:
    if depth == 0:
        idle += t - last
    depth += d
    last = t
span = rows[i1]["s"] - rows[i0]["s"]
print(f"no kernel running: {idle / 1e3:.1f} us of {span / 1e3:.1f} us ({100.0 * idle / span:.2f} %) over {len(nms) - 1 - first} steps")
