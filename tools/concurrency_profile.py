#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: over the steps between two nms_kernel launches, the share of time with 0 / 1 / 2 / 3 ... conv kernels running at once, and per
tenth of the step which kernels own the time (the timeline cut at every kernel start / end).   usage: tools/concurrency_profile.py <kernel_trace.csv> [first_step]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nms = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]


def short(n):
    m = re.search(r"(conv_bf16_mfma|conv_f32_mfma|conv_\w+|nms_kernel|pack_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:24]


tot = collections.Counter()
deciles = [collections.Counter() for _ in range(10)]
nsteps = 0
for k in range(first, len(nms) - 1):
    a, b = nms[k], nms[k + 1]
    t0, t1 = rows[a]["e"], rows[b]["e"]
    seg = [r for r in rows[a + 1:b + 1]]
    ev = sorted([(r["s"], 1, i) for i, r in enumerate(seg)] + [(r["e"], -1, i) for i, r in enumerate(seg)])
    active, last = set(), t0
    for t, d, i in ev:
        if t > last:
            n = sum(1 for j in active if "conv" in seg[j]["Kernel_Name"])
            tot[n] += t - last
            mid = (last + t) / 2
            dec = min(9, int(10 * (mid - t0) / (t1 - t0)))
            for j in active:
                deciles[dec][short(seg[j]["Kernel_Name"])] += (t - last) / max(1, len(active))
            if not active:
                deciles[dec]["(idle)"] += t - last
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
        last = t
    nsteps += 1
span = sum(tot.values())
print(f"{nsteps} steps, {span / nsteps / 1e3:.1f} us per step")
for n in sorted(tot):
    print(f"  {n} conv kernels running: {100.0 * tot[n] / span:5.1f} %")
for d in range(10):
    tt = sum(deciles[d].values())
    top = ", ".join(f"{k} {100.0 * v / tt:.0f}%" for k, v in deciles[d].most_common(4))
    print(f"  step decile {d}: {top}")
