#!/usr/bin/env python3
"""Print a window of a rocprofv3 --kernel-trace CSV as a per-queue timeline (us relative to the window start).
usage: tools/timeline_dump.py <kernel_trace.csv> <first_step> <n_kernels>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
first = int(sys.argv[2])
n = int(sys.argv[3])
i0 = marks[first] + 1
t0 = rows[i0]["s"]
queues = sorted({r["Queue_Id"] for r in rows[i0:i0 + n]})
print("queues:", queues)
for r in rows[i0:i0 + n]:
    m = re.search(r"(conv_\w+|decode_kernel|nms_kernel|pack_kernel)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:20]
    g = int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
    q = queues.index(r["Queue_Id"])
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - t0) / 1e3:9.1f}  {(r['e'] - r['s']) / 1e3:7.1f} us  q{q} {'    ' * q}{name} wgs={g}")
