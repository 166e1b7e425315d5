#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into per-kernel summaries."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"conv_f32_mfmaILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)", name)
    if m:
        tm, tn, wr, wc, cat, st = map(int, m.groups())
        return f"conv_f32_mfma<{32*tm*wr}x{32*tn*wc},w{wr*wc},s{st}{',cat' if cat else ''}>"
    # template <TM, TN, WR, WC, CONCAT, STAGES, MINW, DMA, ...> (round 4; the probe / stream-K / residual-prefetch parameters are gone)
    m = re.search(r"conv_f32_mfma<(\d+), (\d+), (\d+), (\d+), (true|false), (\d+)(?:, (\d+), (\d+))?((?:, \d+)*)>", name)
    if m:
        tm, tn, wr, wc = (int(m.group(i)) for i in range(1, 5))
        dma = ",dma" if m.group(8) == "1" else ""
        extra = "".join(",x" + v.strip() for v in m.group(9).split(",") if v.strip() and v.strip() != "0")
        return f"conv_f32_mfma<{32*tm*wr}x{32*tn*wc},w{wr*wc},s{m.group(6)}{dma}{extra}{',cat' if m.group(5) == 'true' else ''}>"
    m = re.search(r"conv_bf16_mfma<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false)(?:, (true|false), (\d+), (true|false))?", name)
    if m:
        tm, tn, wr, wc, bk = (int(m.group(i)) for i in range(1, 6))
        return (f"conv_bf16_mfma<{32*tm*wr}x{32*tn*wc},w{wr*wc},k{bk}{',dma' if m.group(8) == 'true' else ''}"
                f"{',16x16x32' if m.group(10) == 'true' else ''}{',cat' if m.group(6) == 'true' else ''}"
                f"{',f32out' if m.group(7) == 'true' else ''}>")
    m = re.search(r"conv_f32x3_mfma<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+)", name)
    if m:
        npl, tm, tn, wr, wc, bk = (int(m.group(i)) for i in range(1, 7))
        return (f"conv_f32x{npl}_mfma<{32*tm*wr}x{32*tn*wc},w{wr*wc},k{bk},s{m.group(9)}{',cat' if m.group(7) == 'true' else ''}"
                f"{',f32out' if m.group(8) == 'true' else ''}>")
    m = re.search(r"conv_first_f32x3<\d+, (\d)>", name)
    if m:
        return f"conv_first_f32x{m.group(1)}"
    m = re.search(r"conv_bf16_rs<(\d+), (\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        tm, tn, wr, wc = (int(m.group(i)) for i in range(1, 5))
        return f"conv_bf16_rs<{32*tm*wr}x{32*tn*wc},w{wr*wc}{',16x16x32' if m.group(5) == 'true' else ''}>"
    m = re.search(r"conv3x3_res_bf16<(\d+)>", name)
    if m:
        return f"conv_res3x3_bf16<cin{m.group(1)}>"
    if "conv3x3_res_f32" in name:
        return "conv_res3x3_f32"
    for k in ("conv_head_decode_f32", "conv_stem_f32", "conv_stem_bf16", "conv_first_f32x3", "conv_first_f32", "conv_first_bf16", "decode_kernel", "nms_kernel", "pack_kernel", "class_scores"):
        if k in name:
            return k
    return name[:60]


def main(out):
    # kernel stats
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, "summary_kernel_stats.txt"), "w") as o:
            o.write(f"# rocprofv3 --kernel-trace --stats ({os.path.basename(f)})\n")
            o.write(f"{'kernel':<52s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}\n")
            for r in rows:
                o.write(f"{short(r['Name']):<52s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
                        f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} "
                        f"{float(r['Percentage']):6.2f}\n")
    # conv kernels on the wall clock: with sub-batch lanes the launches of different lanes overlap, so the summed durations above
    # exceed the elapsed time -- report both per step (steps = NMS launches; the steps after the first two, warm-up excluded)
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
        rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
        rows.sort()
        marks = [e for s_, e, n in rows if "nms_kernel" in n]
        if len(marks) >= 4:
            t0, t1, nsteps = marks[1], marks[-1], len(marks) - 2
            conv = [(s_, e) for s_, e, n in rows if "conv_" in n and s_ >= t0 and e <= t1]
            total = sum(e - s_ for s_, e in conv)
            union, cur_s, cur_e = 0, None, None
            for s_, e in conv:
                if cur_e is None or s_ > cur_e:
                    if cur_e is not None:
                        union += cur_e - cur_s
                    cur_s, cur_e = s_, e
                else:
                    cur_e = max(cur_e, e)
            if cur_e is not None:
                union += cur_e - cur_s
            with open(os.path.join(out, "summary_kernel_stats.txt"), "a") as o:
                o.write(f"# conv kernels over the last {nsteps} steps of the traced run: summed durations {total / 1e6 / nsteps:.3f} ms per step, "
                        f"time with at least one conv kernel running {union / 1e6 / nsteps:.3f} ms per step, "
                        f"{len(conv) / nsteps:.1f} launches per step (sub-batch lanes overlap: the second figure is the one to hold against "
                        f"bench.py's conv-stack time)\n")
        break
    # PMC passes: sum counters per kernel name
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    passes = [p for p in sorted(glob.glob(os.path.join(out, "pmc*"))) if os.path.isdir(p)]
    for p in passes:
        seen = set()
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (p, r["Dispatch_Id"])
                if p == passes[0] and key not in seen:   # dispatch count from the first pass present
                    seen.add(key)
                    calls[k] += 1
    if agg:
        names = sorted({c for k in agg for c in agg[k]})
        with open(os.path.join(out, "summary_pmc.txt"), "w") as o:
            o.write("# rocprofv3 --pmc, counter values summed over all dispatches of each kernel (and over XCDs/SEs)\n")
            for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
                o.write(f"{k}  dispatches={calls[k]}\n")
                for c in names:
                    if c in agg[k]:
                        o.write(f"    {c:<34s} {agg[k][c]:.6g}\n")
    # derived per-kernel figures (guides/MI355X_MICROARCH.md: FETCH_SIZE is in KiB and reads 1/2 of a wide
    # coalesced stream on gfx950 -> doubled below; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD;
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    if agg:
        with open(os.path.join(out, "summary_derived.txt"), "w") as o:
            o.write("# derived: per kernel, all dispatches of the run\n")
            o.write(f"{'kernel':<52s} {'n':>4s} {'mfma_busy':>9s} {'waves/simd':>10s} {'fetchx2_MB/launch':>18s} "
                    f"{'write_MB/launch':>16s} {'l2_hit':>7s}\n")
            for k in sorted(agg, key=lambda k: -(agg[k].get("SQ_WAVE_CYCLES", 0) or agg[k].get("FETCH_SIZE", 0))):
                a = agg[k]
                n = max(calls[k], 1)
                cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
                if cyc <= 0 and not (a.get("FETCH_SIZE") or a.get("WRITE_SIZE")):
                    continue
                busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc > 0 else float("nan")
                occ = a.get("SQ_WAVE_CYCLES", 0) * 4 / (cyc * 1024) if cyc > 0 else float("nan")
                fetch = a.get("FETCH_SIZE", 0) * 1024 * 2 / n / 1e6
                write = a.get("WRITE_SIZE", 0) * 1024 / n / 1e6
                hm = a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0)
                hit = a.get("TCC_HIT_sum", 0) / hm if hm else 0
                o.write(f"{k:<52s} {n:>4d} {busy:9.3f} {occ:10.2f} {fetch:18.1f} {write:16.1f} {hit:7.3f}\n")
    # HBM traffic of one conv-stack pass (all conv kernels of one step): FETCH_SIZE (KiB, x2 gfx950 correction) + WRITE_SIZE
    if agg:
        import json
        conv = [k for k in agg if k.startswith("conv_")]
        # steps of the run: the NMS kernel runs once per step whatever the number of sub-batch lanes (the first conv / the
        # stem kernel run once per LANE: counting those halved the bf16 figure of the two-lane plan until round 3)
        once = [k for k in calls if k.startswith("nms_kernel")] or [k for k in conv if "first" in k or "stem" in k]
        steps = max(calls[k] for k in once) if once else 1
        fetch = sum(agg[k].get("FETCH_SIZE", 0) for k in conv) * 1024 * 2 / steps
        write = sum(agg[k].get("WRITE_SIZE", 0) for k in conv) * 1024 / steps
        with open(os.path.join(out, "summary_traffic.json"), "w") as o:
            json.dump({"steps_profiled": steps, "conv_stack_fetch_bytes_per_step": fetch,
                       "conv_stack_write_bytes_per_step": write, "conv_stack_hbm_bytes_per_step": fetch + write,
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled "
                               "(gfx950 reports half of a wide coalesced stream), summed over the conv kernels of one step"},
                      o, indent=1)
    for f in ("summary_kernel_stats.txt", "summary_derived.txt"):
        fp = os.path.join(out, f)
        if os.path.exists(fp):
            print(open(fp).read())


if __name__ == "__main__":
    main(sys.argv[1])
