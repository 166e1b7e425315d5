#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the conv stack against each layer's algorithmic bytes.

Reads the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh (gpurun_out/prof_<tag>/pmc3, pmc4: one row per dispatch),
takes the LAST forward of the run, maps its conv launches to the network's convs in launch order (the fused stem covers
conv0 + conv1 (+ conv2 in fp32 plans)) and prints, per layer, algorithmic read / write bytes next to the measured ones
(FETCH_SIZE in KiB, doubled: the gfx950 correction of MI355X_MICROARCH.md; WRITE_SIZE in KiB).
usage: tools/traffic_per_layer.py gpurun_out/prof_r03 [--batch 64] [--image-size 416] [--bytes 4]"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402


def conv_launches(path):
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path))
            if "conv_" in r["Kernel_Name"]]
    rows.sort()
    forwards, cur = [], None
    for _, k, v in rows:
        if "conv_stem" in k or "conv_first" in k:
            cur = []
            forwards.append(cur)
        if cur is not None:
            cur.append((k, v))
    return forwards


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--bytes", type=int, default=4, help="bytes per activation / weight element (4: fp32 plan, 2: bf16)")
    a = ap.parse_args()
    B, S, E = a.batch, a.image_size, a.bytes
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    convs = [o for o in p.ops if type(o).__name__ == "ConvOp"]
    ff = conv_launches(os.path.join(a.dir, "pmc3", "run_counter_collection.csv"))[-1]
    ww = conv_launches(os.path.join(a.dir, "pmc4", "run_counter_collection.csv"))[-1]
    if len(ff) != len(ww):
        sys.exit(f"the two passes disagree on the launches of a forward: {len(ff)} vs {len(ww)}")
    n = len(convs)
    fused = n - len(ff)                      # convs that run inside the first launch besides its own: 0 (conv_first), 1 or 2
    if not 0 <= fused <= 2:
        sys.exit(f"{len(ff)} conv launches per forward for {n} convs: not a plan this tool knows")
    groups = [list(range(0, fused + 1))] + [[i] for i in range(fused + 1, n)]

    def algo(ci):
        o = convs[ci]
        ho = S // o.out_div
        hi = ho * o.stride
        rd = B * hi * hi * o.cin * (4 if o.cin == 3 else E) + o.size ** 2 * o.cin * o.cout * E
        res = B * ho * ho * o.cout * E if o.residual >= 0 else 0
        out_e = 4 if o.cout == 255 else E     # head grids leave in fp32
        return rd, res, B * ho * ho * o.cout * out_e

    print(f"# {a.dir}: last forward, batch {B}, {S}x{S}; MB; algorithmic = input + weights (+ shortcut) read once, output written once")
    print(f"{'conv':<8s}{'shape':<26s}{'algo rd':>9s}{'algo wr':>9s}{'fetch':>9s}{'write':>9s}{'rd ratio':>9s}{'all ratio':>10s}")
    ta = tm = 0.0
    for (k, fv), (_, wv), g in zip(ff, ww, groups):
        if len(g) > 1:      # fused stem: image in; conv1's output out (conv3's shortcut), conv2's output out when it is inside
            rd = algo(g[0])[0] + sum(convs[c].size ** 2 * convs[c].cin * convs[c].cout * E for c in g[1:])
            wr = sum(algo(c)[2] for c in g[1:])
            sig = "stem conv" + "+".join(str(c) for c in g)
        else:
            r, res, wr = algo(g[0])
            rd = r + res
            o = convs[g[0]]
            sig = f"k{o.size}s{o.stride} {o.cin}->{o.cout}@{S // o.out_div} r{int(o.residual >= 0)}" + (" cat" if o.src1 >= 0 else "")
        fm, wm = fv * 1024 * 2 / 1e6, wv * 1024 / 1e6
        ta += rd + wr
        tm += (fm + wm) * 1e6
        print(f"{g[-1]:<8d}{sig:<26s}{rd / 1e6:9.1f}{wr / 1e6:9.1f}{fm:9.1f}{wm:9.1f}{fm / (rd / 1e6):9.2f}{(fm + wm) / ((rd + wr) / 1e6):10.2f}")
    print(f"total: algorithmic {ta / 1e9:.2f} GB, measured {tm / 1e9:.2f} GB, ratio {tm / ta:.2f}")


if __name__ == "__main__":
    main()
