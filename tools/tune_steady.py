#!/usr/bin/env python3
"""Coordinate-descent tile tuning in the regime that is actually benchmarked.

tools/tune_tiles.py times every conv launch in isolation (hipEvents around each launch).  The step that bench.py
measures differs from that in two ways: the chip sits at its power limit, and (plane-split / bf16 modes) several
sub-batch lanes run concurrently and fill each other's tails.  This tool starts from the existing table and, one conv
signature at a time (largest time share first), tries every legal candidate tile for all convs of that signature,
measures the WHOLE conv stack over --steps forwards with the table's lanes, and keeps a change only if it is faster
by more than --min-gain.  Usage: python tools/tune_steady.py --dtype f32x2 [--batch 64] [--write f32x2_b64_s416.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402

CANDIDATES = {
    "f32": [3, 4, 5, 6, 9, 10, 11, 12, 17, 18, 23, 26, 27, 29, 31, 32, 34],
    "f32x2": [0, 1, 2, 3, 4, 8, 12, 26, 27],
    "f32x3": [0, 1, 2, 3, 4, 5, 8, 12, 26, 27],
    "bf16": [0, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 17, 18, 19, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32x2", choices=list(CANDIDATES))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--min-gain", type=float, default=0.004)
    ap.add_argument("--lanes", type=int, default=0, help="0: the table's")
    ap.add_argument("--write", default="")
    ap.add_argument("--only", default="", help="comma-separated substrings: tune only the signatures containing one of them")
    ap.add_argument("--tiles", default="", help="comma-separated candidate tile ids (default: the mode's list)")
    a = ap.parse_args()
    B, S = a.batch, a.image_size
    dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3, "f32x2": _lib.Y3_DTYPE_F32X2}[a.dtype]
    TL = {"f32": None, "bf16": _lib.TILES_BF16, "f32x3": _lib.TILES_X3, "f32x2": _lib.TILES_X3}[a.dtype]
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p))
    net.plan(B, S, dt)                               # applies the existing table (tiles + lanes)
    if a.lanes:
        net.set_lanes(a.lanes)
    setter = {"f32": net.set_tile, "bf16": net.set_tile_bf16, "f32x3": net.set_tile_x3, "f32x2": net.set_tile_x2}[a.dtype]
    path = os.path.join(ROOT, "yolo-v3-tf2_amd", "tuning", f"{a.dtype}_b{B}_s{S}.json")
    doc = json.load(open(path)) if os.path.exists(path) else {"batch": B, "image_size": S, "tiles": {}}
    doc["dtype"] = a.dtype
    table = dict(doc.get("tiles", {}))
    x = torch.rand((B, S, S, 3), device="cuda")
    sigs = {}
    for slot, o in enumerate(net.conv_ops):
        if o.cin == 3:
            continue
        sig = net.conv_signature(o, S)
        ho = S // o.out_div
        d = sigs.setdefault(sig, {"slots": [], "flops": 0.0, "op": o})
        d["slots"].append(slot)
        d["flops"] += 2.0 * o.size ** 2 * o.cin * o.cout * ho * ho

    def measure():
        for _ in range(2):
            net.forward(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.steps):
            net.forward(x)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    def apply(sig, tile):
        for slot in sigs[sig]["slots"]:
            setter(slot, tile)

    best = measure()
    print(f"start: {best:.3f} ms  ({B / best * 1e3:.0f} img/s)", flush=True)
    # The chip warms up over the run (round 3: 10.07 -> 10.17 ms with nothing changed), so a candidate is compared with
    # the CURRENT tile measured right before it, and a winner must win a second, re-measured pair.
    cands = [int(v) for v in a.tiles.split(",")] if a.tiles else CANDIDATES[a.dtype]
    cands = [t for t in cands if _lib.tile_built(dt, t)]      # (round 5: only tiles some plan selects are built; new candidates need a switch case)
    only = [v for v in a.only.split(",") if v]
    for sig in sorted(sigs, key=lambda s: -sigs[s]["flops"]):
        if only and not any(v in sig for v in only):
            continue
        cur = int(table.get(sig, -1))
        apply(sig, cur)
        ref = measure()
        cand = {}
        for t in cands:
            if t == cur:
                continue
            try:
                apply(sig, t)
            except runtime.Y3Error:
                continue                             # tile does not fit this conv
            cand[t] = measure()
        apply(sig, cur)
        ref = min(ref, measure())
        if cand:
            t = min(cand, key=cand.get)
            if cand[t] < ref * (1.0 - a.min_gain):
                apply(sig, t)
                again = measure()
                apply(sig, cur)
                ref2 = measure()
                if again < ref2 * (1.0 - a.min_gain):
                    print(f"  {sig}: tile {cur} -> {t}: {ref2:.3f} -> {again:.3f} ms", flush=True)
                    cur = t
        apply(sig, cur)
        table[sig] = cur
        print(f"  {sig}: keeps tile {cur} (ref {ref:.3f} ms; best other "
              f"{min(cand.values()) if cand else float('nan'):.3f} ms)", flush=True)   # progress line
    final = measure()
    print(f"final: {final:.3f} ms  ({B / final * 1e3:.0f} img/s)", flush=True)
    if a.write:
        doc["tiles"] = table
        with open(os.path.join(ROOT, "yolo-v3-tf2_amd", "tuning", a.write), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
