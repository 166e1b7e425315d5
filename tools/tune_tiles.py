#!/usr/bin/env python3
"""Per-conv tile sweep on the real network (GPU).  Times every conv launch (hipEvents inside
y3_net_profile_convs) for every legal tile id and prints the table; --write stores the winners in
yolo-v3-tf2_amd/tuning/<name>.json keyed by the conv's shape signature."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402
from yolo_v3_tf2_amd import runtime  # noqa: E402
from yolo_v3_tf2_amd._lib import TILE_NAMES, TILES  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tiles", type=str, default="all")
    ap.add_argument("--write", type=str, default="")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32x3", "f32x2"], default="f32")
    a = ap.parse_args()
    B, S = a.batch, a.image_size
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    w = synthetic_weights(p)
    os.environ["Y3_NO_TUNING"] = "1"
    net = runtime.Net(p)
    net.load_weights(w)
    from yolo_v3_tf2_amd import _lib
    bf = a.dtype in ("bf16", "f32x3", "f32x2")   # tile tables with a BK column
    x2 = a.dtype == "f32x2"
    x3 = a.dtype in ("f32x3", "f32x2")            # the plane-split modes share one tile table
    net.plan(B, S, {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3,
                    "f32x2": _lib.Y3_DTYPE_F32X2}[a.dtype])
    x = torch.rand((B, S, S, 3), device="cuda")
    TL = _lib.TILES_X3 if x3 else _lib.TILES_BF16 if bf else TILES
    TN_ = [f"{bm}x{bn}w{w}k{k}" + ("d" if (i >= 8 and not x3) else "") + ("s1" if (x3 and i in (9, 10, 11, 13, 14, 15)) else "")
           + ("i" if (x3 and 20 <= i < 26) else "") + ("s3" if (x3 and 30 <= i < 34) else "") for i, (bm, bn, w, k) in enumerate(TL)] if bf else TILE_NAMES
    tiles = [int(t) for t in a.tiles.split(",")] if a.tiles != "all" else list(range(len(TL)))
    dt_id = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3, "f32x2": _lib.Y3_DTYPE_F32X2}[a.dtype]
    missing = [t for t in tiles if not _lib.tile_built(dt_id, t)]
    if missing and a.tiles != "all":
        sys.exit(f"tile ids {missing} do not exist in this library (retired ids: y3_tile_built)")
    tiles = [t for t in tiles if _lib.tile_built(dt_id, t)]
    res = {}
    for t in [-1] + tiles:
        bn = TL[t][1] if t >= 0 else 0
        ok = []
        for slot, o in enumerate(net.conv_ops):
            cp = (o.cout + 63) // 64 * 64 if x3 else (o.cout + 31) // 32 * 32
            legal = t < 0 or (o.cin != 3 and cp % bn == 0 and not (not bf and t in (20, 21, 22, 25) and o.src1 >= 0))
            if bf and t >= 0:
                legal = legal and o.cin % TL[t][3] == 0 and (o.src1 < 0 or o.c0 % TL[t][3] == 0)
            if a.dtype == "f32" and t == 33:    # the weight-resident kernel: 3x3 / stride 1, 32 input channels
                legal = legal and o.size == 3 and o.stride == 1 and o.src1 < 0 and o.cin == 32 and o.cout % 64 == 0
            if a.dtype == "bf16" and t == 32:   # the weight-resident kernel: 3x3 / stride 1, 32 or 64 input channels, bf16 output
                legal = legal and o.size == 3 and o.stride == 1 and o.src1 < 0 and o.cin in (32, 64) and o.cout % 64 == 0 and o.dst not in p.outputs
            ok.append(legal)
            (net.set_tile_x2 if x2 else net.set_tile_x3 if x3 else net.set_tile_bf16 if bf else net.set_tile)(slot, t if legal else -1)
        best = None
        for _ in range(a.reps):
            ms = net.profile_convs(x)
            best = ms if best is None else np.minimum(best, ms)
        res[t] = (best, ok)
    names = {-1: "heur"}
    names.update({t: TN_[t] for t in tiles})
    print("conv  shape                      " + " ".join(f"{names[t]:>11s}" for t in res))
    winners = {}
    sig_ms = {}
    tot_best = tot_heur = 0.0
    for slot, o in enumerate(net.conv_ops):
        ho = S // o.out_div
        fl = 2.0 * o.size ** 2 * o.cin * o.cout * ho * ho * B
        row = []
        bt, bms = -1, res[-1][0][slot]
        for t in res:
            ms, ok = res[t]
            if ok[slot]:
                row.append(f"{fl / ms[slot] / 1e9:11.1f}")
                if t >= 0 and ms[slot] < bms * 0.99:
                    bt, bms = t, ms[slot]
            else:
                row.append("          -")
        sig = f"k{o.size}s{o.stride}_c{o.cin}_n{o.cout}_h{ho}_r{int(o.residual >= 0)}_u{int(o.src1 >= 0)}"
        # several convs share a signature: keep the tile with the lowest summed time over all of them
        sig_ms.setdefault(sig, {})
        for t in res:
            ms, ok = res[t]
            if ok[slot]:
                sig_ms[sig][t] = sig_ms[sig].get(t, 0.0) + float(ms[slot])
        tot_best += bms
        tot_heur += res[-1][0][slot]
        print(f"{o.conv_index:<4d}  {sig:<26s} " + " ".join(row) + f"   best={names.get(bt, 'heur')}")
    print(f"sum heuristic {tot_heur:.3f} ms   sum best {tot_best:.3f} ms   "
          f"({net.flops_per_image() * B / tot_best / 1e9:.1f} TF/s)")
    for sig, d in sig_ms.items():
        best_t = min(d, key=d.get)
        winners[sig] = best_t if d[best_t] < d.get(-1, 1e30) * 0.99 else -1
    if a.write:
        d = os.path.join(ROOT, "yolo-v3-tf2_amd", "tuning")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, a.write)
        doc = {"dtype": a.dtype, "batch": B, "image_size": S, "tiles": winners}
        if os.path.exists(path):     # keep the keys other tools own (lanes: tools/lanes_sweep.py)
            with open(path) as f:
                old = json.load(f)
            doc.update({k: v for k, v in old.items() if k not in doc})
        with open(path, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
