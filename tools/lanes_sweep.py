#!/usr/bin/env python3
"""Conv-stack time per step for combinations of concurrent sub-batch lanes (y3_net_set_lanes) and block tiles.

Usage: python tools/lanes_sweep.py [--dtype f32x2] [--batch 64] [--image-size 416]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32x2", choices=["f32", "bf16", "f32x3", "f32x2"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--force", type=str, default="", help="comma list of tile ids to force on every conv they fit (one run each)")
    ap.add_argument("--early", type=str, default="", help="n_convs:chunk pairs for y3_net_set_early_chunk, e.g. 9:4,9:8,4:4")
    ap.add_argument("--lanes", type=str, default="1,2,3,4")
    a = ap.parse_args()
    dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3, "f32x2": _lib.Y3_DTYPE_F32X2}[a.dtype]
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p))
    x = torch.rand((a.batch, a.image_size, a.image_size, 3), device="cuda")
    setter = {"f32": net.set_tile, "bf16": net.set_tile_bf16, "f32x3": net.set_tile_x3, "f32x2": net.set_tile_x2}[a.dtype]

    def measure(label):
        for lanes in [int(v) for v in a.lanes.split(",")]:
            net.set_lanes(lanes)
            for _ in range(3):
                net.forward(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.steps):
                net.forward(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            print(f"{label:24s} lanes={lanes}  conv stack {ms:7.3f} ms  {a.batch / ms * 1e3:8.1f} img/s", flush=True)

    net.plan(a.batch, a.image_size, dt)
    measure("tuning table")
    for pair in [v for v in a.early.split(",") if v]:
        n_convs, chunk = (int(v) for v in pair.split(":"))
        net.set_early_chunk(n_convs, chunk)
        net.plan(a.batch, a.image_size, dt)
        measure(f"early {n_convs} convs x {chunk} img")
    net.set_early_chunk(0, 0)
    for t in [int(v) for v in a.force.split(",") if v]:
        net.plan(a.batch, a.image_size, dt)     # resets to the table
        for slot, o in enumerate(net.conv_ops):
            if o.cin == 3:
                continue
            try:
                setter(slot, t)
            except runtime.Y3Error:
                pass
        measure(f"tile {t} where it fits")


if __name__ == "__main__":
    main()
