#!/usr/bin/env python3
"""Conv-stack TFLOP/s as a function of the batch (GPU).  The per-launch drain (the last workgroups of a kernel leave
CUs idle) is a fixed cost per launch, so its share shrinks as the batch grows: this sweep separates it from the
steady-state efficiency of the K loop.  Uses the tile table of --table-batch for every batch."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import PACKAGE_DIR, _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,32,64,96,128,176")
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--table-batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=1, help="sub-batch lanes of every forward (1: the sweep of round 2)")
    a = ap.parse_args()
    S = a.image_size
    dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3, "f32x2": _lib.Y3_DTYPE_F32X2}[a.dtype]
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    w = synthetic_weights(p)
    net = runtime.Net(p)
    net.load_weights(w)
    path = os.path.join(PACKAGE_DIR, "tuning", f"{a.dtype}_b{a.table_batch}_s{S}.json")
    table = json.load(open(path))["tiles"] if os.path.exists(path) else {}
    setter = {"f32": net.set_tile, "bf16": net.set_tile_bf16, "f32x3": net.set_tile_x3, "f32x2": net.set_tile_x2}[a.dtype]
    for B in [int(b) for b in a.batches.split(",")]:
        net.plan(B, S, dt)
        net.set_lanes(a.lanes)
        for slot, o in enumerate(net.conv_ops):
            if o.cin != 3:
                setter(slot, int(table.get(net.conv_signature(o, S), -1)))
        x = torch.rand((B, S, S, 3), device="cuda")
        grids = [torch.empty((B, g, g, 3, 85), device="cuda") for g in net.grid_sizes()]
        for _ in range(3):
            net.forward(x, out=grids)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            net.forward(x, out=grids)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print(f"batch {B:4d} lanes {a.lanes}  conv stack {ms:8.3f} ms  {net.flops_per_image() * B / ms / 1e9:7.1f} TF/s  {B / ms * 1e3:8.1f} img/s", flush=True)
        del x, grids


if __name__ == "__main__":
    main()
