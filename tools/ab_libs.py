#!/usr/bin/env python3
"""A/B of two builds of liby3hip.so on ONE box: alternating child processes (Y3_LIB_PATH), each timing the conv stack of
the headline workload.  Boxes differ by +-2.5 %, processes on one box by < 0.5 %.
   python tools/ab_libs.py yolo-v3-tf2_amd/lib/liby3hip.so yolo-v3-tf2_amd/lib/liby3hip_var.so [--dtype f32] [--rounds 3]
A contender may also be LIB.so@TABLE.json: that build with Y3_TUNING_FILE=TABLE.json (a tile table for the SAME dtype / batch / size),
and carry environment settings for its child process after a percent sign: LIB.so%Y3_LANE_STAGGER=1,OTHER=2 (tool knobs of the library)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime, _lib
from yolo_v3_tf2_amd.graph import load_program
from yolo_v3_tf2_amd.weights import synthetic_weights
p = load_program(os.path.join(%r, "config/models/yolov3/model.yaml"), 80)
B, S, dt, data, route = %d, %d, %r, %r, %r
w = synthetic_weights(p)
if data == "zeros":   # every operand of every MFMA is zero: no toggling in the matrix pipes or on the data paths (a power experiment)
    for k in w:
        if not k.endswith(".var"): w[k] = w[k] * 0
net = runtime.Net(p); net.load_weights(w)
net.plan(B, S, {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x2": _lib.Y3_DTYPE_F32X2, "f32x3": _lib.Y3_DTYPE_F32X3}[dt])
x = torch.rand((B, S, S, 3), device="cuda") if data != "zeros" else torch.zeros((B, S, S, 3), device="cuda")
g = [torch.empty((B, s, s, 3, 85), device="cuda") for s in net.grid_sizes()]
import numpy as np
from yolo_v3_tf2_amd.core.utils import get_anchors
anchors = get_anchors(os.path.join(%r, "datasets/coco2012/anchors.txt")).astype(np.float32)
step = (lambda: net.forward(x, out=g)) if route == "forward" else (lambda: net.forward_decode(x, anchors))   # decode: the route bench.py times (head convs decode their own tiles)
for _ in range(5): step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(30): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
print("RESULT %%.4f %%.2f" %% (ms, net.flops_per_image() * B / ms / 1e9), flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--route", default="forward", choices=["forward", "decode"], help="forward: y3_net_forward (grids written); decode: y3_net_forward_decode, the route bench.py times")
    ap.add_argument("--data", default="rand", choices=["rand", "zeros"], help="zeros: all-zero weights and images (what the step costs without data toggling: the power wall)")
    a = ap.parse_args()
    code = CHILD % (ROOT, ROOT, a.batch, a.image_size, a.dtype, a.data, a.route, ROOT)
    res = {lib: [] for lib in a.libs}
    for r in range(a.rounds):
        for lib in a.libs:
            spec, _, envs = lib.partition("%")
            so, _, table = spec.partition("@")
            env = dict(os.environ, Y3_LIB_PATH=os.path.abspath(so))
            for kv in [v for v in envs.split(",") if v]:
                env[kv.partition("=")[0]] = kv.partition("=")[2]
            if table:
                env["Y3_TUNING_FILE"] = os.path.abspath(table)
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(out.stdout[-500:], out.stderr[-1500:])
                sys.exit(1)
            ms, tf = line[0].split()[1:]
            res[lib].append(float(ms))
            print(f"round {r} {lib.replace(os.path.dirname(lib.partition(chr(64))[0]) + os.sep, str()):28s} conv stack {ms} ms  {tf} TF/s", flush=True)
    for lib, v in res.items():
        print(f"{lib.replace(os.path.dirname(lib.partition(chr(64))[0]) + os.sep, str()):28s} min {min(v):.3f} ms  mean {sum(v) / len(v):.3f} ms")


if __name__ == "__main__":
    main()
