#!/usr/bin/env python3
"""One conv shape, one tile, many launches (GPU): for rocprofv3 counter passes on a single kernel configuration.
   python tools/one_conv.py --dtype bf16 --cin 256 --cout 512 --size 3 --s 26 --batch 128 --tile 20 --reps 50"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402
from tests.helpers import mini_program  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--size", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--s", type=int, default=26)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    # conv under test feeds three 1x1 heads, so that its own output stays in the mode's format (bf16)
    p = mini_program(a.cin, [dict(filters=a.cout, size=a.size, stride=a.stride)],
                     [dict(filters=64, size=1), dict(filters=64, size=1), dict(filters=64, size=1)])
    w = synthetic_weights(p, seed=1)
    net = runtime.Net(p)
    net.load_weights(w)
    dt = _lib.Y3_DTYPE_BF16 if a.dtype == "bf16" else _lib.Y3_DTYPE_F32
    if a.tile >= 0:
        (net.set_tile_bf16 if a.dtype == "bf16" else net.set_tile)(0, a.tile)
    net.plan(a.batch, a.s, dt)
    x = torch.randn((a.batch, a.s, a.s, a.cin), device="cuda")
    if a.dtype == "bf16":
        x = x.to(torch.bfloat16)
    for _ in range(3):
        net.forward(x)
    best = None
    for _ in range(a.reps):
        ms = net.profile_convs(x)
        best = ms if best is None else np.minimum(best, ms)
    so = a.s // a.stride
    fl = 2.0 * a.size ** 2 * a.cin * a.cout * so * so * a.batch
    print(f"{a.dtype} k{a.size}s{a.stride} {a.cin}->{a.cout} @{so} B={a.batch} tile {a.tile}: {best[0] * 1e3:.1f} us  {fl / best[0] / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
