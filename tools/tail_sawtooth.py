#!/usr/bin/env python3
"""How much of a conv launch is its tail?  One conv shape, one tile, timed over a range of batch sizes: the number of
workgroup "rounds" (tiles / resident workgroups) sweeps through integers, and the TFLOP/s curve shows what the
partially filled last round costs (a sawtooth) on top of the per-launch fixed cost (a slow rise).
   python tools/tail_sawtooth.py --cin 128 --cout 256 --size 3 --s 52 --tile 10 --batches 48:80:2"""
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
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--size", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--s", type=int, default=52)
    ap.add_argument("--tile", type=int, default=10)
    ap.add_argument("--batches", default="48:80:2")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lo, hi, st = (int(v) for v in a.batches.split(":"))
    p = mini_program(a.cin, [dict(filters=a.cout, size=a.size, stride=a.stride)],
                     [dict(filters=32, size=1), dict(filters=32, size=1), dict(filters=32, size=1)])
    w = synthetic_weights(p, seed=1)
    net = runtime.Net(p)
    net.load_weights(w)
    if a.tile >= 0:
        net.set_tile(0, a.tile)
    net.plan(hi, a.s, _lib.Y3_DTYPE_F32)
    bm, bn = _lib.TILES[a.tile][:2] if a.tile >= 0 else (64, 128)
    so = a.s // a.stride
    xall = torch.randn((hi, a.s, a.s, a.cin), device="cuda")
    print(f"# k{a.size}s{a.stride} {a.cin}->{a.cout} @{so}, tile {a.tile} ({bm}x{bn})")
    print("# batch  tiles  us      TF/s")
    for B in range(lo, hi + 1, st):
        x = xall[:B].contiguous()
        for _ in range(3):
            net.forward(x)
        best = None
        for _ in range(a.reps):
            ms = net.profile_convs(x)
            best = ms if best is None else np.minimum(best, ms)
        fl = 2.0 * a.size ** 2 * a.cin * a.cout * so * so * B
        tiles = -(-B * so * so // bm) * (-(-a.cout // bn))
        print(f"{B:5d} {tiles:6d} {best[0] * 1e3:8.1f} {fl / best[0] / 1e9:8.1f}", flush=True)


if __name__ == "__main__":
    main()
