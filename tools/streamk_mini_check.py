#!/usr/bin/env python3
"""Debug aid for the stream-K tiles: a three-conv program, outputs pre-filled with NaN, compared row by row with the
classic tile for a few workgroup counts (GPU).  Shows at once whether cut tiles are left unfinished or summed wrongly."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt
from yolo_v3_tf2_amd.weights import synthetic_weights
from tests.helpers import mini_program
B, S, cin = 3, 14, 32
p = mini_program(cin, [], [dict(filters=64, size=3, stride=2), dict(filters=64, size=3), dict(filters=64, size=1)])
w = synthetic_weights(p, seed=3)
x = torch.from_numpy(np.random.default_rng(1).standard_normal((B, S, S, cin)).astype(np.float32)).cuda()
def run(tile, grid):
    net = rt.Net(p); net.load_weights(w)
    net.set_sk_grid(grid); net.plan(B, S)
    for slot in range(len(net.conv_ops)): net.set_tile(slot, tile)
    outs = [torch.full((B, S // o.out_div, S // o.out_div, o.cout), float("nan"), device="cuda") for o in net.conv_ops]
    net.forward(x, out=outs)
    torch.cuda.synchronize()
    return outs
base = run(11, 0)
for grid in (1, 2, 3):
    got = run(34, grid)
    for k, (a, b) in enumerate(zip(base, got)):
        nan = torch.isnan(b)
        d = (a - b).abs()
        bad = (d > 1e-4) | nan
        rows = bad.reshape(-1, b.shape[-1]).any(1).nonzero().flatten()
        print(f"grid {grid} out{k}: nan {int(nan.sum())} bad elems {int(bad.sum())} of {b.numel()}  bad rows: {rows[:8].tolist()}..{rows[-3:].tolist() if len(rows) else []} n={len(rows)}")
        if len(rows):
            r = int(rows[0]); af = a.reshape(-1, b.shape[-1]); bf = b.reshape(-1, b.shape[-1])
            print("   row", r, "ref", af[r, :4].tolist(), "got", bf[r, :4].tolist(), "ratio", (bf[r, :4] / af[r, :4]).tolist())
