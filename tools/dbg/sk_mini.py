import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt
from yolo_v3_tf2_amd.weights import synthetic_weights
from tests.helpers import mini_program
def run(p, w, x, tile, grid, B, S):
    net = rt.Net(p); net.load_weights(w)
    net.set_sk_grid(grid); net.plan(B, S)
    for slot in range(len(net.conv_ops)): net.set_tile(slot, tile)
    g = [t.clone() for t in net.forward(x)]
    torch.cuda.synchronize()
    return g
for cin in (32, 64):
    for S, B in ((14, 3), (48, 3)):
        p = mini_program(cin, [], [dict(filters=64, size=3, stride=2), dict(filters=64, size=3), dict(filters=64, size=1)])
        w = synthetic_weights(p, seed=3)
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((B, S, S, cin)).astype(np.float32)).cuda()
        base = run(p, w, x, 11, 0, B, S)
        for grid in (1, 2, 3, 8, 9, 16, 100, 0):
            got = run(p, w, x, 34, grid, B, S)
            d = [float((a - b).abs().max()) for a, b in zip(base, got)]
            print(f"cin {cin} S {S} B {B} grid {grid:5d}: maxdiff s2 {d[0]:.2e} s1 {d[1]:.2e} 1x1 {d[2]:.2e}", flush=True)
