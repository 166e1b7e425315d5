#!/usr/bin/env python3
"""A/B of the residual-prefetch tiles (41..45) against their base tiles on the shortcut convs of the headline table:
whole conv stack, alternating, and per conv (GPU).  Output kept as profiles/r02_residual_prefetch_ab.txt."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt, PACKAGE_DIR
from yolo_v3_tf2_amd.graph import load_program
from yolo_v3_tf2_amd.weights import synthetic_weights
p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
w = synthetic_weights(p, seed=4321)
B, S = 64, 416
x = torch.rand((B, S, S, 3), device="cuda")
net = rt.Net(p); net.load_weights(w); net.plan(B, S)
table = json.load(open(os.path.join(PACKAGE_DIR, "tuning", "f32_b64_s416.json")))["tiles"]
grids = [torch.empty((B, g, g, 3, 85), device="cuda") for g in net.grid_sizes()]
MAP = {10: 41, 31: 42, 27: 43, 11: 44, 26: 45}
def apply(respf):
    for slot, o in enumerate(net.conv_ops):
        if o.cin == 3: continue
        t = int(table.get(net.conv_signature(o, S), -1))
        if o.residual >= 0:
            if t < 0: t = 10 if o.cout % 128 == 0 else 11      # the library heuristic's choices
            if respf: t = MAP.get(t, t)
        net.set_tile(slot, t)
def measure(n=20):
    for _ in range(3): net.forward(x, out=grids)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): net.forward(x, out=grids)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    for respf in (0, 1):
        apply(respf)
        ms = measure()
        print(f"residual prefetch {respf}: conv stack {ms:.3f} ms  {net.flops_per_image()*B/ms/1e9:.1f} TF/s", flush=True)
res = {}
for respf in (0, 1):
    apply(respf)
    res[respf] = np.minimum(net.profile_convs(x), net.profile_convs(x))
for o, a, b in zip(net.conv_ops, res[0], res[1]):
    if o.residual < 0: continue
    ho = S // o.out_div
    fl = 2.0 * o.size * o.size * o.cin * o.cout * ho * ho * B
    print(f"conv{o.conv_index:<3d} k{o.size}s{o.stride} {o.cin:>4d}->{o.cout:<4d} @{ho:<3d}  base {fl/a/1e9:7.1f}  prefetch {fl/b/1e9:7.1f} TF/s")
