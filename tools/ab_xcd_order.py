#!/usr/bin/env python3
"""A/B of the fp32 conv tile placement on the 8 XCDs (y3_net_set_xcd_mode 0 / 1): whole conv stack, alternating, and
per conv (GPU).  Output kept as profiles/r02_xcd_order_ab_run*.txt."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd
from yolo_v3_tf2_amd import runtime as rt
from yolo_v3_tf2_amd.graph import load_program
from yolo_v3_tf2_amd.weights import synthetic_weights
p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
w = synthetic_weights(p, seed=4321)
B, S = 64, 416
x = torch.rand((B, S, S, 3), device="cuda")
net = rt.Net(p); net.load_weights(w); net.plan(B, S)
grids = [torch.empty((B, g, g, 3, 85), device="cuda") for g in net.grid_sizes()]
res = {}
for rep in range(2):
    for mode in (0, 1):
        net.set_xcd_mode(mode)
        for _ in range(3): net.forward(x, out=grids)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): net.forward(x, out=grids)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"xcd_mode {mode}: conv stack {ms:.3f} ms  {net.flops_per_image()*B/ms/1e9:.1f} TF/s", flush=True)
for mode in (0, 1):
    net.set_xcd_mode(mode)
    ms = np.minimum(net.profile_convs(x), net.profile_convs(x))
    res[mode] = ms
for o, a, b in zip(net.conv_ops, res[0], res[1]):
    ho = S // o.out_div
    fl = 2.0 * o.size * o.size * o.cin * o.cout * ho * ho * B
    print(f"conv{o.conv_index:<3d} k{o.size}s{o.stride} {o.cin:>4d}->{o.cout:<4d} @{ho:<3d}  mode0 {fl/a/1e9:7.1f}  mode1 {fl/b/1e9:7.1f} TF/s  {'+' if b < a*0.99 else '-' if b > a*1.01 else ''}")
