#!/usr/bin/env python3
"""A/B on one box, one process: the conv stack of the headline workload with the fused stem kernel on and off
(alternating, interleaved rounds), plus the time of the stem alone (conv0 + conv1 launches vs the one fused launch)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p))
    B, S = a.batch, a.image_size
    net.plan(B, S)
    x = torch.rand((B, S, S, 3), device="cuda")
    g = [torch.empty((B, s, s, 3, 85), device="cuda") for s in net.grid_sizes()]
    res = {True: [], False: []}
    for r in range(a.rounds):
        for fused in (False, True):
            net.set_stem_fusion(fused)
            for _ in range(3):
                net.forward(x, out=g)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                net.forward(x, out=g)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            res[fused].append(ms)
            lay = net.profile_convs(x)
            print(f"round {r} fused={int(fused)} conv stack {ms:.3f} ms  {net.flops_per_image() * B / ms / 1e9:.1f} TF/s   "
                  f"conv0 {lay[0]:.3f} conv1 {lay[1]:.3f} conv2 {lay[2]:.3f} ms", flush=True)
    for k, v in res.items():
        print(f"fused={int(k)}: min {min(v):.3f} ms  mean {sum(v) / len(v):.3f} ms")


if __name__ == "__main__":
    main()
