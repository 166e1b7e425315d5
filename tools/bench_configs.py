#!/usr/bin/env python3
"""BASELINE config 2 (Darknet-53 backbone forward only, batch 32, 416x416, 1 GPU): images/s and conv TFLOP/s per mode.
(Configs 3-5 go through bench.py: --image-size 608; default; --dtype bf16 --batch 128 --graph.)"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.graph import build_program, find_config_root, load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    mf = os.path.join(ROOT, "config/models/yolov3/model.yaml")
    cfg = yaml.safe_load(open(mf))
    bb_cfg = [c for c in cfg["sub_models_configs"] if c["name"] == "backbone"]
    bb = build_program(bb_cfg, "backbone", 0, find_config_root(mf, bb_cfg))
    w = synthetic_weights(load_program(mf, 80))
    bw = {k: v for k, v in w.items() if int(k.split(".")[0][4:]) < 52}
    B, S, steps = 32, 416, 30
    x = torch.rand((B, S, S, 3), device="cuda")
    net = runtime.Net(bb)
    net.load_weights(bw)
    # the config names fp32; the other modes are listed for comparison
    for tag, dt in (("f32", _lib.Y3_DTYPE_F32), ("f32x3", _lib.Y3_DTYPE_F32X3), ("f32x2", _lib.Y3_DTYPE_F32X2), ("bf16", _lib.Y3_DTYPE_BF16)):
        net.plan(B, S, dt)
        for _ in range(3):
            net.forward(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            net.forward(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        print(f"config 2 backbone-only B={B} {S}x{S} {tag:6s} {ms:7.3f} ms/batch  {B / ms * 1e3:8.1f} img/s  "
              f"{net.flops_per_image() * B / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
