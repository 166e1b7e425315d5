import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from yolo_v3_tf2_amd import runtime
from yolo_v3_tf2_amd.core.utils import get_anchors
anchors = get_anchors("datasets/coco2012/anchors.txt").astype(np.float32)
B = 64
grids = [torch.randn((B, g, g, 3, 85), device="cuda") for g in (13, 26, 52)]
for fn, name in ((runtime.yolo_decode_scores, "decode+scores (fused)"), (runtime.yolo_decode, "decode (with probs)")):
    for _ in range(3): fn(grids, anchors, 80)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fn(grids, anchors, 80)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    rd = B * 10647 * 85 * 4
    print(f"{name}: {us:.1f} us  read {rd/us/1e6:.2f} TB/s")
