#!/usr/bin/env python3
"""nms_kernel + pack_kernel on the boxes / scores the bench's own network produces (seeded weights and images): microseconds per call.
   python tools/time_nms.py [--dtype f32] [--batch 64]"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from yolo_v3_tf2_amd import _lib, runtime  # noqa: E402
from yolo_v3_tf2_amd.core.utils import get_anchors  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=416)
    a = ap.parse_args()
    p = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
    net = runtime.Net(p)
    net.load_weights(synthetic_weights(p, seed=4321))
    net.plan(a.batch, a.image_size, {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16}[a.dtype])
    anchors = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype(np.float32)
    x = torch.from_numpy(np.random.default_rng(1234).random((a.batch, a.image_size, a.image_size, 3), dtype=np.float32)).cuda()
    bb, cls, sc = net.forward_decode(x, anchors)
    torch.cuda.synchronize()
    above = (sc > 0.1).sum(dim=1)
    for name, fn in (("nms_padded", lambda: runtime.nms_padded(bb, sc, 100, 0.5, 0.1)),
                     ("nms_padded + pack", lambda: runtime.pack_detections(bb, cls, sc, *runtime.nms_padded(bb, sc, 100, 0.5, 0.1)))):
        for _ in range(5):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per call ({a.batch} images x {sc.shape[1]} boxes, {a.dtype} network; "
              f"candidates above the score threshold per image: min {int(above.min())} mean {float(above.float().mean()):.0f} max {int(above.max())})")
    sel, nv = runtime.nms_padded(bb, sc, 100, 0.5, 0.1)
    torch.cuda.synchronize()
    print("DIGEST sel", hashlib.sha256(sel.cpu().numpy().tobytes()).hexdigest(), "num_valid", hashlib.sha256(nv.cpu().numpy().tobytes()).hexdigest())


if __name__ == "__main__":
    main()
