#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/).

PARITY UNPINNED: TensorFlow is absent and the reference holds no golden outputs, so these
vectors pin the *oracle's* behaviour (regressions in oracle/ or in the host logic show up), cross-checked at
generation time against the independent implementations named below.  Inputs are stored, not re-derived.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_v3_tf2_amd  # noqa: E402,F401
from oracle import nms_tiled_ref as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.helpers import nms_stress_set  # noqa: E402
from yolo_v3_tf2_amd.core.utils import get_anchors  # noqa: E402
from yolo_v3_tf2_amd.graph import load_program  # noqa: E402
from yolo_v3_tf2_amd.weights import synthetic_weights  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
anchors = get_anchors(os.path.join(ROOT, "datasets/coco2012/anchors.txt")).astype(np.float32)

# 1. NMS: stress set with duplicates; oracle C == literal tiled NumPy restatement at generation time
boxes, scores = nms_stress_set(np.random.default_rng(99), 2, 3000)
sel, nv = O.nms_padded(boxes, scores, 100, 0.5, 0.1)
sel2, nv2 = T.non_max_suppression_padded(boxes, scores, 100, 0.5, 0.1)
assert np.array_equal(sel, sel2) and np.array_equal(nv, nv2)
np.savez_compressed(os.path.join(G, "nms_stress_n3000.npz"), boxes=boxes, scores=scores, sel=sel, num_valid=nv,
                    params=np.array([100, 0.5, 0.1], np.float32))

# 2. decode: small grids
rng = np.random.default_rng(11)
grids = [rng.normal(0, 1.5, (2, g, g, 3, 85)).astype(np.float32) for g in (2, 4, 8)]
b, c, p = O.yolo_decode(grids, anchors, 80)
cls = np.empty((2, b.shape[1]), np.int64)
sc = np.empty((2, b.shape[1]), np.float32)
O.lib().y3o_scores(c.reshape(-1), p, cls.size, 80, cls.reshape(-1), sc.reshape(-1))
np.savez_compressed(os.path.join(G, "decode_g2_4_8.npz"), g0=grids[0], g1=grids[1], g2=grids[2], anchors=anchors,
                    bboxes=b, conf=c, probs=p, cls=cls, scores=sc)

# 3. end to end at 64x64 (grids 2/4/8, N=252): images + gathered detections only
program = load_program(os.path.join(ROOT, "config/models/yolov3/model.yaml"), 80)
weights = synthetic_weights(program, seed=4321)
x = np.random.default_rng(1234).random((2, 64, 64, 3), dtype=np.float32)
gr = O.forward(program, weights, x)
bb, cc, ss, sl, n = O.detect(program, weights, x, anchors, 100, 0.5, 0.05)
out = {"images": x, "sel": sl, "num_valid": n, "grid0": gr[0]}
for i in range(2):
    gb, gc, gs = O.gather_valid(bb[i], cc[i], ss[i], sl[i], n[i])
    out[f"boxes{i}"], out[f"classes{i}"], out[f"scores{i}"] = gb, gc, gs
np.savez_compressed(os.path.join(G, "e2e_s64_seed4321.npz"), **out)
# 4. BASELINE config 1 (SURVEY.md 8c last row): the reference's own test image datasets/coco2012/images/girl.png
#    (812 x 667 RGBA, config/detect_config_coco.yaml:11) -> decode -> bilinear 416 x 416 -> seeded synthetic weights ->
#    decode/NMS with the config's limits (100, 0.5, 0.1).  Stored: a digest of the network input and the GATHERED
#    detections only (inference.py:21-28), not the 10647-row tensors.
img = O.resize_bilinear(O.decode_image_rgb01(os.path.join(ROOT, "datasets/coco2012/images/girl.png")), 416, 416)
bb, cc, ss, sl, ng = O.detect(program, weights, img[None], anchors, 100, 0.5, 0.1)
gb, gc, gs = O.gather_valid(bb[0], cc[0], ss[0], sl[0], ng[0])
np.savez_compressed(os.path.join(G, "girl_416_detections.npz"), input_sum=np.float64(img.astype(np.float64).sum()),
                    input_probe=img[::52, ::52].copy(), sel=sl[0, :ng[0]], num_valid=ng, boxes=gb, classes=gc, scores=gs,
                    params=np.array([100, 0.5, 0.1], np.float32))
print("girl.png detections:", int(ng[0]))
print("golden written:", sorted(os.listdir(G)), "num_valid e2e:", n)
