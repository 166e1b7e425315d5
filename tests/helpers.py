"""Shared builders for the parity tests."""
import numpy as np

from yolo_v3_tf2_amd.graph import Program, _Builder, _lower


def mini_program(in_channels, convs_chain, heads):
    """input [B,S,S,in_channels] -> chain of convs -> three 'head' convs, all raw outputs (nclasses=0).
    convs_chain / heads: lists of dict(filters,size,stride,bn,act[,shortcut]) in the model-YAML vocabulary."""
    b = _Builder(0, None)
    x = b.new_tensor(in_channels, 1, "input")
    inp = x
    layers = [x]
    for c in convs_chain:
        conf = {"filters": c["filters"], "size": c["size"], "stride": c.get("stride", 1),
                "activation": c.get("act", "leaky")}
        if c.get("bn", True):
            conf["batch_normalize"] = 1
        x = b.conv(x, conf, "body")
        layers.append(x)
        if c.get("shortcut"):
            x = b.shortcut(x, {"from": c["shortcut"], "activation": "linear"}, layers, "body")
            layers.append(x)
    outs = []
    for h in heads:
        conf = {"filters": h["filters"], "size": h["size"], "stride": h.get("stride", 1),
                "activation": h.get("act", "leaky")}
        if h.get("bn", True):
            conf["batch_normalize"] = 1
        outs.append(b.conv(x, conf, "head"))
    p = Program(b.tensors, b.nodes, [], inp, outs, 0)
    p.conv_nodes = [n for n in b.nodes if n.kind == "conv"]
    _lower(p)
    return p


def nms_stress_set(rng, B, N, dup_frac=0.05, score_scale=1.0):
    """centres U(0,1), w,h log-normal(-2,0.8), scores Beta(0.5,4), 5 % exact duplicates of
    boxes and of scores (forces IoU == 1 and sort ties)."""
    cx, cy = rng.random((B, N)), rng.random((B, N))
    w = np.exp(rng.normal(-2.0, 0.8, (B, N)))
    h = np.exp(rng.normal(-2.0, 0.8, (B, N)))
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32)
    scores = (rng.beta(0.5, 4.0, (B, N)) * score_scale).astype(np.float32)
    nd = int(N * dup_frac)
    for b in range(B):
        src = rng.integers(0, N, nd)
        dst = rng.integers(0, N, nd)
        boxes[b, dst] = boxes[b, src]
        src = rng.integers(0, N, nd)
        dst = rng.integers(0, N, nd)
        scores[b, dst] = scores[b, src]
    return boxes, scores


import io  # noqa: E402
import os  # noqa: E402


def jpeg_bytes(rng, h, w):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=95)
    return buf.getvalue()


def make_tfrecord_dataset(dirpath, rng, n=5, sizes=((40, 56), (64, 64), (30, 90))):
    from yolo_v3_tf2_amd.core import load_tfrecords as m
    payloads, truth = [], []
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        k = i % 3          # 0, 1 or 2 boxes
        lo = rng.random((k, 2)).astype(np.float32) * 0.5
        hi = lo + 0.1 + rng.random((k, 2)).astype(np.float32) * 0.4
        names = [[b"circle", b"square", b"no such class"][j % 3] for j in range(k)]
        jpg = jpeg_bytes(rng, h, w)
        payloads.append(m.make_example({"image/encoded": jpg, "image/object/class/text": names,
                                        "image/object/bbox/xmin": lo[:, 0], "image/object/bbox/ymin": lo[:, 1],
                                        "image/object/bbox/xmax": hi[:, 0], "image/object/bbox/ymax": hi[:, 1]}))
        truth.append((jpg, lo, hi, names))
    m.write_records(os.path.join(dirpath, "a_00.tfrec"), payloads[:3])
    m.write_records(os.path.join(dirpath, "b_01.tfrec"), payloads[3:])
    return truth
