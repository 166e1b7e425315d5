"""The oracle against (a) the committed golden vectors, (b) independent implementations.

PARITY UNPINNED (see oracle/y3_oracle.c): these tests pin the restatement to a second, independently
written implementation (PyTorch-CPU ops for the network, a literal NumPy restatement of TF's tiled NMS,
closed-form NumPy for decode), not to TensorFlow itself."""
import os

import numpy as np
import pytest

from oracle import nms_tiled_ref as T
from oracle import oracle as O
from tests.helpers import mini_program, nms_stress_set

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_nms():
    d = np.load(os.path.join(G, "nms_stress_n3000.npz"))
    M, Tt, S = int(d["params"][0]), float(d["params"][1]), float(d["params"][2])
    sel, nv = O.nms_padded(d["boxes"], d["scores"], M, Tt, S)
    assert np.array_equal(sel, d["sel"]) and np.array_equal(nv, d["num_valid"])


def test_golden_decode():
    d = np.load(os.path.join(G, "decode_g2_4_8.npz"))
    b, c, p = O.yolo_decode([d["g0"], d["g1"], d["g2"]], d["anchors"], 80)
    assert np.array_equal(b, d["bboxes"]) and np.array_equal(c, d["conf"]) and np.array_equal(p, d["probs"])
    _, cls, sc, _, _ = O.yolo_nms((b, c, p), 100, 0.5, 0.1)
    assert np.array_equal(cls, d["cls"]) and np.array_equal(sc, d["scores"])


def test_golden_end_to_end(program, weights, anchors):
    d = np.load(os.path.join(G, "e2e_s64_seed4321.npz"))
    grids = O.forward(program, weights, d["images"])
    assert np.abs(grids[0] - d["grid0"]).max() <= 1e-6
    bb, cc, ss, sel, nv = O.detect(program, weights, d["images"], anchors, 100, 0.5, 0.05)
    assert np.array_equal(nv, d["num_valid"]) and np.array_equal(sel, d["sel"])
    for i in range(2):
        gb, gc, gs = O.gather_valid(bb[i], cc[i], ss[i], sel[i], nv[i])
        assert np.allclose(gb, d[f"boxes{i}"], atol=1e-6) and np.array_equal(gc, d[f"classes{i}"])
        assert np.allclose(gs, d[f"scores{i}"], atol=1e-6)


def test_golden_config1_girl_png(program, weights, anchors):
    """BASELINE config 1 on the reference's own test image (datasets/coco2012/images/girl.png, 812 x 667 RGBA; reference:
    config/detect_config_coco.yaml:11, inference.py:157-163): decode -> bilinear 416^2 -> network -> decode/NMS ->
    gathered detections equal the committed fixture (tools/gen_golden.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = np.load(os.path.join(G, "girl_416_detections.npz"))
    raw = O.decode_image_rgb01(os.path.join(root, "datasets/coco2012/images/girl.png"))
    assert raw.shape == (667, 812, 3) and raw.dtype == np.float32 and 0.0 <= raw.min() and raw.max() <= 1.0
    img = O.resize_bilinear(raw, 416, 416)
    assert np.array_equal(img[::52, ::52], d["input_probe"]) and abs(img.astype(np.float64).sum() - float(d["input_sum"])) < 1e-6
    M, Tt, S = int(d["params"][0]), float(d["params"][1]), float(d["params"][2])
    bb, cc, ss, sl, nv = O.detect(program, weights, img[None], anchors, M, Tt, S)
    assert np.array_equal(nv, d["num_valid"]) and np.array_equal(sl[0, :nv[0]], d["sel"])
    gb, gc, gs = O.gather_valid(bb[0], cc[0], ss[0], sl[0], nv[0])
    assert np.array_equal(gc, d["classes"]) and np.abs(gb - d["boxes"]).max() <= 1e-6 and np.abs(gs - d["scores"]).max() <= 1e-6
    # the host-side input stage of the product (core/utils.py) is a second statement of the same two TF ops
    from yolo_v3_tf2_amd.core.utils import load_image_rgb01, resize_bilinear
    assert np.array_equal(resize_bilinear(load_image_rgb01(os.path.join(root, "datasets/coco2012/images/girl.png")), 416, 416), img)


def test_decode_closed_form(anchors):
    """reference: core/yolo_decode_layer.py:4-36 written out with NumPy broadcasting."""
    rng = np.random.default_rng(3)
    gs = (3, 6, 12)
    grids = [rng.normal(0, 1.5, (2, g, g, 3, 85)).astype(np.float32) for g in gs]
    b, c, p = O.yolo_decode(grids, anchors, 80)
    sig = lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64))))
    outs = []
    for s, g in enumerate(grids):
        n = gs[s]
        col, row = np.meshgrid(np.arange(n), np.arange(n))
        grid = np.stack([col, row], -1)[None, :, :, None, :]
        xy = (sig(g[..., 0:2]) + grid) / n
        wh = np.exp(g[..., 2:4].astype(np.float64)) * anchors[s][None, None, None]
        outs.append(np.concatenate([xy - wh / 2, xy + wh / 2], -1).reshape(2, -1, 4))
    ref = np.concatenate(outs, 1)
    assert b.shape == ref.shape == (2, 3 * (9 + 36 + 144), 4)
    assert np.abs(b - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    refp = np.concatenate([sig(g[..., 5:]).reshape(2, -1, 80) for g in grids], 1)
    assert np.abs(p - refp).max() <= 2e-7 and c.shape == (2, b.shape[1], 1)


def _torch_forward(program, weights, x):
    import torch
    import torch.nn.functional as F
    vals = {program.input_tensor: torch.from_numpy(x).permute(0, 3, 1, 2)}
    for n in program.nodes:
        if n.kind == "conv":
            i = n.conv_index
            w = torch.from_numpy(weights[f"conv{i}.w"]).permute(3, 2, 0, 1).contiguous()
            xin = vals[n.inputs[0]]
            if n.stride == 2:
                y = F.conv2d(F.pad(xin, (1, 0, 1, 0)), w, stride=2)      # ZeroPadding2D(((1,0),(1,0))) + 'valid'
            else:
                y = F.conv2d(xin, w, padding=n.size // 2)                  # 'same'
            if n.bn:
                t = lambda k: torch.from_numpy(weights[f"conv{i}.{k}"])
                y = F.batch_norm(y, t("mean"), t("var"), t("gamma"), t("beta"), False, 0.0, 1e-3)
            else:
                y = y + torch.from_numpy(weights[f"conv{i}.bias"]).view(1, -1, 1, 1)
            if n.leaky:
                y = F.leaky_relu(y, 0.1)
        elif n.kind == "add":
            y = vals[n.inputs[0]] + vals[n.inputs[1]]
        elif n.kind == "upsample":
            y = F.interpolate(vals[n.inputs[0]], scale_factor=2, mode="nearest")
        elif n.kind == "concat":
            y = torch.cat([vals[n.inputs[0]], vals[n.inputs[1]]], 1)
        else:
            y = vals[n.inputs[0]]
        vals[n.output] = y
    return [vals[o].permute(0, 2, 3, 1).numpy() for o in program.outputs]


def test_network_vs_torch_cpu(program, weights):
    """Independent second implementation (different code path, same maths), SURVEY.md section 4."""
    x = np.random.default_rng(1234).random((1, 64, 64, 3), dtype=np.float32)
    ref = _torch_forward(program, weights, x)
    got = O.forward(program, weights, x)
    for r, g in zip(ref, got):
        assert np.abs(g.reshape(r.shape) - r).max() <= 5e-5


def test_network_vs_torch_ref_independent_reader(program, weights):
    """oracle/torch_ref.py: PyTorch-CPU operators driven by the oracle's own YAML interpreter (the CPU baseline that
    bench.py times) against the C restatement."""
    from oracle import torch_ref
    x = np.random.default_rng(99).random((2, 64, 64, 3), dtype=np.float32)
    ref = torch_ref.forward(program.model_config_file, weights, x, 80)
    got = O.forward(program, weights, x)
    assert [r.shape for r in ref] == [g.shape for g in got] == [(2, 2, 2, 3, 85), (2, 4, 4, 3, 85), (2, 8, 8, 3, 85)]
    for r, g in zip(ref, got):
        assert np.abs(g - r).max() <= 5e-5


def test_acc64_bounds_fp32_error(program, weights):
    """fp32 accumulation stays within 1e-4 of fp64 accumulation on the head logits (headroom for the 1e-4 bar)."""
    x = np.random.default_rng(7).random((1, 64, 64, 3), dtype=np.float32)
    a = O.forward(program, weights, x, acc64=False)
    b = O.forward(program, weights, x, acc64=True)
    assert max(np.abs(u - v).max() for u, v in zip(a, b)) <= 1e-4


def test_stride2_padding_is_top_left_only():
    """ZeroPadding2D(((1,0),(1,0))) + 'valid' (reference: core/parse_model.py:34-35): out[0,0] sees only x[0:2,0:2]."""
    x = np.zeros((1, 4, 4, 1), np.float32)
    x[0, 3, 3, 0] = 1.0
    w = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1)
    y = O.conv2d(x, w, stride=2)
    assert y.shape == (1, 2, 2, 1)
    # out[1,1] covers rows/cols 1..3 of x -> tap (2,2)
    assert y[0, 1, 1, 0] == 8.0 and y[0, 0, 0, 0] == 0.0 and y[0, 0, 1, 0] == 0.0


@pytest.mark.parametrize("N,B", [(2000, 2), (700, 3), (513, 1)])
def test_nms_c_equals_tiled_numpy(N, B):
    boxes, scores = nms_stress_set(np.random.default_rng(N), B, N)
    a = O.nms_padded(boxes, scores, 100, 0.5, 0.1)
    b = T.non_max_suppression_padded(boxes, scores, 100, 0.5, 0.1)
    c = T.non_max_suppression_padded(boxes, scores, 100, 0.5, 0.1, converge=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("M,Tt,S", [(10, 0.3, 0.05), (1, 0.5, 0.1), (300, 0.7, 0.0), (100, 0.0, 0.1), (100, 1.0, 0.1),
                                    (100, 0.5, -1.0)])
def test_nms_parameter_corners(M, Tt, S):
    boxes, scores = nms_stress_set(np.random.default_rng(2), 2, 600)
    a = O.nms_padded(boxes, scores, M, Tt, S)
    b = T.non_max_suppression_padded(boxes, scores, M, Tt, S, converge=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_nms_edge_cases():
    # nothing passes the score filter
    boxes, scores = nms_stress_set(np.random.default_rng(0), 1, 100)
    sel, nv = O.nms_padded(boxes, scores * 0, 100, 0.5, 0.1)
    assert nv[0] == 0 and not sel.any()
    # N < max_output_size
    sel, nv = O.nms_padded(boxes[:, :7], np.full((1, 7), 0.9, np.float32), 100, 0.99, 0.1)
    t = T.non_max_suppression_padded(boxes[:, :7], np.full((1, 7), 0.9, np.float32), 100, 0.99, 0.1)
    assert np.array_equal(sel, t[0]) and np.array_equal(nv, t[1])
    # exact duplicates: IoU == 1 >= T, the lower index survives; score ties order by index
    b2 = np.tile(np.array([[0.1, 0.1, 0.5, 0.5]], np.float32), (1, 4, 1))
    s2 = np.array([[0.5, 0.9, 0.9, 0.2]], np.float32)
    sel, nv = O.nms_padded(b2, s2, 100, 0.5, 0.1)
    assert nv[0] == 1 and sel[0, 0] == 1
    # strict score threshold and `>=` IoU threshold
    b3 = np.array([[[0.0, 0.0, 0.4, 0.4], [0.0, 0.0, 0.4, 0.2]]], np.float32)     # IoU exactly 0.5
    sel, nv = O.nms_padded(b3, np.array([[0.9, 0.8]], np.float32), 100, 0.5, 0.8)
    assert nv[0] == 1 and sel[0, 0] == 0                                               # 0.8 > 0.8 is false
    sel, nv = O.nms_padded(b3, np.array([[0.9, 0.8]], np.float32), 100, 0.5, 0.1)
    t = T.non_max_suppression_padded(b3, np.array([[0.9, 0.8]], np.float32), 100, 0.5, 0.1, converge=True)
    assert np.array_equal(sel, t[0]) and np.array_equal(nv, t[1])
    # coordinate canonicalisation follows box [0,0] of the batch
    b4 = np.array([[[0.5, 0.5, 0.1, 0.1], [0.52, 0.5, 0.12, 0.1], [0.9, 0.9, 0.7, 0.7]]], np.float32)
    s4 = np.array([[0.9, 0.8, 0.7]], np.float32)
    a = O.nms_padded(b4, s4, 100, 0.5, 0.1)
    t = T.non_max_suppression_padded(b4, s4, 100, 0.5, 0.1, converge=True)
    assert np.array_equal(a[0], t[0]) and np.array_equal(a[1], t[1]) and a[1][0] == 2


def test_class_scores_first_max_and_int64():
    conf = np.array([[[0.5]], [[0.25]]], np.float32)
    probs = np.zeros((2, 1, 5), np.float32)
    probs[0, 0] = [0.1, 0.7, 0.7, 0.2, 0.7]
    probs[1, 0] = [0.3, 0.3, 0.3, 0.3, 0.3]
    boxes = np.array([[[0.1, 0.1, 0.2, 0.2]], [[0.1, 0.1, 0.2, 0.2]]], np.float32)
    _, cls, sc, _, _ = O.yolo_nms((boxes, conf, probs), 10, 0.5, 0.1)
    assert cls.dtype == np.int64 and cls.tolist() == [[1], [0]]
    assert sc[0, 0] == np.float32(0.5) * np.float32(0.7) and sc[1, 0] == np.float32(0.25) * np.float32(0.3)


def test_mini_program_layers_vs_torch():
    from yolo_v3_tf2_amd.weights import synthetic_weights
    p = mini_program(64, [dict(filters=32, size=1), dict(filters=64, size=3, shortcut=-3)],
                     [dict(filters=128, size=3, stride=2), dict(filters=255, size=1, bn=False, act="linear"),
                      dict(filters=64, size=3)])
    w = synthetic_weights(p, seed=1)
    x = np.random.default_rng(1).standard_normal((2, 10, 10, 64)).astype(np.float32)
    ref = _torch_forward(p, w, x)
    got = O.forward(p, w, x)
    for r, g in zip(ref, got):
        assert np.abs(g.reshape(r.shape) - r).max() <= 2e-5


def test_bf16_free_running_floor(program, weights):
    """Why a free-running bf16 comparison cannot be held to a few ulp: the bf16-emulating oracle against ITSELF with
    fp64 instead of fp32 partial sums (same roundings, same places; only the summation changes, by ~1e-6 relative).
    The first layers agree almost everywhere; by conv12 a third of the stored values differ, from conv20 on more than
    half, and the head logits end 2e-3 ... 1.5e-2 apart (relative L2) -- the level measured between the HIP kernels
    and the oracle (5.4e-3).  A value perturbed by e << ulp flips its rounding with probability e/ulp and then moves a
    whole ulp: rms sqrt(e * ulp), whose fixed point over layers is e = ulp.  Hence the GPU tests bound the free-running
    deviation by this floor and hold every single layer to one ulp on identical inputs (teacher forced)."""
    x = np.random.default_rng(1234).random((1, 96, 96, 3), dtype=np.float32)
    ops = program.conv_ops()
    probe = {ops[i].dst for i in (0, 4, 12, 20, 40)}
    a, ka = O.forward(program, weights, x, bf16=True, keep=probe)
    b, kb = O.forward(program, weights, x, bf16=True, acc64=True, keep=probe)
    frac = {i: float((ka[ops[i].dst] != kb[ops[i].dst]).mean()) for i in (0, 4, 12, 20, 40)}
    assert frac[0] < 1e-3 and frac[4] < 0.05 and frac[12] > 0.1 and frac[20] > 0.3 and frac[40] > 0.3, frac
    rel = [float(np.linalg.norm(u - v) / np.linalg.norm(v)) for u, v in zip(a, b)]
    assert all(2e-3 < r < 1.5e-2 for r in rel), rel
    # every stored value is a bf16 number in both runs (the roundings are where they should be)
    for t in probe:
        assert np.array_equal(O.round_bf16(ka[t]), ka[t]) and np.array_equal(O.round_bf16(kb[t]), kb[t])


def test_two_cpu_fp32_implementations_bound_the_box_bar(program, weights, anchors):
    """Evidence under the box tolerance.  BASELINE.json says "boxes within 1e-4".  The same image
    through TWO independent CPU fp32 implementations of the network (the C restatement and PyTorch-CPU/oneDNN; same
    decode and NMS) already differs by MORE than 1e-4 in raw box coordinates: random-init heads emit unclipped boxes
    tens of image widths wide (w = exp(tw) * anchor, reference core/yolo_decode_layer.py:23) and a 1e-5 summation-
    order difference in tw scales with w.  Inside the unit range -- where a detection lives -- they agree to 1e-4
    absolutely, and the deviation relative to max(1, |coord|) stays under 1e-4 as well.  That is the bar the GPU
    tests use: strict absolute 1e-4 for |coord| <= 1, 1e-4 * |coord| beyond; it is a property of fp32 arithmetic on
    these inputs, not slack granted to the kernel."""
    from oracle import torch_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img = O.decode_image_rgb01(os.path.join(root, "datasets/coco2012/images/girl.png"))
    x = O.resize_bilinear(img, 416, 416)[None]
    a = O.detect(program, weights, x, anchors)
    b = torch_ref.detect(program, weights, x, anchors)
    d = torch_ref.box_deviation(b[0], a[0], b[2], a[2])
    print("cpu_vs_cpu", d)
    assert d["largest_abs_box_coord"] > 10.0                       # the wide boxes are there
    assert d["max_abs_dbox_raw"] > 1e-4                            # two CPU fp32 runs miss an absolute 1e-4 on them
    assert d["max_abs_dbox_coords_within_unit_range"] <= 1e-4      # strict bar where |coord| <= 1
    assert d["max_dbox_over_max1_abs_coord"] <= 1e-4               # scaled bar everywhere
    assert d["max_abs_dscore"] <= 1e-4
    # the two CPU runs pick the same detections; if they did not, the flip would have to be a near-tie
    from oracle import flip_attribution as FA
    flips = FA.attribute(a[0], a[2], a[3], a[4], b[0], b[2], b[3], b[4])
    assert FA.explained(flips, d["max_abs_dscore"], d["max_abs_dbox_coords_within_unit_range"]), flips
