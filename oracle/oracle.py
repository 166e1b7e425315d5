"""ctypes front-end of the CPU oracle (oracle/y3_oracle.c) plus the network walker.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  PARITY UNPINNED (TensorFlow absent; see y3_oracle.c header).

The walker executes the *node-level* graph (one op per reference layer: conv, BN,
LeakyReLU, Add, UpSampling2D, Concatenate as separate passes, the way
reference core/parse_model.py:13-160 composes Keras layers), not the fused program
the HIP runtime runs, so that it is an independent statement of the same network.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
BN_EPS = 1e-3

_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liby3oracle.so")
    src = os.path.join(_HERE, "y3_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liby3oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.y3o_conv2d.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, C.c_int, _f, C.c_int]
        L.y3o_bn_fold.argtypes = [_f, _f, _f, _f, C.c_float, C.c_int, _f, _f]
        L.y3o_affine_act.argtypes = [_f, C.c_size_t, C.c_int, _f, _f, C.c_int]
        L.y3o_bias.argtypes = [_f, C.c_size_t, C.c_int, _f]
        L.y3o_add.argtypes = [_f, _f, _f, C.c_size_t]
        L.y3o_upsample2x.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f]
        L.y3o_concat.argtypes = [_f, C.c_int, _f, C.c_int, C.c_size_t, _f]
        L.y3o_decode_scale.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, _f, _f, _f]
        L.y3o_scores.argtypes = [_f, _f, C.c_size_t, C.c_int, _i64, _f]
        L.y3o_nms_padded.argtypes = [_f, _f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _i32, _i32]
        L.y3o_gather_valid.argtypes = [_f, _i64, _f, _i32, C.c_int, _f, _i64, _f]
        for fn in ("y3o_conv2d", "y3o_bn_fold", "y3o_affine_act", "y3o_bias", "y3o_add", "y3o_upsample2x",
                   "y3o_concat", "y3o_decode_scale", "y3o_scores", "y3o_nms_padded", "y3o_gather_valid"):
            getattr(L, fn).restype = None
        _LIB = L
    return _LIB


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------------------- layers
def conv2d(x, w, stride=1, acc64=False):
    """x [B,H,W,Cin], w [k,k,Cin,Cout] -> [B,Ho,Wo,Cout]  (reference: core/parse_model.py:27-43)"""
    x, w = _c(x), _c(w)
    B, H, W, Cin = x.shape
    k, _, cin2, Cout = w.shape
    assert cin2 == Cin
    Ho = H if stride == 1 else (H + 1 - k) // stride + 1
    Wo = W if stride == 1 else (W + 1 - k) // stride + 1
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    lib().y3o_conv2d(x, B, H, W, Cin, w, k, stride, Cout, y, int(acc64))
    return y


def bn_fold(gamma, beta, mean, var, eps=BN_EPS):
    C_ = gamma.shape[0]
    scale, shift = np.empty(C_, np.float32), np.empty(C_, np.float32)
    lib().y3o_bn_fold(_c(gamma), _c(beta), _c(mean), _c(var), eps, C_, scale, shift)
    return scale, shift


def conv_block(x, weights, i, size, stride, bn, leaky, acc64=False):
    """Conv2D [+ BatchNormalization] [+ LeakyReLU(0.1)]  (reference: core/parse_model.py:13-56)"""
    y = conv2d(x, weights[f"conv{i}.w"], stride, acc64)
    cout = y.shape[-1]
    npix = y.size // cout
    if bn:
        scale, shift = bn_fold(weights[f"conv{i}.gamma"], weights[f"conv{i}.beta"], weights[f"conv{i}.mean"],
                               weights[f"conv{i}.var"])
        lib().y3o_affine_act(y, npix, cout, scale, shift, int(leaky))
    else:
        lib().y3o_bias(y, npix, cout, _c(weights[f"conv{i}.bias"]))
        if leaky:
            one, zero = np.ones(cout, np.float32), np.zeros(cout, np.float32)
            lib().y3o_affine_act(y, npix, cout, one, zero, 1)
    return y


def add(a, b):
    y = np.empty_like(a)
    lib().y3o_add(_c(a), _c(b), y, a.size)
    return y


def upsample2x(x):
    B, H, W, Cc = x.shape
    y = np.empty((B, 2 * H, 2 * W, Cc), np.float32)
    lib().y3o_upsample2x(_c(x), B, H, W, Cc, y)
    return y


def concat(a, b):
    B, H, W, Ca = a.shape
    Cb = b.shape[-1]
    y = np.empty((B, H, W, Ca + Cb), np.float32)
    lib().y3o_concat(_c(a), Ca, _c(b), Cb, B * H * W, y)
    return y


# ---------------------------------------------------------------------------- network
def round_bf16(a):
    """fp32 -> bf16 (round to nearest even) -> fp32, NumPy."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32)


def forward(program, weights, images, acc64=False, keep=None, bf16=False):
    """Head grids [[B,g,g,3,5+nc] x 3] of the network `program` describes (reference: core/parse_model.py:279-314).

    A program that records the model.yaml it was read from (`program.model_config_file`, set by the product's
    load_program) is NOT walked: the oracle re-reads that YAML with its own interpreter (oracle/model_reader.py), so
    the product's YAML -> graph code is not part of the checker.  Programs assembled in memory by the layer tests
    (tests/helpers.mini_program, sub-model subsets) carry no file and are walked node by node.
    `keep`: optional set of tensor ids (creation order: 0 = input, then one per layer that creates a tensor) whose
    values are also returned (dict).  bf16=True emulates a bf16 pipeline: see model_reader.bf16_stored."""
    mf = getattr(program, "model_config_file", None)
    if mf:
        return forward_model(mf, weights, images, program.nclasses, acc64=acc64, keep=keep, bf16=bf16)
    return _forward_nodes(program, weights, images, acc64, keep, bf16)


class _Numeric:
    """model_reader backend that computes: one C call per reference layer."""

    def __init__(self, weights, acc64, stored, keep):
        self.weights, self.acc64, self.stored, self.keep = weights, acc64, stored, set(keep or ())
        self.kept = {}
        self.nconv = 0
        self.ntensor = 0          # id of the last created tensor (0 = model input)

    def _out(self, y):
        self.ntensor += 1
        if self.stored is not None and self.ntensor in self.stored:
            y = round_bf16(y)
        if self.ntensor in self.keep:
            self.kept[self.ntensor] = y
        return y

    def conv(self, x, filters, size, stride, bn, leaky, pad, sub):
        i = self.nconv
        self.nconv += 1
        assert pad == 1, "the reference's YAMLs all carry pad: 1 (padding='same' iff stride == 1)"
        w = self.weights[f"conv{i}.w"]
        assert w.shape == (size, size, x.shape[-1], filters), (i, w.shape, (size, size, x.shape[-1], filters))
        if self.stored is not None and x.shape[-1] != 3:
            w = round_bf16(w)      # the MFMA convs hold their weights in bf16; the Cin = 3 first layer stays fp32
        return self._out(conv_block(x, {**{k: v for k, v in self.weights.items() if k.startswith(f"conv{i}.")},
                                        f"conv{i}.w": w}, i, size, stride, bn, leaky, self.acc64))

    def add(self, a, b):
        return self._out(add(a, b))

    def concat(self, a, b):
        return self._out(concat(a, b))

    def upsample(self, x, stride):
        assert stride == 2
        return self._out(upsample2x(x))

    def yolo(self, x, nclasses):
        B, g, g2, ch = x.shape
        return self._out(x.reshape(B, g, g2, 3, ch // 3))   # Reshape((g,g,3,5+nc)), core/parse_model.py:209-210


def forward_model(model_config_file, weights, images, nclasses, acc64=False, keep=None, bf16=False,
                  sub_models=None, output_stage=None):
    """The network of model_config_file, read by the oracle's own interpreter."""
    from . import model_reader as R
    stored = None
    if bf16:
        tr, outs = R.trace(model_config_file, nclasses, sub_models, output_stage, images.shape[-1])
        stored = R.bf16_stored(tr, outs)
    be = _Numeric(weights, acc64, stored, keep)
    x = _c(images)
    if bf16 and x.shape[-1] != 3:
        x = round_bf16(x)
    grids = R.run_model(model_config_file, nclasses, be, x, sub_models, output_stage)
    return (grids, be.kept) if keep else grids


def _forward_nodes(program, weights, images, acc64=False, keep=None, bf16=False):
    """Walk an in-memory node list (duck-typed: .nodes with kind/inputs/output/..., .ops, .outputs)."""
    vals = {program.input_tensor: _c(images)}
    if bf16:
        convs = [o for o in program.ops if hasattr(o, "conv_index")]
        materialised = {o.dst for o in convs}
        # a network output is produced in fp32 straight from the accumulators unless something inside the net reads
        # it again or the launch that writes it is the first-layer kernel / carries a residual (then it is held in
        # bf16 and converted at the end)
        for t in program.outputs:
            read_again = any(t in (o.src0, o.src1, o.residual) for o in convs)
            special = any(o.dst == t and (o.residual >= 0 or o.cin == 3) for o in convs)
            if not (read_again or special):
                materialised.discard(t)
        weights = dict(weights)
        for n in program.conv_nodes:
            if program.tensors[n.inputs[0]].channels != 3:
                weights[f"conv{n.conv_index}.w"] = round_bf16(weights[f"conv{n.conv_index}.w"])
        if program.tensors[program.input_tensor].channels != 3:
            vals[program.input_tensor] = round_bf16(vals[program.input_tensor])
    last_use = {}
    for idx, n in enumerate(program.nodes):
        for i in n.inputs:
            last_use[i] = idx
    keep = set(keep or ())
    outs_needed = set(_yolo_outputs(program)) | set(program.outputs)
    kept = {}
    for idx, n in enumerate(program.nodes):
        if n.kind == "conv":
            y = conv_block(vals[n.inputs[0]], weights, n.conv_index, n.size, n.stride, n.bn, n.leaky, acc64)
        elif n.kind == "add":
            y = add(vals[n.inputs[0]], vals[n.inputs[1]])
        elif n.kind == "upsample":
            y = upsample2x(vals[n.inputs[0]])
        elif n.kind == "concat":
            y = concat(vals[n.inputs[0]], vals[n.inputs[1]])
        elif n.kind == "yolo":
            x = vals[n.inputs[0]]
            B, g, g2, ch = x.shape
            y = x.reshape(B, g, g2, 3, ch // 3)  # Reshape((g,g,3,5+nc)), core/parse_model.py:209-210
        else:
            raise ValueError(n.kind)
        if bf16 and n.output in materialised:
            y = round_bf16(y)
        vals[n.output] = y
        if n.output in keep:
            kept[n.output] = y
        for i in n.inputs:
            if last_use[i] == idx and i not in outs_needed and i != program.input_tensor:
                vals.pop(i, None)
    grids = [vals[o] for o in _yolo_outputs(program)]
    return (grids, kept) if keep else grids


def _yolo_outputs(program):
    # program.outputs were aliased to the head conv outputs by the lowering; find the yolo views
    outs = []
    for o in program.outputs:
        y = [n.output for n in program.nodes if n.kind == "yolo" and n.inputs[0] == o]
        outs.append(y[0] if y else o)
    return outs


# ---------------------------------------------------------------------------- image input (config 1)
def decode_image_rgb01(path):
    """tf.image.decode_image(tf.io.read_file(path), channels=3, dtype=tf.float32) -- reference: inference.py:157.
    PNG/JPEG -> uint8 RGB (an alpha channel is dropped, palette images expanded) -> value / 255 in fp32
    (convert_image_dtype multiplies by the fp32 constant 1/255).  Pillow does the container decode."""
    from PIL import Image
    with Image.open(path) as im:
        u8 = np.asarray(im.convert("RGBA"), np.uint8)[:, :, :3]
    return u8.astype(np.float32) * np.float32(1.0 / 255.0)


def resize_bilinear(img, out_h, out_w):
    """tf.image.resize(img, (out_h, out_w)) -- reference: inference.py:158.  Bilinear, antialias=False, half-pixel
    centres: src = (dst + 0.5) * (in / out) - 0.5; lo = max(floor(src), 0); hi = min(ceil(src), in - 1);
    frac = src - floor(src); the kernel lerps along x on the two source rows, then along y; all in fp32."""
    img = np.ascontiguousarray(img, np.float32)
    H, W, Cc = img.shape
    out = np.empty((out_h, out_w, Cc), np.float32)
    sy, sx = np.float32(H) / np.float32(out_h), np.float32(W) / np.float32(out_w)
    xs = (np.arange(out_w).astype(np.float32) + np.float32(0.5)) * sx - np.float32(0.5)
    x0 = np.floor(xs)
    xl = np.clip(x0, 0, None).astype(np.intp)
    xh = np.minimum(np.ceil(xs), W - 1).astype(np.intp)
    xf = (xs - x0).astype(np.float32)[:, None]
    for y in range(out_h):
        s = (np.float32(y) + np.float32(0.5)) * sy - np.float32(0.5)
        y0 = np.floor(s)
        yl, yh = int(max(y0, 0)), int(min(np.ceil(s), H - 1))
        yf = np.float32(s - y0)
        top = img[yl, xl] + (img[yl, xh] - img[yl, xl]) * xf
        bot = img[yh, xl] + (img[yh, xh] - img[yh, xl]) * xf
        out[y] = top + (bot - top) * yf
    return out


# ---------------------------------------------------------------------------- decode / nms
def yolo_decode(grids, anchors_table, nclasses):
    """reference: core/yolo_decode_layer.py:15-36 -> (bboxes [B,N,4], conf [B,N,1], probs [B,N,nc])"""
    B = grids[0].shape[0]
    N = sum(g.shape[1] * g.shape[2] * 3 for g in grids)
    bboxes = np.empty((B, N, 4), np.float32)
    conf = np.empty((B, N, 1), np.float32)
    probs = np.empty((B, N, nclasses), np.float32)
    off = 0
    anchors_table = _c(anchors_table)
    for s, g in enumerate(grids):
        g = _c(g)
        gh, gw = g.shape[1], g.shape[2]
        lib().y3o_decode_scale(g, B, gh, gw, nclasses, _c(anchors_table[s]), off, N, bboxes, conf, probs)
        off += gh * gw * 3
    return bboxes, conf, probs


def yolo_nms(outputs, yolo_max_boxes, nms_iou_threshold, nms_score_threshold):
    """reference: core/yolo_nms.py:16-34 -> (bboxes, class_indices i64, scores, sel_idx_padded i32, num_valid i32)"""
    bboxes, conf, probs = outputs
    B, N = bboxes.shape[0], bboxes.shape[1]
    nc = probs.shape[-1]
    cls = np.empty((B, N), np.int64)
    scores = np.empty((B, N), np.float32)
    lib().y3o_scores(_c(conf).reshape(-1), _c(probs), B * N, nc, cls.reshape(-1), scores.reshape(-1))
    bboxes = _c(bboxes).reshape(B, -1, 4)
    sel, nv = nms_padded(bboxes, scores, yolo_max_boxes, nms_iou_threshold, nms_score_threshold)
    return bboxes, cls, scores, sel, nv


def nms_padded(boxes, scores, max_output_size, iou_threshold, score_threshold):
    B, N = scores.shape
    sel = np.zeros((B, max_output_size), np.int32)
    nv = np.zeros((B,), np.int32)
    lib().y3o_nms_padded(_c(boxes), _c(scores), B, N, int(max_output_size), float(iou_threshold),
                         float(score_threshold), sel, nv)
    return sel, nv


def gather_valid(bboxes, cls, scores, sel, num_valid):
    """reference: inference.py:21-28 (per image)"""
    n = int(num_valid)
    ob, oc, os_ = np.empty((n, 4), np.float32), np.empty((n,), np.int64), np.empty((n,), np.float32)
    lib().y3o_gather_valid(_c(bboxes), _c(cls, np.int64), _c(scores), _c(sel, np.int32), n, ob, oc, os_)
    return ob, oc, os_


def detect(program, weights, images, anchors_table, yolo_max_boxes=100, nms_iou_threshold=0.5,
           nms_score_threshold=0.1, acc64=False):
    """image batch -> the reference's 5-tuple (reference: inference.py:109-117)."""
    grids = forward(program, weights, images, acc64)
    dec = yolo_decode(grids, anchors_table, program.nclasses)
    return yolo_nms(dec, yolo_max_boxes, nms_iou_threshold, nms_score_threshold)
