"""Thin host wrappers around the C ABI: torch tensors in, torch tensors out.

PyTorch is used for device memory and streams only; every arithmetic step below is a call into
liby3hip.so on the current torch stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import AuxDesc, ConvDesc, TensorDesc, Y3Error, check
from .graph import AuxOp, ConvOp, Program
from .weights import BN_EPS

_AUX_KIND = {"add": _lib.Y3_AUX_ADD, "upsample": _lib.Y3_AUX_UPSAMPLE2X, "concat": _lib.Y3_AUX_CONCAT}


def _fptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dev(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise Y3Error("expected contiguous CUDA(HIP) tensors; the y3 kernels have no CPU fallback")


def tuning_table_path(tag: str, batch: int, image_size: int) -> str:
    """The tile table of a plan: tuning/<mode>_b<batch>_s<size>.json of the package (may not exist: heuristic tiles).
    Y3_TUNING_FILE (tools: A/B of tables) replaces it only for the plan it was made for: the file names its mode
    ("dtype": "f32" | "bf16" | "f32x3" | "f32x2") and its batch / image_size; a plan of another mode or geometry in the
    same process (bench.py's alt measurements re-plan the net) keeps its own packaged table."""
    import json
    import os
    from . import PACKAGE_DIR
    path = os.path.join(PACKAGE_DIR, "tuning", f"{tag}_b{batch}_s{image_size}.json")
    override = os.environ.get("Y3_TUNING_FILE")
    if override and os.path.exists(override):
        with open(override) as f:
            odoc = json.load(f)
        if odoc.get("dtype") == tag and int(odoc.get("batch", -1)) == batch and int(odoc.get("image_size", -1)) == image_size:
            path = override
    return path


class Net:
    """Device-side network = the fused conv program (reference counterpart: the Keras Model returned by
    ParseModel.build_model, core/parse_model.py:279-314)."""

    def __init__(self, program: Program):
        _lib.require_gpu()
        self.program = program
        self.lib = _lib.load()
        p = program
        tens = (TensorDesc * len(p.tensors))(*[TensorDesc(t.channels, t.div) for t in p.tensors])
        kinds, convs, auxs = [], [], []
        self.conv_ops: List[ConvOp] = []
        for o in p.ops:
            if isinstance(o, ConvOp):
                kinds.append(0)
                convs.append(ConvDesc(o.size, o.stride, o.cin, o.cout, int(o.bn), int(o.leaky), o.src0,
                                      int(o.src0_upsample), o.c0, o.src1, o.residual, o.dst, o.in_div, o.out_div))
                self.conv_ops.append(o)
            elif isinstance(o, AuxOp):
                kinds.append(1)
                auxs.append(AuxDesc(_AUX_KIND[o.kind], o.inputs[0], o.inputs[1] if len(o.inputs) > 1 else -1, o.dst))
        if len(p.outputs) != 3:
            raise Y3Error("the HIP path expects exactly three detection heads")
        self._h = C.c_void_p()
        k_arr = (C.c_int32 * len(kinds))(*kinds)
        c_arr = (ConvDesc * max(1, len(convs)))(*convs)
        a_arr = (AuxDesc * max(1, len(auxs)))(*auxs)
        outs = (C.c_int32 * 3)(*p.outputs)
        check(self.lib.y3_net_create(tens, len(p.tensors), k_arr, len(kinds), c_arr, len(convs), a_arr, len(auxs),
                                     p.input_tensor, outs, p.nclasses, C.byref(self._h)), "y3_net_create")
        self.image_size = 0
        self.max_batch = 0
        self.dtype = _lib.Y3_DTYPE_F32
        self.weights_loaded = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.y3_net_destroy(h)
            self._h = None      # (module globals may already be gone at interpreter shutdown)

    # -- weights ----------------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray], eps: float = BN_EPS):
        for slot, o in enumerate(self.conv_ops):
            i = o.conv_index
            w = np.ascontiguousarray(weights[f"conv{i}.w"], np.float32)
            if w.shape != (o.size, o.size, o.cin, o.cout):
                raise Y3Error(f"conv{i}.w has shape {w.shape}, expected {(o.size, o.size, o.cin, o.cout)}")
            get = lambda k: np.ascontiguousarray(weights[f"conv{i}.{k}"], np.float32)
            if o.bn:
                g, b, m, v = get("gamma"), get("beta"), get("mean"), get("var")
                st = self.lib.y3_net_set_conv_weights(self._h, slot, _fptr(w), _fptr(g), _fptr(b), _fptr(m), _fptr(v),
                                                      None, eps)
            else:
                bias = get("bias")
                st = self.lib.y3_net_set_conv_weights(self._h, slot, _fptr(w), None, None, None, None, _fptr(bias), eps)
            check(st, f"y3_net_set_conv_weights(conv{i})")
        self.weights_loaded = True

    # -- planning / execution ---------------------------------------------------------------------
    def keep_activations(self, keep=True):
        check(self.lib.y3_net_keep_activations(self._h, int(keep)), "y3_net_keep_activations")

    def set_lanes(self, lanes: int):
        check(self.lib.y3_net_set_lanes(self._h, int(lanes)), "y3_net_set_lanes")
        self.lanes = int(lanes)

    def set_xcd_mode(self, mode: int):
        """fp32 conv tile placement on the 8 XCDs: 1 = XCD-blocked order chosen per conv (default), 0 = contiguous runs."""
        check(self.lib.y3_net_set_xcd_mode(self._h, int(mode)), "y3_net_set_xcd_mode")

    def set_k_chunk(self, channels: int):
        """K order of the fp32 3x3 convs: channels per chunk (multiple of 32) walked chunk-major, 0 tap-major, -1 default."""
        check(self.lib.y3_net_set_k_chunk(self._h, int(channels)), "y3_net_set_k_chunk")

    def set_stem_fusion(self, on):
        """conv0 + conv1 (+ the 1x1 conv that follows them) as one kernel (default on; applies when the program starts with
        the Darknet-53 stem and the plan is fp32 or bf16 without keep_activations).  2: conv0 + conv1 only."""
        check(self.lib.y3_net_set_stem_fusion(self._h, int(on)), "y3_net_set_stem_fusion")

    def set_early_chunk(self, n_convs: int, chunk_images: int):
        """Before plan(): the first n_convs convs run chunk_images images at a time (their activations then stay in the
        Infinity Cache between producer and consumer); 0, 0 switches it off."""
        check(self.lib.y3_net_set_early_chunk(self._h, int(n_convs), int(chunk_images)), "y3_net_set_early_chunk")

    # A tile forced through these setters survives plan(): the tuning table only fills the convs left on automatic
    # (tile -1 hands the conv back to the table / the library's heuristic).
    def _force(self, kind: str, fn, slot: int, tile: int):
        check(fn(self._h, slot, tile), f"y3_net_set_tile{kind}")
        forced = self.__dict__.setdefault("_forced_tiles", {})
        if tile >= 0:
            forced[(kind, slot)] = tile
        else:
            forced.pop((kind, slot), None)

    def set_tile(self, slot: int, tile: int):
        self._force("", self.lib.y3_net_set_tile, slot, tile)

    def set_tile_x3(self, slot: int, tile: int):
        self._force("_x3", self.lib.y3_net_set_tile_x3, slot, tile)

    def set_tile_x2(self, slot: int, tile: int):
        self._force("_x2", self.lib.y3_net_set_tile_x2, slot, tile)

    def set_tile_bf16(self, slot: int, tile: int):
        self._force("_bf16", self.lib.y3_net_set_tile_bf16, slot, tile)

    def plan(self, max_batch: int, image_size: int, dtype: Optional[int] = None):
        """dtype: _lib.Y3_DTYPE_F32 (default, fp32 MFMA), _lib.Y3_DTYPE_F32X3 (fp32-accurate on the bf16 matrix cores:
        three bf16 planes per value), _lib.Y3_DTYPE_F32X2 (two fp16 planes per value, 2^-22 representation, |x| < 65504)
        or _lib.Y3_DTYPE_BF16 (bf16 activations/weights, fp32 accumulate)."""
        if dtype is None:
            dtype = self.dtype
        check(self.lib.y3_net_plan(self._h, max_batch, image_size, dtype), "y3_net_plan")
        self.max_batch, self.image_size, self.dtype = max_batch, image_size, dtype
        self._apply_tuning()

    @staticmethod
    def conv_signature(o: ConvOp, image_size: int) -> str:
        ho = image_size // o.out_div
        return f"k{o.size}s{o.stride}_c{o.cin}_n{o.cout}_h{ho}_r{int(o.residual >= 0)}_u{int(o.src1 >= 0)}"

    def _apply_tuning(self):
        """Per-conv tile ids measured by tools/tune_tiles.py for this (dtype, batch, image size), if a table exists;
        otherwise the library's heuristic stays in force."""
        import json
        import os
        tag = _lib.DTYPE_TAGS[self.dtype]
        path = tuning_table_path(tag, self.max_batch, self.image_size)
        kind, fn = {_lib.Y3_DTYPE_F32: ("", self.lib.y3_net_set_tile), _lib.Y3_DTYPE_BF16: ("_bf16", self.lib.y3_net_set_tile_bf16),
                    _lib.Y3_DTYPE_F32X3: ("_x3", self.lib.y3_net_set_tile_x3),
                    _lib.Y3_DTYPE_F32X2: ("_x2", self.lib.y3_net_set_tile_x2)}[self.dtype]
        forced = self.__dict__.get("_forced_tiles", {})
        table, lanes = {}, 1
        if os.path.exists(path) and not os.environ.get("Y3_NO_TUNING"):
            with open(path) as f:
                doc = json.load(f)
            table, lanes = doc.get("tiles", {}), int(doc.get("lanes", 1))
        # concurrent sub-batches (tools/lanes_sweep.py): the tail of one sub-batch's kernel overlaps another's bulk
        self.set_lanes(lanes)
        for slot, o in enumerate(self.conv_ops):
            if o.cin == 3 or (kind, slot) in forced:
                continue
            check(fn(self._h, slot, int(table.get(self.conv_signature(o, self.image_size), -1))), f"y3_net_set_tile{kind}")

    def grid_sizes(self, image_size=None):
        s = image_size or self.image_size
        return [s // self.program.tensors[o].div for o in self.program.outputs]

    def forward(self, images: torch.Tensor, out: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        """images [B,S,S,3] fp32 on the GPU -> [grid13, grid26, grid52], each [B,g,g,3,5+nc]."""
        _need_cuda(images)
        cin = self.program.tensors[self.program.input_tensor].channels
        # bf16 plan + an input that feeds an MFMA conv directly (layer tests): the input is bf16 as well
        if self.dtype == _lib.Y3_DTYPE_F32X3 and cin != 3:
            # layer tests: an input feeding an MFMA conv directly is handed over as three bf16 planes [B,S,S,3,C]
            if images.dtype != torch.bfloat16 or images.dim() != 5 or images.shape[3] != 3 or images.shape[4] != cin:
                raise Y3Error(f"images must be bfloat16 [B,S,S,3,{cin}] (split3_planes) in the three-plane mode")
        elif self.dtype == _lib.Y3_DTYPE_F32X2 and cin != 3:
            if images.dtype != torch.float16 or images.dim() != 5 or images.shape[3] != 2 or images.shape[4] != cin:
                raise Y3Error(f"images must be float16 [B,S,S,2,{cin}] (split2_planes) in the two-plane mode")
        else:
            want = torch.bfloat16 if (self.dtype == _lib.Y3_DTYPE_BF16 and cin != 3) else torch.float32
            if images.dtype != want or images.dim() != 4 or images.shape[3] != cin or images.shape[1] != images.shape[2]:
                raise Y3Error(f"images must be {want} [B,S,S,{cin}]")
        B, S = images.shape[0], images.shape[1]
        if S != self.image_size or B > self.max_batch:
            self.plan(max(B, self.max_batch), S)
        nc = self.program.nclasses
        gs = self.grid_sizes()
        if out is None:
            if nc > 0:
                out = [torch.empty((B, g, g, 3, 5 + nc), dtype=torch.float32, device=images.device) for g in gs]
            else:  # raw feature outputs (layer tests)
                out = [torch.empty((B, g, g, self.program.tensors[o].channels), dtype=torch.float32,
                                   device=images.device) for g, o in zip(gs, self.program.outputs)]
        _need_cuda(*out)
        ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in out])
        check(self.lib.y3_net_forward(self._h, _dev(images), B, ptrs, _lib.stream_ptr()), "y3_net_forward")
        return list(out)

    __call__ = forward

    def measure_sclk(self, images: torch.Tensor, out: Sequence[torch.Tensor], forwards: int = 30, conv: int = -1) -> float:
        """Shader clock (MHz) the chip holds under this network's load: `forwards` forwards back to back, in the last one
        a conv launch stamps s_memtime / s_memrealtime (y3_net_measure_sclk: the conv with the most FLOPs; conv >= 0:
        that conv slot, y3_net_measure_sclk_conv).  Raises when no launch of the plan / not that launch carries stamps."""
        _need_cuda(images, *out)
        ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in out])
        mhz = C.c_float()
        if conv >= 0:
            check(self.lib.y3_net_measure_sclk_conv(self._h, _dev(images), images.shape[0], ptrs, int(forwards), int(conv),
                                                    C.byref(mhz), _lib.stream_ptr()), "y3_net_measure_sclk_conv")
        else:
            check(self.lib.y3_net_measure_sclk(self._h, _dev(images), images.shape[0], ptrs, int(forwards), C.byref(mhz),
                                               _lib.stream_ptr()), "y3_net_measure_sclk")
        return float(mhz.value)

    def measure_sclk_all(self, images: torch.Tensor, out: Sequence[torch.Tensor], forwards: int = 30):
        """Per conv slot: (MHz, start_us, end_us) of the stamped workgroup of its launch in the last of `forwards`
        back-to-back forwards (y3_net_measure_sclk_all); MHz 0 where a conv leaves no stamps.  numpy arrays."""
        import numpy as np
        _need_cuda(images, *out)
        ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in out])
        n = len(self.conv_ops)
        mhz = (C.c_float * n)()
        t0 = (C.c_double * n)()
        t1 = (C.c_double * n)()
        check(self.lib.y3_net_measure_sclk_all(self._h, _dev(images), images.shape[0], ptrs, int(forwards), mhz, t0, t1,
                                               _lib.stream_ptr()), "y3_net_measure_sclk_all")
        return np.array(mhz[:], dtype=np.float64), np.array(t0[:]), np.array(t1[:])

    def read_tensor(self, tensor_id: int, batch: int) -> torch.Tensor:
        n = C.c_size_t()
        check(self.lib.y3_net_read_tensor(self._h, tensor_id, batch, None, C.byref(n), None), "y3_net_read_tensor")
        t = self.program.tensors[tensor_id]
        s = self.image_size // t.div
        out = torch.empty((batch, s, s, t.channels), dtype=torch.float32, device="cuda")
        assert out.numel() == n.value
        check(self.lib.y3_net_read_tensor(self._h, tensor_id, batch, _dev(out), C.byref(n), _lib.stream_ptr()),
              "y3_net_read_tensor")
        return out

    def forward_decode(self, images: torch.Tensor, anchors):
        """images -> (bboxes [B,N,4], class_indices [B,N] int64, scores [B,N]): the conv program with the head convs decoding
        their own tiles (y3_net_forward_decode; reference: model(inputs) -> yolo_decode -> argmax / score).  Same bits as
        forward() + yolo_decode_scores()."""
        _need_cuda(images)
        if images.dtype != torch.float32 or images.dim() != 4 or images.shape[3] != 3:
            raise Y3Error("images must be float32 [B,S,S,3]")
        B, S = images.shape[0], images.shape[1]
        if S != self.image_size or B > self.max_batch:
            self.plan(max(B, self.max_batch), S)
        a = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(3, 3, 2))
        n = sum(3 * g * g for g in self.grid_sizes())
        bboxes = torch.empty((B, n, 4), dtype=torch.float32, device=images.device)
        cls = torch.empty((B, n), dtype=torch.int64, device=images.device)
        scores = torch.empty((B, n), dtype=torch.float32, device=images.device)
        check(self.lib.y3_net_forward_decode(self._h, _dev(images), B, _fptr(a), _dev(bboxes), _dev(cls), _dev(scores),
                                             _lib.stream_ptr()), "y3_net_forward_decode")
        return bboxes, cls, scores

    def detect(self, images: torch.Tensor, anchors, max_boxes: int, iou_threshold: float, score_threshold: float):
        """The whole path in one C call (y3_net_detect): -> (packed [B,max_boxes,7] int32 words, num_valid [B] int32);
        `unpack_detections` splits the rows."""
        _need_cuda(images)
        if images.dtype != torch.float32 or images.dim() != 4 or images.shape[3] != 3:
            raise Y3Error("images must be float32 [B,S,S,3]")
        B, S = images.shape[0], images.shape[1]
        if S != self.image_size or B > self.max_batch:
            self.plan(max(B, self.max_batch), S)
        a = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(3, 3, 2))
        packed = torch.empty((B, int(max_boxes), 7), dtype=torch.int32, device=images.device)
        nv = torch.empty((B,), dtype=torch.int32, device=images.device)
        check(self.lib.y3_net_detect(self._h, _dev(images), B, _fptr(a), int(max_boxes), float(iou_threshold),
                                     float(score_threshold), _dev(packed), _dev(nv), _lib.stream_ptr()), "y3_net_detect")
        return packed, nv

    def flops_per_image(self) -> float:
        return float(self.lib.y3_net_flops_per_image(self._h))

    def profile_convs(self, images: torch.Tensor) -> np.ndarray:
        _need_cuda(images)
        ms = np.zeros(len(self.conv_ops), np.float32)
        check(self.lib.y3_net_profile_convs(self._h, _dev(images), images.shape[0], _fptr(ms), len(ms),
                                            _lib.stream_ptr()), "y3_net_profile_convs")
        return ms


# ------------------------------------------------------------------------------------------------
def _grids_args(grids):
    _need_cuda(*grids)
    if len(grids) != 3:
        raise Y3Error("expected three grids")
    for g in grids:
        if g.dtype != torch.float32 or g.dim() != 5 or g.shape[1] != g.shape[2] or g.shape[3] != 3:
            raise Y3Error("each grid must be float32 [B,g,g,3,5+nc]")
    ptrs = (C.c_void_p * 3)(*[g.data_ptr() for g in grids])
    gs = (C.c_int32 * 3)(*[g.shape[1] for g in grids])
    B = grids[0].shape[0]
    N = sum(3 * g.shape[1] * g.shape[2] for g in grids)
    return ptrs, gs, B, N


def _anchors(anchors_table):
    a = np.ascontiguousarray(anchors_table.detach().cpu().numpy() if isinstance(anchors_table, torch.Tensor)
                             else anchors_table, np.float32)
    if a.shape != (3, 3, 2):
        raise Y3Error("anchors_table must be [3,3,2]")
    return a


def yolo_decode(grids, anchors_table, nclasses):
    """-> (bboxes [B,N,4], confidence [B,N,1], class_probs [B,N,nc])"""
    ptrs, gs, B, N = _grids_args(grids)
    a = _anchors(anchors_table)
    dev = grids[0].device
    bboxes = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    conf = torch.empty((B, N, 1), dtype=torch.float32, device=dev)
    probs = torch.empty((B, N, nclasses), dtype=torch.float32, device=dev)
    check(_lib.load().y3_yolo_decode(ptrs, gs, B, nclasses, _fptr(a), _dev(bboxes), _dev(conf), _dev(probs),
                                     _lib.stream_ptr()), "y3_yolo_decode")
    return bboxes, conf, probs


def yolo_decode_scores(grids, anchors_table, nclasses):
    """fused decode + class arg-max/score -> (bboxes [B,N,4], class_indices [B,N] i64, scores [B,N])"""
    ptrs, gs, B, N = _grids_args(grids)
    a = _anchors(anchors_table)
    dev = grids[0].device
    bboxes = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    cls = torch.empty((B, N), dtype=torch.int64, device=dev)
    scores = torch.empty((B, N), dtype=torch.float32, device=dev)
    check(_lib.load().y3_yolo_decode_scores(ptrs, gs, B, nclasses, _fptr(a), _dev(bboxes), _dev(cls), _dev(scores),
                                            _lib.stream_ptr()), "y3_yolo_decode_scores")
    return bboxes, cls, scores


def class_scores(conf, probs):
    _need_cuda(conf, probs)
    B, N, nc = probs.shape
    cls = torch.empty((B, N), dtype=torch.int64, device=probs.device)
    scores = torch.empty((B, N), dtype=torch.float32, device=probs.device)
    check(_lib.load().y3_class_scores(_dev(conf), _dev(probs), B, N, nc, _dev(cls), _dev(scores), _lib.stream_ptr()),
          "y3_class_scores")
    return cls, scores


def split3_planes(x: torch.Tensor) -> torch.Tensor:
    """fp32 [...,C] -> bf16 [...,3,C] with hi + mid + lo == x exactly (the activation format of Y3_DTYPE_F32X3)."""
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo], dim=-2).contiguous()


def split2_planes(x: torch.Tensor) -> torch.Tensor:
    """fp32 [...,C] -> fp16 [...,2,C]: h = fp16(x), l' = fp16((x - h) * 2^11), x == h + l' * 2^-11 up to 2^-22 |x|
    (the activation format of Y3_DTYPE_F32X2)."""
    h = x.to(torch.float16)
    lo = ((x - h.float()) * 2048.0).to(torch.float16)
    return torch.stack([h, lo], dim=-2).contiguous()


def preprocess_image(image: torch.Tensor, batch: torch.Tensor, slot: int, divide_after: bool = False):
    """image [H,W,3|4] uint8 or float32 on the GPU -> batch[slot] ([S,S,3] fp32, values in [0,1] for uint8 input):
    decode_image's uint8->float conversion fused with tf.image.resize's bilinear resampling.  divide_after=True is
    the tfrecords source's order (reference: core/load_tfrecords.py:46-48): resize the 0..255 values, then / 255."""
    _need_cuda(image, batch)
    if image.dim() != 3 or image.dtype not in (torch.uint8, torch.float32) or batch.dtype != torch.float32:
        raise Y3Error("image must be [H,W,C] uint8/float32 and batch float32 [B,S,S,3]")
    if batch.dim() != 4 or batch.shape[1] != batch.shape[2] or batch.shape[3] != 3 or not (0 <= slot < batch.shape[0]):
        raise Y3Error("batch must be [B,S,S,3] and slot inside it")
    H, W, C_ = image.shape
    mode = (2 if divide_after else 1) if image.dtype == torch.uint8 else 0
    if divide_after and mode == 0:
        raise Y3Error("divide_after applies to uint8 images")
    check(_lib.load().y3_preprocess_image(_dev(image), mode, H, W, C_, _dev(batch), slot,
                                          batch.shape[1], _lib.stream_ptr()), "y3_preprocess_image")
    return batch


_ws_cache: Dict[tuple, torch.Tensor] = {}


def nms_padded(bboxes, scores, max_output_size, iou_threshold, score_threshold):
    """-> (selected_indices_padded [B,M] i32, num_valid [B] i32)"""
    _need_cuda(bboxes, scores)
    if bboxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise Y3Error("boxes and scores must be float32")
    B, N = scores.shape
    lib = _lib.load()
    need = lib.y3_nms_workspace_bytes(B, N)
    key = (bboxes.device.index, need)
    ws = _ws_cache.get(key)
    if ws is None:
        _ws_cache.clear()
        ws = _ws_cache[key] = torch.empty(need, dtype=torch.uint8, device=bboxes.device)
    sel = torch.empty((B, int(max_output_size)), dtype=torch.int32, device=bboxes.device)
    nv = torch.empty((B,), dtype=torch.int32, device=bboxes.device)
    check(lib.y3_nms_padded(_dev(bboxes), _dev(scores), B, N, int(max_output_size), float(iou_threshold),
                            float(score_threshold), _dev(sel), _dev(nv), _dev(ws), need, _lib.stream_ptr()),
          "y3_nms_padded")
    return sel, nv


def pack_detections(bboxes, cls, scores, sel, nv):
    """-> [B,M,7] int32 words: box (4 x f32 bits), score (f32 bits), class, index; rows >= num_valid zero"""
    _need_cuda(bboxes, cls, scores, sel, nv)
    B, N = scores.shape
    M = sel.shape[1]
    out = torch.empty((B, M, 7), dtype=torch.int32, device=bboxes.device)
    check(_lib.load().y3_pack_detections(_dev(bboxes), _dev(cls), _dev(scores), _dev(sel), _dev(nv), B, N, M, _dev(out),
                                         _lib.stream_ptr()), "y3_pack_detections")
    return out


def unpack_detections(packed: torch.Tensor):
    """[.., M, 7] int32 words -> (boxes f32 [..,M,4], scores f32 [..,M], classes i32, indices i32)"""
    f = packed[..., :5].contiguous().view(torch.float32)
    return f[..., :4], f[..., 4], packed[..., 5], packed[..., 6]
