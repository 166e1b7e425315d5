"""ctypes binding of liby3hip.so (C ABI declared in include/y3.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails,
`Y3Error` is raised.  Loading the library and resolving its symbols works without a GPU
(used by the CPU test tier); any compute call needs one.
"""
from __future__ import annotations

import ctypes as C
import os

from . import PACKAGE_DIR

# Y3_LIB_PATH: load another build of the same ABI (A/B of kernel variants on one box: tools/ab_libs.py)
LIB_PATH = os.environ.get("Y3_LIB_PATH") or os.path.join(PACKAGE_DIR, "lib", "liby3hip.so")

Y3_OK = 0
Y3_DTYPE_F32, Y3_DTYPE_BF16, Y3_DTYPE_F32X3, Y3_DTYPE_F32X2 = 0, 1, 2, 3
DTYPE_TAGS = {Y3_DTYPE_F32: "f32", Y3_DTYPE_BF16: "bf16", Y3_DTYPE_F32X3: "f32x3", Y3_DTYPE_F32X2: "f32x2"}
TILES_X2_BUILT = (0, 1, 2, 3, 4, 8, 12, 26, 27)
TILES_X3_BUILT = (0, 1, 2, 3, 4, 8, 9, 12, 13, 14)
Y3_AUX_ADD, Y3_AUX_UPSAMPLE2X, Y3_AUX_CONCAT = 0, 1, 2
# (BM, BN, waves, LDS stages) of every tile id of the fp32 MFMA conv kernel (mirror of kTiles in csrc/conv_f32.hip)
# (0, 0, 0, 0): a retired id (y3_tile_built answers 0).  Round 5: only tiles a packaged tuning table or the library's heuristic selects are built.
_Z = (0, 0, 0, 0)
TILES = [_Z, _Z, _Z, _Z, _Z, _Z,
         (128, 128, 4, 1), _Z, (256, 32, 4, 1), (128, 64, 4, 1), (64, 128, 4, 1), (64, 64, 4, 1),   # 6..11: single LDS stage
         (128, 128, 8, 1), _Z, _Z, _Z,
         _Z, (128, 64, 8, 1), _Z, _Z,
         _Z, _Z, _Z,
         (128, 128, 4, 1), _Z,                              # 23: 128x128 within 3 waves per SIMD of registers
         _Z,
         (64, 128, 4, 2), (64, 64, 4, 2), _Z, _Z, _Z,       # 26, 27: LDS-DMA operand loads, two stages
         (64, 128, 4, 1), (64, 64, 4, 1),                   # 31, 32: LDS-DMA, single LDS stage
         (128, 64, 8, 2)]                                   # 33: weight-resident 3x3 / stride 1 / Cin 32 (csrc/conv_res_f32.hip)
RETIRED_TILES = tuple(i for i, t in enumerate(TILES) if t[0] == 0)   # ids y3_tile_built answers 0 for
# three-plane (fp32-accurate on bf16 MFMA) kernel tiles: (BM, BN, waves, BK)
_Z32, _Z64, _Z16 = (0, 0, 0, 32), (0, 0, 0, 64), (0, 0, 0, 16)
TILES_X3 = [(128, 128, 4, 32), (128, 64, 4, 32), (64, 64, 4, 32), (64, 128, 4, 32), (256, 128, 8, 32), _Z32,
            _Z64, _Z64, (128, 256, 8, 32),
            (128, 128, 4, 32), _Z32, _Z32, (128, 128, 8, 32), (128, 64, 4, 32), (64, 128, 4, 32), _Z32,   # 9, 13, 14: single LDS stage
            _Z16, _Z16, _Z16, _Z16,
            _Z32, _Z32, _Z32, _Z32,
            _Z32, _Z32,
            (256, 128, 16, 32), (128, 256, 16, 32),                                       # 26..27: 16 waves (two-plane mode)
            _Z32, _Z32,
            _Z32, _Z32, _Z32, _Z32]
# bf16 kernel tiles: (BM, BN, waves, BK)
TILES_BF16 = [(128, 128, 4, 64), _Z64, _Z64, (64, 64, 4, 64), (128, 32, 4, 64),
              (128, 64, 4, 32), (64, 64, 4, 32), _Z64,
              (128, 128, 4, 64), _Z64, (128, 64, 4, 64), (64, 64, 4, 64), (64, 128, 4, 64), _Z64,  # 8..13: LDS-DMA loads
              _Z64, _Z64, _Z64,
              (256, 256, 16, 64), _Z64, (128, 256, 16, 64),              # 17..19: LDS-DMA, 16 waves
              _Z64,
              _Z32, (256, 128, 8, 32), _Z32,                               # 21..23: LDS-DMA, BK 32, two workgroups per CU
              (256, 256, 16, 64), _Z64, (128, 256, 16, 64), (128, 128, 4, 64), _Z64, (64, 128, 4, 64),   # 24..29: 16x16x32 MFMAs
              _Z32, _Z32,
              (128, 64, 8, 32),                                           # 32: weight-resident 3x3 / stride 1, Cin 32 / 64 (csrc/conv_res_bf16.hip)
              _Z64, _Z64, _Z64, _Z64]                                     # 33..36: retired in round 5
# Tile ids the library's own heuristics can pick (choose_tile* and the head-decode fallback in csrc/y3_api.cpp); with the packaged tuning tables they
# name every tile the library builds (tests/test_abi.py::test_tile_built_reports_the_default_tile_set)
HEURISTIC_TILES = {"f32": (8, 10, 11, 33), "bf16": (4, 5, 6, 8, 10, 11, 12, 17, 19, 24, 26, 27, 29, 32), "f32x3": (0, 1, 2, 3), "f32x2": (0, 1, 2, 3, 4, 8)}
TILE_NAMES = [f"{bm}x{bn}w{w}s{st}" + ("dma" if 26 <= i <= 32 else "") + ("res" if i == 33 else "") for i, (bm, bn, w, st) in enumerate(TILES)]


class Y3Error(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "size", "stride", "cin", "cout", "bn", "leaky", "src0", "src0_upsample", "c0", "src1", "residual", "dst",
        "in_div", "out_div")]


class AuxDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "src0", "src1", "dst")]


class TensorDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("div", C.c_int32)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)

# every symbol include/y3.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "y3_version": (_i, []),
    "y3_last_error": (C.c_char_p, []),
    "y3_device_count": (_i, []),
    "y3_tile_built": (_i, [_i, _i]),
    "y3_net_create": (_i, [C.POINTER(TensorDesc), _i, C.POINTER(C.c_int32), _i, C.POINTER(ConvDesc), _i,
                           C.POINTER(AuxDesc), _i, _i, C.POINTER(C.c_int32), _i, C.POINTER(_vp)]),
    "y3_net_destroy": (None, [_vp]),
    "y3_net_set_conv_weights": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _f]),
    "y3_net_set_tile": (_i, [_vp, _i, _i]),
    "y3_net_set_tile_bf16": (_i, [_vp, _i, _i]),
    "y3_net_set_tile_x3": (_i, [_vp, _i, _i]),
    "y3_net_set_tile_x2": (_i, [_vp, _i, _i]),
    "y3_net_set_early_chunk": (_i, [_vp, _i, _i]),
    "y3_net_keep_activations": (_i, [_vp, _i]),
    "y3_net_set_lanes": (_i, [_vp, _i]),
    "y3_net_set_xcd_mode": (_i, [_vp, _i]),
    "y3_net_set_k_chunk": (_i, [_vp, _i]),
    "y3_net_set_stem_fusion": (_i, [_vp, _i]),
    "y3_net_measure_sclk": (_i, [_vp, _vp, _i, C.POINTER(_vp), _i, _fp, _vp]),
    "y3_net_measure_sclk_conv": (_i, [_vp, _vp, _i, C.POINTER(_vp), _i, _i, _fp, _vp]),
    "y3_net_measure_sclk_all": (_i, [_vp, _vp, _i, C.POINTER(_vp), _i, _fp, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "y3_net_plan": (_i, [_vp, _i, _i, _i]),
    "y3_net_forward": (_i, [_vp, _vp, _i, C.POINTER(_vp), _vp]),
    "y3_net_read_tensor": (_i, [_vp, _i, _i, _vp, C.POINTER(_sz), _vp]),
    "y3_net_flops_per_image": (C.c_double, [_vp]),
    "y3_net_profile_convs": (_i, [_vp, _vp, _i, _fp, _i, _vp]),
    "y3_preprocess_image": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "y3_yolo_decode": (_i, [C.POINTER(_vp), C.POINTER(C.c_int32), _i, _i, _fp, _vp, _vp, _vp, _vp]),
    "y3_yolo_decode_scores": (_i, [C.POINTER(_vp), C.POINTER(C.c_int32), _i, _i, _fp, _vp, _vp, _vp, _vp]),
    "y3_class_scores": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "y3_nms_workspace_bytes": (_sz, [_i, _i]),
    "y3_nms_padded": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "y3_pack_detections": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "y3_crc32c": (C.c_uint32, [_vp, C.c_size_t]),
    "y3_net_forward_decode": (_i, [_vp, _vp, _i, _fp, _vp, _vp, _vp, _vp]),
    "y3_net_detect": (_i, [_vp, _vp, _i, _fp, _i, C.c_float, C.c_float, _vp, _vp, _vp]),
    "y3_comm_get_unique_id": (_i, [_vp]),
    "y3_comm_init_rank": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "y3_comm_destroy": (None, [_vp]),
    "y3_comm_info": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "y3_allgather_results": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
}
Y3_COMM_ID_BYTES = 128
Y3_ERR_INVALID, Y3_ERR_HIP, Y3_ERR_OOM, Y3_ERR_STATE, Y3_ERR_NODEVICE, Y3_ERR_COMM, Y3_ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7   # include/y3.h

_lib = None


def load():
    """Load liby3hip.so and bind every symbol; raises Y3Error when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Y3Error(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                          f"(or yolo-v3-tf2_amd/csrc/build.py); there is no CPU fallback")
        import torch  # noqa: F401  -- first, so that torch's bundled HIP runtime (same soname) is the one we share
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def tile_built(dtype: int, tile: int) -> bool:
    """False for a retired tile id (the ablations / schedules that lost their A/Bs in rounds 1-3 and were removed)."""
    return bool(load().y3_tile_built(int(dtype), int(tile)))


def check(status: int, what: str = ""):
    if status != Y3_OK:
        msg = load().y3_last_error()
        raise Y3Error(f"{what or 'y3 call'} failed ({status}): {msg.decode() if msg else ''}")


def require_gpu():
    import torch
    if not torch.cuda.is_available() or load().y3_device_count() < 1:
        raise Y3Error("no MI355X/HIP device visible: the y3 kernels have no CPU fallback")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
