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
TILES_X2_BUILT = (0, 1, 2, 3, 4, 6, 8, 9, 10, 12, 26, 27, 30, 31, 32, 33)
TILES_X3_BUILT = tuple(range(28)) + (30,)
Y3_AUX_ADD, Y3_AUX_UPSAMPLE2X, Y3_AUX_CONCAT = 0, 1, 2
# (BM, BN, waves, LDS stages) of every tile id of the fp32 MFMA conv kernel (mirror of kTiles in csrc/conv_f32.hip)
TILES = [(128, 128, 4, 2), (256, 64, 4, 2), (256, 32, 4, 2), (128, 64, 4, 2), (64, 128, 4, 2), (64, 64, 4, 2),
         (128, 128, 4, 1), (256, 64, 4, 1), (256, 32, 4, 1), (128, 64, 4, 1), (64, 128, 4, 1), (64, 64, 4, 1),
         (128, 128, 8, 1), (128, 128, 8, 2), (128, 128, 16, 1), (128, 128, 16, 2),
         (256, 128, 16, 1), (128, 64, 8, 1), (256, 64, 8, 1), (128, 64, 8, 2),
         (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0),           # 20..22: retired ids (y3_tile_built answers 0)
         (128, 128, 4, 1), (128, 128, 4, 1),                # 23/24: 128x128 within 3 / 4 waves per SIMD of registers
         (0, 0, 0, 0),                                      # 25: retired id
         (64, 128, 4, 2), (64, 64, 4, 2), (128, 128, 4, 2), (128, 64, 4, 2), (256, 32, 4, 2),  # 26..30: LDS-DMA operand loads
         (64, 128, 4, 1), (64, 64, 4, 1),                   # 31, 32: LDS-DMA, single LDS stage
         (128, 64, 8, 2),                                   # 33: weight-resident 3x3 / stride 1 / Cin 32 (csrc/conv_res_f32.hip)
         (32, 128, 4, 1)]                                   # 34: 32x128 on one row of four waves, LDS-DMA, single stage
RETIRED_TILES = tuple(i for i, t in enumerate(TILES) if t[0] == 0)   # ids y3_tile_built answers 0 for
# three-plane (fp32-accurate on bf16 MFMA) kernel tiles: (BM, BN, waves, BK)
TILES_X3 = [(128, 128, 4, 32), (128, 64, 4, 32), (64, 64, 4, 32), (64, 128, 4, 32), (256, 128, 8, 32), (256, 64, 4, 32),
            (128, 64, 4, 64), (64, 64, 4, 64), (128, 256, 8, 32),
            (128, 128, 4, 32), (256, 128, 8, 32), (128, 256, 8, 32), (128, 128, 8, 32), (128, 64, 4, 32), (64, 128, 4, 32),
            (64, 64, 4, 32),   # 9..11, 13..15: single LDS stage
            (256, 256, 8, 16), (256, 128, 8, 16), (128, 256, 8, 16), (128, 128, 4, 16),   # 16..19: 16-deep K stages
            (256, 128, 8, 32), (128, 256, 8, 32), (128, 128, 4, 32), (128, 128, 8, 32),   # 20..23: interleaved DMA issue
            (256, 128, 8, 32), (128, 256, 8, 32),                                         # 24..25: + pinned issue order
            (256, 128, 16, 32), (128, 256, 16, 32),                                       # 26..27: 16 waves
            (0, 0, 0, 32), (0, 0, 0, 32),                                                 # 28..29: retired ids
            (128, 128, 8, 32), (256, 128, 16, 32), (256, 128, 8, 32), (128, 256, 16, 32)]  # 30..33: three LDS stages
# bf16 kernel tiles: (BM, BN, waves, BK)
TILES_BF16 = [(128, 128, 4, 64), (256, 128, 8, 64), (128, 64, 4, 64), (64, 64, 4, 64), (128, 32, 4, 64),
              (128, 64, 4, 32), (64, 64, 4, 32), (64, 128, 4, 64),
              (128, 128, 4, 64), (256, 128, 8, 64), (128, 64, 4, 64), (64, 64, 4, 64), (64, 128, 4, 64), (128, 256, 8, 64),  # 8..13: LDS-DMA loads
              (256, 256, 8, 64), (256, 128, 4, 64), (128, 256, 4, 64),  # 14..16: LDS-DMA, bigger wave tiles
              (256, 256, 16, 64), (256, 128, 16, 64), (128, 256, 16, 64),  # 17..19: LDS-DMA, 16 waves
              (0, 0, 0, 64),      # 20: retired id (the pipelined tile of round 2)
              (128, 256, 8, 32), (256, 128, 8, 32), (128, 128, 4, 32),  # 21..23: LDS-DMA, BK 32: several workgroups per CU
              (256, 256, 16, 64), (256, 128, 16, 64), (128, 256, 16, 64),  # 24..26: tiles 17..19 on 16x16x32 MFMAs
              (128, 128, 4, 64), (128, 64, 4, 64), (64, 128, 4, 64),       # 27..29: tiles 8, 10, 12 on 16x16x32 MFMAs
              (128, 64, 4, 32), (64, 64, 4, 32),                          # 30, 31: LDS-DMA, BK 32, 64 output channels
              (128, 64, 8, 32),                                           # 32: weight-resident 3x3 / stride 1, Cin 32 / 64 (csrc/conv_res_bf16.hip)
              (0, 0, 0, 64), (0, 0, 0, 64), (0, 0, 0, 64), (0, 0, 0, 64)]  # 33..36: retired in round 5 (tap-row reuse, the four-wave 256x256 tile: no plan selected them)
TILE_NAMES = [f"{bm}x{bn}w{w}s{st}" + ("dma" if 26 <= i <= 32 or i == 34 else "") + ("res" if i == 33 else "") for i, (bm, bn, w, st) in enumerate(TILES)]


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
    "y3_net_set_block_fusion": (_i, [_vp, _i]),
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
