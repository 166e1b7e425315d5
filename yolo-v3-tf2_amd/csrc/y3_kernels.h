// Internal launch interface between the C-ABI layer (y3_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "decode_box.h"

namespace y3 {

// One fused conv launch: Conv2D [+BN] [+LeakyReLU(0.1)] [+shortcut add], optional
// (nearest-x2-upsampled src0) (+) src1 channel concat in the A-operand gather.
struct ConvArgs {
    const void *src0;      // [B,H0,W0,C0]  (H0 = H/2 when up0)
    const void *src1;      // [B,H,W,Cin-C0] or nullptr
    const void *wpk;       // packed weights [CoutPad][K], K = taps*Cin, k = tap*Cin + c
    const float *scale;    // [CoutPad] BN scale (1 for bias convs)
    const float *shift;    // [CoutPad] BN shift / bias
    const void *residual;  // [B,Ho,Wo,Cout] or nullptr
    void *dst;             // [B,Ho,Wo,Cout]
    int B, H, W;           // logical input spatial size
    int Ho, Wo;
    int Cin, C0;
    int Cout, CoutPad;
    int ksize, stride, pad;
    int up0;
    int leaky;
    int M;                 // B*Ho*Wo
    int K;                 // ksize*ksize*Cin
    unsigned src0_bytes, src1_bytes, w_bytes, dst_bytes;
    // fp32 tile order: 0 = every XCD takes a contiguous run of tiles (N fastest); gn in {1,2,4,8} = the XCDs form
    // an (8/gn) x gn grid over the (M-tile, N-tile) matrix (see conv_f32.hip)
    int xcd_gn;
    // head convs only (conv_head.hip, the fp32-output epilogue of conv_bf16.hip): dec.boxes != nullptr -> the launch decodes
    // its own output tile (y3_net_forward_decode); dst may then be nullptr (the raw grid is not wanted)
    DecodeHead dec;
    int k_chunk;           // fp32 MFMA kernel, 3x3 convs: > 0 walks K chunk-major, k_chunk input channels at a time (conv_f32.hip); 0: tap-major
    // measurement only (y3_net_measure_sclk): when non-null, thread 0 of the middle workgroup stores {s_memtime, s_memrealtime}
    // at its entry and after its epilogue -> the shader clock held while that workgroup ran.  Null in every product launch.
    unsigned long long *clk_stamps;   // [4]
    int n_cus;             // compute units of the net's device, read once by y3_net_plan: sizes the grids of the persistent kernels (conv_res_*.hip)
    int device;            // the net's device index (the launch goes there: Y3_ENTER_DEVICE); -1 = not known, launchers ask the runtime
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the attribute belongs to the
// device's copy of the code object, so a process driving several GPUs must set it on each.
struct LdsAttrOnce { unsigned long long done = 0; };   // bit d: set on device d (d < 64)
// `dev`: the device the launch goes to, when the caller knows it (ConvArgs::device, set from the net at plan time) -- then a launch
// makes no runtime call here at all once the attribute is set; -1: ask the runtime (stand-alone launches outside a net).
inline hipError_t set_max_lds_once(LdsAttrOnce &st, const void *fn, int bytes, int dev = -1)
{
    hipError_t e = hipSuccess;
    if (dev < 0 && (e = hipGetDevice(&dev)) != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((st.done >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) st.done |= 1ull << dev;
    return e;
}

// tile configurations of the fp32 MFMA kernel (index into the table in conv_f32.hip)
static constexpr int TILE_COUNT = 34;  // 33: the weight-resident 3x3 kernel (conv_res_f32.hip); ids without a selecting plan are retired (conv_f32.hip)
struct TileInfo { int bm, bn, waves, stages; };
TileInfo conv_tile_info(int tile);
bool conv_tile_built(int tile);        // false: retired id

hipError_t launch_conv_f32(const ConvArgs &a, int tile, hipStream_t s);
// weight-resident 3x3 / stride-1 / Cin = 32 conv (conv_res_f32.hip): tile id 33
bool conv_res_f32_fits(const ConvArgs &a);
hipError_t launch_conv_res_f32(const ConvArgs &a, hipStream_t s);
// 1x1 head conv (Cout = 3 * (5 + nc) <= 256) + bias with yolo_decode + arg-max / score fused in (conv_head.hip)
bool conv_head_decode_f32_fits(const ConvArgs &a);
hipError_t launch_conv_head_decode_f32(const ConvArgs &a, hipStream_t s);
// first layer: 3x3 stride-1 conv with Cin=3 (direct, VALU)
hipError_t launch_conv_first_f32(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s);

// fused stem (conv_stem.hip): conv0 (3x3/1, 3->32) + conv1 (3x3/2, 32->64), both BN + optional leaky, one launch
struct StemArgs {
    const float *img;      // [B,S,S,3] fp32
    const float *w0;       // conv0 weights [28][32] fp32, row k = (u*3 + v)*3 + c, row 27 = 0 (fp32 kernel: BN scale folded in)
    const float *scale0;   // [32]  (bf16 kernel only: y = acc*scale + shift like the stand-alone bf16 launches)
    const float *shift0;   // [32]
    const void *w1;        // conv1 packed [64][288], k = tap*32 + c: fp32 with the BN scale folded in / bf16 unscaled
    const float *scale1;   // [64]  (bf16 kernel only)
    const float *shift1;   // [64]
    void *dst;             // [B,S/2,S/2,64] fp32 / bf16
    // optional third layer: the 1x1 conv reading conv1's output (64 -> 32, BN, leaky), computed from the tile
    // while it is still on chip.  w2 = nullptr: absent.
    const void *w2;        // packed [32][64]: fp32 with the BN scale folded in / bf16 unscaled
    const float *scale2;   // [32]  (bf16 kernel only)
    const float *shift2;   // [32]
    void *dst2;            // [B,S/2,S/2,32] fp32 / bf16
    int leaky2;
    unsigned dst2_bytes;
    int B, S;              // S % 32 == 0
    int leaky0, leaky1;
    unsigned img_bytes, dst_bytes;
    int tiles_y, tiles_x, n_tiles;   // filled by the launcher
    // measurement only (y3_net_measure_sclk): when non-null, wave 0 of workgroup 0 stores {s_memtime, s_memrealtime} at
    // kernel entry and exit -> shader clock held during the kernel = d(memtime) / d(memrealtime) x 100 MHz.  Null in
    // every product launch: no stamp instruction executes then.
    unsigned long long *clk_stamps;  // [4]
    int device;            // as ConvArgs::device
    int n_cus;             // as ConvArgs::n_cus (one / two persistent workgroups per CU)
};
hipError_t launch_conv_stem_f32(const StemArgs &a, hipStream_t s);
hipError_t launch_conv_stem_bf16(const StemArgs &a, hipStream_t s);   // conv0 on bf16 MFMA from split (hi + lo) operands (~2^-16 per product), bf16 patch, conv1 on bf16 MFMA

// bf16 path (conv_bf16.hip); TileInfo.stages holds BK for these tiles
static constexpr int BF16_TILE_COUNT = 37;   // 32: the weight-resident 3x3 kernel (conv_res_bf16.hip); 20, 33..36: retired ids
TileInfo conv_bf16_tile_info(int tile);
bool conv_bf16_tile_built(int tile);
hipError_t launch_conv_bf16(const ConvArgs &a, int tile, bool out_f32, hipStream_t s);
// weight-resident 3x3 / stride-1 conv for Cin = 32 / 64 (conv_res_bf16.hip): the early short-K layers of the bf16 path
bool conv_res_bf16_fits(const ConvArgs &a);
hipError_t launch_conv_res_bf16(const ConvArgs &a, hipStream_t s);
hipError_t launch_conv_first_bf16(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s);
hipError_t launch_bf16_to_f32(const void *x, float *y, size_t n, hipStream_t s);

// fp32-accurate path on the bf16 matrix cores, three bf16 planes per value (conv_f32x3.hip); TileInfo.stages holds BK
static constexpr int X3_TILE_COUNT = 34;
TileInfo conv_x3_tile_info(int tile);
hipError_t launch_conv_f32x3(const ConvArgs &a, int tile, bool out_f32, hipStream_t s);
hipError_t launch_conv_first_f32x3(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s);
hipError_t launch_x3_to_f32(const void *x, float *y, size_t npix, int C, hipStream_t s);
// two fp16 planes per value on the fp16 matrix cores (same kernel, same tile ids; a subset is instantiated)
hipError_t launch_conv_f32x2(const ConvArgs &a, int tile, bool out_f32, hipStream_t s);
bool conv_x2_tile_built(int tile);
bool conv_x3_tile_built(int tile);
hipError_t launch_conv_first_f32x2(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s);
hipError_t launch_x2_to_f32(const void *x, float *y, size_t npix, int C, hipStream_t s);

hipError_t launch_add(const float *a, const float *b, float *y, size_t n, hipStream_t s);
hipError_t launch_upsample2x(const float *x, int B, int H, int W, int C, float *y, hipStream_t s);
hipError_t launch_concat(const float *a, int Ca, const float *b, int Cb, size_t npix, float *y, hipStream_t s);

struct DecodeArgs {
    const float *grid[3];
    int g[3];
    int off[3];     // first box index of each scale
    float anchors[3][3][2];
    int B, N, nc;
};
hipError_t launch_decode(const DecodeArgs &a, float *bboxes, float *conf, float *probs, int64_t *cls,
                         float *scores, hipStream_t s);
hipError_t launch_class_scores(const float *conf, const float *probs, size_t n, int nc, int64_t *cls,
                               float *scores, hipStream_t s);

hipError_t launch_resize(const void *src, int is_u8, int H, int W, int pix_stride, float *dst, int S, hipStream_t s);

size_t nms_workspace_bytes(int B, int N);
hipError_t launch_nms(const float *boxes, const float *scores, int B, int N, int M, float T, float S, int32_t *sel,
                      int32_t *num_valid, void *ws, hipStream_t s);
hipError_t launch_pack(const float *boxes, const int64_t *cls, const float *scores, const int32_t *sel,
                       const int32_t *nv, int B, int N, int M, void *packed, hipStream_t s);

}  // namespace y3
