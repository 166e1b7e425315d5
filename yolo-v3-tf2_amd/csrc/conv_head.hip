// Detection head with its decode fused in (fp32 path): the 1x1 conv + bias that produces a scale's [B,g,g,3*(5+nc)] grid
// (reference: config/models/yolov3/head*.yaml last layer -> core/parse_model.py:27-52, :209-213 yolo reshape) followed, in the
// same workgroup, by yolo_decode + the class arg-max / score of yolo_nms (reference: core/yolo_decode_layer.py:15-36,
// core/yolo_nms.py:18-24) on the tile while it is still on chip.  y3_net_detect uses it: the 232 MB (64 x 416^2) of head grids
// are then neither written by the conv nor read back by decode_kernel.
//
// GEMM: M = B*g*g pixels, N = 255 channels padded to 256, K = Cin.  One workgroup = 64 pixels x ALL 256 channels (a box's 85
// logits must meet in one place), 8 waves as 2 (M) x 4 (N), wave tile 32 x 64 on v_mfma_f32_32x32x2_f32.  Operand tiles, K
// walk and fragment order are those of conv_f32_mfma's single-stage LDS-DMA tile (conv_f32.hip), so every logit is the SAME
// k-ordered fp32 sum as in a stand-alone launch: the fused route gives the same bits as y3_net_forward + y3_yolo_decode_scores
// (tests/test_gpu_parity.py::test_detect_single_call_equals_composed_pipeline).  After the K loop the tile (+ bias) goes to LDS
// as [64][257] floats; the raw grid is written from there only when the caller wants it (ConvArgs.dst != nullptr).
#include "decode_box.h"
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int HBK = 32, HBM = 64, HBN = 256, HNT = 512, HRP = HNT / 8;   // 64 rows per load pass: A in one pass, B in four
constexpr int HCROW = 257;                                                // floats per row of the output tile in LDS (odd: conflict-free column walks)
constexpr int HSTAGE = (HBM + HBN) * HBK;                                  // floats of the operand stage
constexpr size_t HLDS = sizeof(float) * (HBM * HCROW > 2 * HSTAGE ? HBM * HCROW : 2 * HSTAGE);   // two operand stages (80 KB), the output tile overlays them: 2 workgroups per CU

__global__ __launch_bounds__(HNT, 2) void conv_head_decode_f32(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int m0 = (int)blockIdx.x * HBM;
    const int KT = p.K / HBK;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);

    const int lrow = tid >> 3;
    const int lchunk = (((tid & 7) ^ ((lrow >> 1) & 7)) * 4);   // logical 16-B chunk that lands in physical chunk tid & 7 (see conv_f32.hip)
    const unsigned m = (unsigned)(m0 + lrow);
    const unsigned avoff = m < (unsigned)p.M ? (m * (unsigned)p.Cin + (unsigned)lchunk) * 4u : p.src0_bytes;   // 1x1: row m is pixel m
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = (unsigned)((j * HRP + lrow) * p.K + lchunk) * 4u;

    typedef __attribute__((address_space(3))) void *lds_ptr;
    int kglob = 0;
    auto fetch = [&](int buf) {
        float *sa = smem + buf * HSTAGE + wave * 8 * HBK;   // wave w fills rows [pass * 64 + 8 w, +8)
        float *sb = sa + HBM * HBK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)sa, 16, (int)avoff, kglob * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * HRP * HBK), 16, (int)boff[j], kglob * 4, 0, 0);
        kglob += HBK;
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int a_frag = (wr * 32 + fr) * HBK;
    const int b_frag = HBM * HBK + (wc * 64 + fr) * HBK;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = (((2 * q + fh) ^ ((fr >> 1) & 7)) * 4);

    // double-buffered K loop (the conv_f32_mfma LDS-DMA two-stage form): K tile kt+1 lands under the MFMAs of K tile kt
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) fetch(cur ^ 1);
        const float *st = smem + cur * HSTAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 fa = *reinterpret_cast<const f32x4 *>(st + a_frag + foff[q]);
            f32x4 fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(st + b_frag + j * 32 * HBK + foff[q]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[j][t], acc[j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // K tile kt+1 has landed for every wave; every wave is done reading stage cur (and, at the end, both stages)
    }

    // ---- the tile (+ bias; the BN scale of a BN head would be folded into the weights) into LDS: C[pixel][channel] -----------
    // accumulator element e of lane l: column (n) = l & 31, row (m) = (e & 3) + 8*(e >> 2) + 4*(l >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wc * 64 + j * 32 + fr;
        const float sh = p.shift[n];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[j][e] + sh;
            if (p.leaky) v = fmaxf(v, 0.1f * v);
            smem[(wr * 32 + 4 * fh + (e & 3) + 8 * (e >> 2)) * HCROW + n] = v;
        }
    }
    __syncthreads();
    if (p.dst != nullptr) {   // the raw grid, for callers that want it: rows of Cout floats, contiguous over the tile's pixels
        float *dst = static_cast<float *>(p.dst);
        const int rows = min(HBM, p.M - m0);
        for (int idx = tid; idx < rows * p.Cout; idx += HNT) {
            const int r = idx / p.Cout, c = idx - r * p.Cout;
            dst[(size_t)(m0 + r) * p.Cout + c] = smem[r * HCROW + c];
        }
    }
    if (p.dec.boxes != nullptr)
        decode_rows_from_lds<HNT>(smem, HCROW, HBM, [&](int r) { return m0 + r; }, p.M, p.dec);
}
}  // namespace

bool conv_head_decode_f32_fits(const ConvArgs &a)
{
    return a.ksize == 1 && a.stride == 1 && !a.src1 && !a.residual && a.CoutPad == HBN && a.Cin % HBK == 0 && a.K == a.Cin &&
           a.dec.boxes != nullptr && a.Cout == 3 * (5 + a.dec.nc);
}

hipError_t launch_conv_head_decode_f32(const ConvArgs &a, hipStream_t s)
{
    if (!conv_head_decode_f32_fits(a)) return hipErrorInvalidValue;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_head_decode_f32), (int)HLDS, a.device); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_head_decode_f32, dim3((a.M + HBM - 1) / HBM), dim3(HNT), HLDS, s, a);
    return hipGetLastError();
}

}  // namespace y3
