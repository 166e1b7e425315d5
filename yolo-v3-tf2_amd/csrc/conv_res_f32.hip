// Weight-resident 3x3 / stride-1 / Cin = 32 convolution for the fp32 path: the first residual block's 3x3 conv (32 -> 64 @208 +
// shortcut; reference: config/models/yolov3/backbone.yaml layers 4-5 -> core/parse_model.py:27-52, :155-156).
//
// In the generic implicit-GEMM kernel this launch is the slowest big conv of the fp32 stack (0.74 of the clock-limited peak,
// profiles/r03_sclk_per_layer_f32_b64_s416.txt): K = 288 is nine K tiles, so a workgroup's prologue, first fetch and epilogue are
// as long as its K loop.  Here (the structure of conv_res_bf16.hip):
//   * persistent workgroups, one per CU, 8 waves; the weights of a wave's 32 output channels live in 144 REGISTERS for the whole
//     kernel (v_mfma_f32_32x32x2_f32 takes one float per lane and k pair);
//   * a tile is 8 rows x 16 columns of output pixels (208 = 13 x 16: no ragged tiles); its 10 x 18 x 32-channel input patch comes
//     in by direct-to-LDS loads, double buffered, zero padding = the buffer bounds check;
//   * the nine taps are contracted straight from the patch: per tap one ds_read_b128 per 4 MFMAs, per-lane addresses + immediate
//     offsets, no global load / barrier / vector ALU instruction in the K phase (the fp32 MFMA shares the SIMD with vector work);
//   * per-wave epilogue through a private LDS scratch: + shift, leaky, + shortcut (requested a tile ahead), 16-byte stores.
// SAME k order and lane grouping as conv_f32_mfma (k = tap * 32 + c; lanes 0-31 / 32-63 take c = 8q + t / 8q + 4 + t of MFMA
// (q, t)), so the result is bit-identical to the generic tiles (tests/test_gpu_parity.py::test_f32_weight_resident_conv...).
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int FTH = 8, FTW = 16;              // output tile: wave (wm, wn) = rows 2 wm, 2 wm + 1 (16 pixels each), channels [32 wn, +32) of the slice
constexpr int FPH = FTH + 2, FPW = FTW + 2;   // input patch 10 x 18
constexpr int FNT = 512, FCIN = 32, FSLICE = 64;
constexpr int FPB = FCIN * 4;                 // 128 bytes per patch pixel = 8 chunks of 16 B
constexpr int FNDMA = (FPH * FPW + 7) / 8;    // 23 wave instructions of 8 pixels
constexpr int FDMA_PER_WAVE = (FNDMA + 7) / 8;   // 3
constexpr int FPATCH_BYTES = FNDMA * 1024;    // 23,552
constexpr int FSCRATCH_BYTES = 8 * 32 * 32 * 4;
constexpr int FLDS_BYTES = 2 * FPATCH_BYTES + FSCRATCH_BYTES;   // 79,872

__device__ __forceinline__ int fkey(int col) { return (col >> 1) & 7; }   // swizzle key of a patch column (FPW is even: a pixel's bank-row half is its column parity)

__global__ __launch_bounds__(FNT, 2) void conv3x3_res_f32(const ConvArgs p, int tiles_x, int tiles_y, int n_spatial, int slices)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const patch0 = lds;
    float *const scratch = reinterpret_cast<float *>(lds + 2 * FPATCH_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    // measurement launches only (y3_net_measure_sclk*, same protocol as conv_f32_mfma): thread 0 of the middle workgroup stamps
    // {s_memtime, s_memrealtime} at its entry and after its last tile, workgroup 0 the kernel's begin.  Null in product launches.
    if (p.clk_stamps != nullptr && blockIdx.x == (gridDim.x >> 1) && tid == 0) {
        p.clk_stamps[0] = __builtin_amdgcn_s_memtime();
        p.clk_stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
    if (p.clk_stamps != nullptr && blockIdx.x == 0 && tid == 0) p.clk_stamps[4] = __builtin_amdgcn_s_memrealtime();

    const int slice = (int)blockIdx.x % slices;
    const int sstep = (int)gridDim.x / slices;
    int st = (int)blockIdx.x / slices;
    const int n0 = slice * FSLICE;
    if (st >= n_spatial) return;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.residual ? p.residual : p.dst), 0, p.dst_bytes, 0x00020000);
    const bool has_res = p.residual != nullptr;
    const unsigned OOB = p.src0_bytes;
    const int H = p.H, W = p.W;

    // ---- weights of this wave's 32 output channels: registers.  wq[t][q] = W[n][k = 32 t + 8 q + 4 fh .. + 3] (BN scale folded in) ----
    f32x4 wq[9][4];
    {
        const float *wrow = static_cast<const float *>(p.wpk) + (size_t)(n0 + wn * 32 + fr) * (9 * FCIN);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[t][q] = *reinterpret_cast<const f32x4 *>(wrow + t * FCIN + 8 * q + 4 * fh);
    }

    // ---- patch DMA: wave instruction i = wave + 8 k fills patch pixels [8 i, 8 i + 8); lane L: pixel + L / 8, physical chunk L % 8 ----
    int dpy[FDMA_PER_WAVE], dpx[FDMA_PER_WAVE];
    unsigned dconst[FDMA_PER_WAVE];
#pragma unroll
    for (int k = 0; k < FDMA_PER_WAVE; ++k) {
        const int P = (wave + 8 * k) * 8 + (lane >> 3);
        const int py = P / FPW, px = P - py * FPW;
        dpy[k] = P < FPH * FPW ? py : 1 << 20;
        dpx[k] = px;
        dconst[k] = (unsigned)((py * W + px) * FPB + (((lane & 7) ^ fkey(px)) << 4));
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto tile_coords = [&](int s_, int &b, int &ty, int &tx) {
        const int per_img = tiles_y * tiles_x;
        b = s_ / per_img;
        const int r = s_ - b * per_img;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
    };
    auto fetch_patch = [&](int s_, int buf) {
        int b, ty, tx;
        tile_coords(s_, b, ty, tx);
        const int gy0 = ty * FTH - 1, gx0 = tx * FTW - 1;
        const int origin = ((b * H + gy0) * W + gx0) * FPB;
#pragma unroll
        for (int k = 0; k < FDMA_PER_WAVE; ++k) {
            if (wave + 8 * k < FNDMA) {
                const bool ok = (unsigned)(gy0 + dpy[k]) < (unsigned)H && (unsigned)(gx0 + dpx[k]) < (unsigned)W;
                const unsigned vo = ok ? (unsigned)origin + dconst[k] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(patch0 + buf * FPATCH_BYTES + (wave + 8 * k) * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
    };

    // A fragment of (tap (u, v), q): the lane's patch pixel (row 2 wm + (fr >> 4) + u, column (fr & 15) + v), chunk 2 q + fh
    const int prow = 2 * wm + (fr >> 4), pcol = fr & 15;
    int a_addr[3][4];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) a_addr[v][q] = (prow * FPW + pcol + v) * FPB + (((2 * q + fh) ^ fkey(pcol + v)) << 4);

    const int nw = n0 + wn * 32;
    const float sh = p.shift[nw + fr];
    float *const S = scratch + wave * (32 * 32);

    // shortcut / output pieces of this wave's 32 pixels x 32 channels: 8 pieces of 4 channels per pixel -> 4 per lane
    auto request_shortcut = [&](int s_, u32x4 (&rr_)[4], unsigned (&ooff_)[4]) {
        int b, ty, tx;
        tile_coords(s_, b, ty, tx);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = lane + it * 64;
            const int r = q >> 3, pc = q & 7;                       // pixel of the wave's 32, piece of 4 channels
            const int oy = ty * FTH + 2 * wm + (r >> 4), ox = tx * FTW + (r & 15);
            const bool live = oy < p.Ho && ox < p.Wo;
            ooff_[it] = live ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + nw + pc * 4) * 4u : p.dst_bytes;
            rr_[it] = __builtin_amdgcn_raw_buffer_load_b128(rsr, (int)(has_res ? ooff_[it] : p.dst_bytes), 0, 0);
        }
    };

    u32x4 rr[4], rrn[4];
    unsigned ooff[4], ooffn[4];
    fetch_patch(st, 0);
    request_shortcut(st, rr, ooff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (; st < n_spatial; st += sstep, buf ^= 1) {
        const bool more = st + sstep < n_spatial;
        if (more) {
            fetch_patch(st + sstep, buf ^ 1);
            request_shortcut(st + sstep, rrn, ooffn);
        }

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const unsigned char *pb = patch0 + buf * FPATCH_BYTES;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 fa = *reinterpret_cast<const f32x4 *>(pb + a_addr[v][q] + u * (FPW * FPB));
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], wq[u * 3 + v][q][t], acc, 0, 0, 0);
                }

        // ---- per-wave epilogue ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[e] + sh;
            if (p.leaky) v = fmaxf(v, 0.1f * v);
            S[(4 * fh + (e & 3) + 8 * (e >> 2)) * 32 + fr] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = lane + it * 64;
            const int r = q >> 3, pc = q & 7;
            f32x4 v = *reinterpret_cast<const f32x4 *>(S + r * 32 + pc * 4);
            if (has_res) v = __builtin_bit_cast(f32x4, rr[it]) + v;     // shortcut + conv, like conv_f32_mfma (r + v)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsd, (int)ooff[it], 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the next patch (requested at the top) is older in the memory queue than the four shortcut loads of the next tile and this
        // tile's four stores: those eight may still be in flight
        if (more) {
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                rr[it] = rrn[it];
                ooff[it] = ooffn[it];
            }
        }
    }
    if (p.clk_stamps != nullptr && blockIdx.x == (gridDim.x >> 1) && tid == 0) {
        p.clk_stamps[2] = __builtin_amdgcn_s_memtime();
        p.clk_stamps[3] = __builtin_amdgcn_s_memrealtime();
    }
}
}  // namespace

bool conv_res_f32_fits(const ConvArgs &a)
{
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.src1 && a.Cin == FCIN && a.Cout % FSLICE == 0 && a.Cout == a.CoutPad &&
           a.H == a.Ho && a.W == a.Wo && a.K == 9 * FCIN && a.k_chunk == 0;
}

hipError_t launch_conv_res_f32(const ConvArgs &a, hipStream_t s)
{
    if (!conv_res_f32_fits(a)) return hipErrorInvalidValue;
    const int tiles_x = (a.Wo + FTW - 1) / FTW, tiles_y = (a.Ho + FTH - 1) / FTH;
    const int n_spatial = a.B * tiles_y * tiles_x, slices = a.Cout / FSLICE;
    const int cus = a.n_cus > 0 ? a.n_cus : 256;   // read once at plan time (y3_net_plan): no runtime query on the enqueue path
    int per_slice = cus / slices;
    if (per_slice < 1) per_slice = 1;
    if (per_slice > n_spatial) per_slice = n_spatial;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv3x3_res_f32), FLDS_BYTES, a.device); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv3x3_res_f32, dim3(per_slice * slices), dim3(FNT), FLDS_BYTES, s, a, tiles_x, tiles_y, n_spatial, slices);
    return hipGetLastError();
}

}  // namespace y3
