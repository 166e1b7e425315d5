// Weight-resident 3x3 / stride-1 convolution for the early, short-K layers of the bf16 path (BASELINE config 5).
//
// Same fused op as conv_bf16.hip (reference: core/parse_model.py:27-52 conv + BN + LeakyReLU, :155-156 shortcut Add) for the
// convs with Cin = 32 or 64 (K = 288 / 576): 3x3 32 -> 64 @208 and the two 3x3 64 -> 128 @104, each with a shortcut.  In the
// generic implicit-GEMM kernel these launches run at 190-340 TFLOP/s while moving only 1.5 TB/s (profiles/r03_timeline_bf16_*):
// a K loop of 4.5-9 K tiles is shorter than the workgroup's prologue + first fetch + epilogue around it.  Here
//   * the weights of a 64-channel output slice (36.9 / 73.7 KB) stay in LDS for the whole kernel (persistent workgroups, one per
//     CU, 8 waves); a launch with Cout = 128 is two such slices on different workgroups;
//   * a tile is 4 rows x 32 columns of output pixels; its (4+2) x (32+2) input patch comes in by direct-to-LDS loads, DOUBLE
//     BUFFERED: the patch of tile i+1 is requested right after the barrier that opens tile i; out-of-image pixels read as zeros
//     through the buffer bounds check (= the conv's zero padding);
//   * all nine taps x Cin are contracted straight from the patch: the A fragment of tap (u, v) is the lane's patch pixel + an
//     immediate offset (rows) / one of three per-lane addresses (columns: the XOR swizzle key follows the patch COLUMN), the B
//     fragment an immediate offset into the resident weights -- no global load, no barrier, no address arithmetic in the K loop;
//   * per-wave epilogue through a private LDS scratch (no barrier): y = acc * scale + shift, leaky, + shortcut in fp32, ONE
//     rounding to bf16 (where the oracle's bf16 mode rounds), 16-byte stores; the shortcut rows are requested before the K loop.
// One barrier per tile.  k order = tap * Cin + c in groups of 16 (v_mfma_f32_32x32x16_bf16), as in conv_bf16_mfma's 32x32x16 tiles.
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int RTH = 4, RTW = 32;            // output tile (rows x columns): wave (wm, wn) = row wm, channels [32 wn, +32) of the slice
constexpr int RPH = RTH + 2, RPW = RTW + 2; // input patch
constexpr int RNT = 512;                    // 8 waves
constexpr int RSLICE = 64;                  // output channels per workgroup
constexpr int RPATCH_PIX = 208;             // 6 x 34 = 204 patch pixels, rounded up to whole DMA instructions

template <int CIN> struct ResGeom {
    static constexpr int PB = CIN * 2;                  // bytes per patch pixel: 64 (4 chunks of 16 B) or 128 (8 chunks)
    static constexpr int CPP = PB / 16;                 // chunks per pixel
    static constexpr int PIX_PER_DMA = 64 / CPP;        // patch pixels one wave instruction (1 KiB) fills: 16 or 8
    static constexpr int NDMA = (RPH * RPW + PIX_PER_DMA - 1) / PIX_PER_DMA;   // wave instructions per patch: 13 or 26
    static constexpr int DMA_PER_WAVE = (NDMA + 7) / 8;                        // 2 or 4
    static constexpr int KB = 9 * CIN * 2;              // bytes per weight row: 576 or 1152
    static constexpr int W_BYTES = RSLICE * KB;         // 36,864 or 73,728
    static constexpr int PATCH_BYTES = RPATCH_PIX * PB; // 13,312 or 26,624
    static constexpr int SCRATCH_BYTES = 8 * 32 * 32 * 4;
    static constexpr int LDS_BYTES = 2 * PATCH_BYTES + SCRATCH_BYTES;             // 59,392 or 86,016: one workgroup per CU (8 waves at up to 256 registers)
    static constexpr int KS = CIN / 16;                 // MFMA k steps per tap: 2 or 4
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ unsigned pack_bf16_(float lo, float hi)
{
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// swizzle key of a patch column: the 16 lanes of a ds_read_b128 group (consecutive columns of one patch row) then touch 16
// different 16-byte slots of the 256-byte bank row
template <int CIN> __device__ __forceinline__ int key_col(int col) { return CIN == 64 ? (col >> 1) & 7 : (col >> 2) & 3; }

template <int CIN>
__global__ __launch_bounds__(RNT, 2) void conv3x3_res_bf16(const ConvArgs p, int tiles_x, int tiles_y, int n_spatial, int slices)
{
    using G = ResGeom<CIN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const patch0 = lds;                     // two patch buffers
    float *const scratch = reinterpret_cast<float *>(lds + 2 * G::PATCH_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;

    // workgroup w: output-channel slice w % slices (fixed: its weights are resident), spatial tiles w / slices, + gridDim / slices, ...
    const int slice = (int)blockIdx.x % slices;
    const int sstep = (int)gridDim.x / slices;
    int st = (int)blockIdx.x / slices;
    const int n0 = slice * RSLICE;
    if (st >= n_spatial) return;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const unsigned OOB = p.src0_bytes;
    const int H = p.H, W = p.W;

    // ---- the weights of this wave's 32 output channels: resident in REGISTERS for the whole kernel ------------------------------
    // B fragment of (tap t, k step s): lane (n = fr, half fh) holds k = t * CIN + 16 s + 8 fh .. + 7 of weight row n: 9 * KS
    // fragments of 16 bytes = 72 / 144 registers.  (First build: the 64 x K slice in LDS, one ds_read_b128 per MFMA for it -- with
    // 32 x 32 wave tiles that made 2 LDS reads per MFMA and the K phase LDS-bound: 576 KB per tile against 256 B/clk.)
    bf16x8 wfrag[9 * G::KS];
    {
        const unsigned short *wrow = static_cast<const unsigned short *>(p.wpk) + (size_t)(n0 + wn * 32 + fr) * (9 * CIN);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s_ = 0; s_ < G::KS; ++s_)
                wfrag[t * G::KS + s_] = *reinterpret_cast<const bf16x8 *>(wrow + t * CIN + 16 * s_ + 8 * fh);
    }

    // ---- patch DMA: this lane's pixels and chunks (fixed), the tile's origin (per tile) ---------------------------------------
    // wave instruction i = wave + 8 k fills patch pixels [i * PIX_PER_DMA, +PIX_PER_DMA); lane L: pixel + L / CPP, physical chunk L % CPP
    int dpy[G::DMA_PER_WAVE], dpx[G::DMA_PER_WAVE];
    unsigned dconst[G::DMA_PER_WAVE];   // byte offset of the lane's piece relative to the patch origin pixel
#pragma unroll
    for (int k = 0; k < G::DMA_PER_WAVE; ++k) {
        const int P = (wave + 8 * k) * G::PIX_PER_DMA + lane / G::CPP;
        const int py = P / RPW, px = P - py * RPW;
        dpy[k] = P < RPH * RPW ? py : 1 << 20;        // beyond the patch: never inside the image
        dpx[k] = px;
        const int lc = (lane % G::CPP) ^ key_col<CIN>(px);
        dconst[k] = (unsigned)((py * W + px) * G::PB + lc * 16);
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
        const int gy0 = ty * RTH - 1, gx0 = tx * RTW - 1;
        const int origin = ((b * H + gy0) * W + gx0) * G::PB;   // may be negative; only in-image pixels use it
#pragma unroll
        for (int k = 0; k < G::DMA_PER_WAVE; ++k) {
            if (wave + 8 * k < G::NDMA) {
                const bool ok = (unsigned)(gy0 + dpy[k]) < (unsigned)H && (unsigned)(gx0 + dpx[k]) < (unsigned)W;
                const unsigned vo = ok ? (unsigned)origin + dconst[k] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(patch0 + buf * G::PATCH_BYTES + (wave + 8 * k) * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------------
    int a_addr[3][G::KS];   // A: patch pixel (row wm, column fr + v), k step s; tap row u adds u * RPW * PB (immediate)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int s = 0; s < G::KS; ++s)
            a_addr[v][s] = (wm * RPW + fr + v) * G::PB + (((2 * s + fh) ^ key_col<CIN>(fr + v)) << 4);
    const int nw = n0 + wn * 32;                 // first output channel of this wave
    const float sc = p.scale[nw + fr], sh = p.shift[nw + fr];
    float *const S = scratch + wave * (32 * 32);
    // output / shortcut through buffer descriptors: a dead pixel (tile columns beyond the image) gets the out-of-range offset, so
    // every lane ALWAYS issues its two loads and two stores -- the counted wait at the end of a tile relies on that
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.residual ? p.residual : p.dst), 0, p.dst_bytes, 0x00020000);
    const bool has_res = p.residual != nullptr;

    // shortcut rows of a tile's 32 pixels x 32 channels of this wave: 2 pieces of 8 channels per lane.  Requested ONE TILE AHEAD
    // (the shortcut tensor was written two layers ago: its lines come from beyond L2, ~2 us away under load)
    auto request_shortcut = [&](int s_, u32x4 (&rr_)[2], unsigned (&ooff_)[2]) {
        int b, ty, tx;
        tile_coords(s_, b, ty, tx);
        const int oy = ty * RTH + wm;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = lane + it * 64;
            const int r = q >> 2, pc = q & 3;
            const int ox = tx * RTW + r;
            const bool live = oy < p.Ho && ox < p.Wo;
            // byte offset of the lane's 16-byte piece in the output (and shortcut) tensor, or the out-of-range sentinel
            ooff_[it] = live ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + nw + pc * 8) * 2u : p.dst_bytes;
            rr_[it] = __builtin_amdgcn_raw_buffer_load_b128(rsr, (int)(has_res ? ooff_[it] : p.dst_bytes), 0, 0);   // no shortcut: zeros
        }
    };

    u32x4 rr[2], rrn[2];
    unsigned ooff[2], ooffn[2];
    fetch_patch(st, 0);
    request_shortcut(st, rr, ooff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the first patch is in place
    int buf = 0;
    for (; st < n_spatial; st += sstep, buf ^= 1) {
        const bool more = st + sstep < n_spatial;
        if (more) {
            fetch_patch(st + sstep, buf ^ 1);   // the other buffer: every wave left it at the barrier that opened this tile
            request_shortcut(st + sstep, rrn, ooffn);
        }

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const int pboff = buf * G::PATCH_BYTES;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int s = 0; s < G::KS; ++s) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(lds + a_addr[v][s] + pboff + u * (RPW * G::PB));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, wfrag[(u * 3 + v) * G::KS + s], acc, 0, 0, 0);
                }

        // ---- per-wave epilogue: transpose through the private scratch, add the shortcut in fp32, one rounding, 16-byte stores ----
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[e] * sc + sh;
            if (p.leaky) v = fmaxf(v, 0.1f * v);
            S[(4 * fh + (e & 3) + 8 * (e >> 2)) * 32 + fr] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // lanes read what other lanes of this wave wrote (in-order LDS; pins the compiler)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = lane + it * 64;
            const int r = q >> 2, pc = q & 3;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(S + r * 32 + pc * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(S + r * 32 + pc * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (has_res) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[2 * k] = __uint_as_float(rr[it][k] << 16) + v[2 * k];
                    v[2 * k + 1] = __uint_as_float(rr[it][k] & 0xffff0000u) + v[2 * k + 1];
                }
            }
            u32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = pack_bf16_(v[2 * k], v[2 * k + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(out, rsd, (int)ooff[it], 0, 0);   // (dropped by the bounds check for a dead pixel)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // ... and the next tile's scratch writes stay below these reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // next tile: its patch (requested at the top: OLDER in the memory queue than the next tile's two shortcut loads and this
        // tile's two stores) has landed -- those four may still be in flight (vector-memory operations complete in issue order);
        // raw barrier: __syncthreads() would drain them too
        if (more) {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                rr[it] = rrn[it];
                ooff[it] = ooffn[it];
            }
        }
    }
}
}  // namespace

bool conv_res_bf16_fits(const ConvArgs &a)
{
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.src1 && (a.Cin == 32 || a.Cin == 64) && a.Cout % RSLICE == 0 &&
           a.Cout == a.CoutPad && a.H == a.Ho && a.W == a.Wo && a.K == 9 * a.Cin;
}

hipError_t launch_conv_res_bf16(const ConvArgs &a, hipStream_t s)
{
    if (!conv_res_bf16_fits(a)) return hipErrorInvalidValue;
    const int tiles_x = (a.Wo + RTW - 1) / RTW, tiles_y = (a.Ho + RTH - 1) / RTH;
    const int n_spatial = a.B * tiles_y * tiles_x, slices = a.Cout / RSLICE;
    const int cus = a.n_cus > 0 ? a.n_cus : 256;                // read once at plan time (y3_net_plan): no runtime query on the enqueue path
    int per_slice = cus / slices;                               // one persistent workgroup per CU (the resident weights + two patches fill its LDS)
    if (per_slice < 1) per_slice = 1;
    if (per_slice > n_spatial) per_slice = n_spatial;
    const int grid = per_slice * slices;
    if (a.Cin == 32) {
        static LdsAttrOnce attr;
        if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv3x3_res_bf16<32>), ResGeom<32>::LDS_BYTES, a.device); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv3x3_res_bf16<32>, dim3(grid), dim3(RNT), ResGeom<32>::LDS_BYTES, s, a, tiles_x, tiles_y, n_spatial, slices);
    } else {
        static LdsAttrOnce attr;
        if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv3x3_res_bf16<64>), ResGeom<64>::LDS_BYTES, a.device); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv3x3_res_bf16<64>, dim3(grid), dim3(RNT), ResGeom<64>::LDS_BYTES, s, a, tiles_x, tiles_y, n_spatial, slices);
    }
    return hipGetLastError();
}

}  // namespace y3
