// A whole residual block of the early, HBM-bound stages of the bf16 path in ONE launch (BASELINE config 5; round 5):
//     mid = LeakyReLU(BN(conv1x1(x)))  (CX -> CX / 2)        out = LeakyReLU(BN(conv3x3(mid))) + x  (CX / 2 -> CX)
// (reference: config/models/yolov3/backbone.yaml residual blocks -> core/parse_model.py:27-52 conv + BN + LeakyReLU, :143-160 shortcut Add)
// for CX = 64 (the block at 208 x 208) and CX = 128 (the two blocks at 104 x 104).  These launches move bytes, not FLOPs: alone on the chip the
// weight-resident 3x3 (conv_res_bf16.hip) already runs at the 5 TB/s HBM delivers (profiles/r05_ab_bf16_res_triple_buffer.txt), so only FEWER
// bytes help -- and the two-launch form moves the block's input three times (the 1x1 reads it, the 3x3 adds it as shortcut) and the middle
// tensor twice (written, then read with its halo).  Here a persistent workgroup (one per CU, 8 waves) walks 4 x 32-pixel output tiles:
//   * the (4 + 2) x (32 + 2) patch of x arrives by direct-to-LDS loads, double buffered (tile i + 1's patch requested when tile i opens);
//     out-of-image pixels read as zeros through the buffer bounds check;
//   * phase 1: the 1x1 conv for all 204 patch pixels, 7 x (CX / 64) blocks of 32 pixels x 32 channels dealt to the 8 waves, its weights in
//     registers; issued with the MFMA operands exchanged (D' = W1 x X^T: a lane then holds 16 channels of ONE pixel -- its validity is two
//     compares per tile, its result leaves as four 8-byte LDS stores; exchanging the operands changes no bit, profiles/r05_ab_bf16_swap_epilogue.txt);
//     y = acc * scale + shift, leaky, ONE rounding to bf16 -- exactly where the stand-alone launch stores -- into the middle patch in LDS, with ZEROS
//     for patch pixels outside the image (the 3x3's zero padding pads the middle tensor, not x: LeakyReLU(BN(0)) is not 0);
//   * phase 2: conv_res_bf16.hip's K loop on the middle patch (weights of the wave's 32 output channels in registers, taps = immediate offsets);
//   * epilogue: transposition through a private LDS scratch (two 16-row passes), the shortcut read from the CENTRE of the x patch in LDS,
//     one rounding, 16-byte stores.
// k order of both convs = the stand-alone launches' (ascending k in steps of 16 on one accumulator chain, v_mfma_f32_32x32x16_bf16): the block's
// output is bit-identical to the two-launch form (tests/test_gpu_parity.py::test_bf16_fused_block_bit_identical_to_two_launches).  The middle tensor
// is never written: 0.18 / 0.35 GB per 128 images and block less to write, ~1.2 x that less to read, and the shortcut's 0.35 / 0.71 GB not read twice.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int KTH = 4, KTW = 32;            // output tile (rows x columns): wave (wm, wn) = row wm, channels [32 wn, +32) of the slice
constexpr int KPH = KTH + 2, KPW = KTW + 2; // patch
constexpr int KNT = 512;                    // 8 waves
constexpr int KSLICE = 64;                  // output channels per workgroup
constexpr int KPIX = 208;                   // patch pixels allocated: 204 used.  The phase-1 GEMM has 7 blocks of 32 rows = 224: its rows >= 208 read whatever
                                            // follows the buffer (finite or not: every output row depends on its own input row only) and are not written

template <int CX> struct BlockGeom {
    static constexpr int CM = CX / 2;
    static constexpr int XB = CX * 2;                    // bytes per x pixel: 128 / 256
    static constexpr int MB = CM * 2;                    // bytes per middle pixel: 64 / 128
    static constexpr int XCPP = XB / 16;                 // 16-byte chunks per x pixel: 8 / 16
    static constexpr int XPIX_PER_DMA = 64 / XCPP;       // patch pixels one wave instruction (1 KiB) fills: 8 / 4
    static constexpr int XNDMA = (KPH * KPW + XPIX_PER_DMA - 1) / XPIX_PER_DMA;   // 26 / 51
    static constexpr int XDMA_PER_WAVE = (XNDMA + 7) / 8;                         // 4 / 7
    static constexpr int X_BYTES = KPIX * XB;            // 26,624 / 53,248
    static constexpr int MID_BYTES = KPIX * MB;          // 13,312 / 26,624
    static constexpr int SCRATCH_BYTES = 8 * 16 * 32 * 4;                         // 16 KB: the epilogue goes in two 16-row passes
    static constexpr int MID_OFF = 2 * X_BYTES;
    static constexpr int SCRATCH_OFF = MID_OFF + MID_BYTES;
    static constexpr int CONST_OFF = SCRATCH_OFF + SCRATCH_BYTES;                 // scale / shift of the 1x1 conv: 2 x CM floats
    static constexpr int W1_OFF = CONST_OFF + 2 * CM * 4;                         // second K half of the 1x1's weights: [CM rows][CX bytes]
    static constexpr int W1_BYTES = CM * CX;                                      // 2,048 / 8,192
    static constexpr int LDS_BYTES = W1_OFF + W1_BYTES;                           // 78,592 / 158,208
    static constexpr int KS1 = CX / 16;                  // phase-1 k steps: 4 / 8
    static constexpr int KS2 = CM / 16;                  // phase-2 k steps per tap: 2 / 4
    static constexpr int NCB = CM / 32;                  // phase-1 channel blocks: 1 / 2
    static constexpr int NTASK = 7 * NCB;                // phase-1 blocks of 32 pixels x 32 channels: 7 / 14
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ unsigned pack_bf16_k(float lo, float hi)
{
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// x patch: pixel P (linear, row-major over the 6 x 34 patch) at P * XB, 16-byte chunk c at c ^ key_x(P): the 16 lanes of a ds_read_b128 group read the
// same chunk of 16 consecutive-ish pixels (phase 1: rows of the GEMM = pixels in linear order)
template <int CX> __device__ __forceinline__ int key_x(int P) { return CX == 128 ? (P & 15) : ((P >> 1) & 7); }
// middle patch: keyed by the patch COLUMN (conv_res_bf16.hip): a tap row is then an immediate offset
template <int CM> __device__ __forceinline__ int key_m(int col) { return CM == 64 ? (col >> 1) & 7 : (col >> 2) & 3; }

template <int CX>
__global__ __launch_bounds__(KNT, 2) void conv_block_bf16(const ConvArgs p, int tiles_x, int tiles_y, int n_spatial, int slices)
{
    using G = BlockGeom<CX>;
    constexpr int CM = G::CM;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *const scratch = reinterpret_cast<float *>(lds + G::SCRATCH_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;

    // workgroup w: output-channel slice w % slices (its 3x3 weights are resident), spatial tiles w / slices, + gridDim / slices, ...
    const int slice = (int)blockIdx.x % slices;
    const int sstep = (int)gridDim.x / slices;
    int st = (int)blockIdx.x / slices;
    const int n0 = slice * KSLICE;
    if (st >= n_spatial) return;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.residual), 0, p.dst_bytes, 0x00020000);   // x: the block's input = its shortcut
    const unsigned OOB = p.dst_bytes;
    const int H = p.H, W = p.W;

    // ---- weights, resident in registers for the whole kernel ------------------------------------------------------------------------------
    // 3x3: B fragment of (tap t, k step s): lane (n = fr, half fh) holds k = t * CM + 16 s + 8 fh .. + 7 of weight row n0 + 32 wn + fr
    bf16x8 wfrag[9 * G::KS2];
    {
        const unsigned short *wrow = static_cast<const unsigned short *>(p.wpk) + (size_t)(n0 + wn * 32 + fr) * (9 * CM);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s_ = 0; s_ < G::KS2; ++s_) wfrag[t * G::KS2 + s_] = *reinterpret_cast<const bf16x8 *>(wrow + t * CM + 16 * s_ + 8 * fh);
    }
    // 1x1: phase-1 task t = wave + 8 j covers patch pixels [32 (t / NCB), +32) x middle channels [32 (t % NCB), +32); with NCB in {1, 2} both tasks
    // of a wave have the same channel block cb = wave % NCB, so one set of fragments serves them: row 32 cb + fr, k = 16 s + 8 fh .. + 7
    // The 3x3's 144 weight registers (CX = 128) leave room for HALF of them: k steps [0, KS1 / 2) in registers, the rest in LDS ([CM rows][CX bytes]:
    // 16-byte chunk c of row n at c ^ (n & (chunks per row - 1)), so that the 32 rows a fragment read touches fall into different banks)
    const int cb = wave % G::NCB;
    constexpr int KH = G::KS1 / 2;
    bf16x8 w1frag[KH];
    {
        const unsigned short *wrow = static_cast<const unsigned short *>(p.blk.w) + (size_t)(cb * 32 + fr) * CX;
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) w1frag[s_] = *reinterpret_cast<const bf16x8 *>(wrow + 16 * s_ + 8 * fh);
        constexpr int CPR = CX / 16;                      // 16-byte chunks per half row
        for (int g = tid; g < CM * CPR; g += KNT) {
            const int n = g / CPR, c = g - n * CPR;
            *reinterpret_cast<u32x4 *>(lds + G::W1_OFF + n * CX + ((c ^ (n & (CPR - 1))) << 4)) =
                *reinterpret_cast<const u32x4 *>(static_cast<const unsigned short *>(p.blk.w) + (size_t)n * CX + CX / 2 + c * 8);
        }
    }
    const int w1l = G::W1_OFF + (cb * 32 + fr) * CX + ((fh ^ (fr & (CX / 16 - 1))) << 4);   // k step KH (chunk fh); step KH + s: XOR (2 s) << 4
    // exchanged operands: this lane's 16 phase-1 results are middle channels 32 cb + 8 (e >> 2) + 4 fh + (e & 3) of ONE patch pixel; their 2 x 16
    // constants are read from LDS where they are used (the 3x3's 144 weight registers leave no room to hold them)
    float *const s_c1 = reinterpret_cast<float *>(lds + G::CONST_OFF);
    if (tid < CM) {
        s_c1[tid] = p.blk.scale[tid];
        s_c1[CM + tid] = p.blk.shift[tid];
    }
    const float *const c1 = s_c1 + cb * 32 + 4 * fh;       // + 8 q: scale; + CM + 8 q: shift

    // ---- x patch DMA: this lane's pixels and chunks (fixed), the tile's origin (per tile) -------------------------------------------------------
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
        const int gy0 = ty * KTH - 1, gx0 = tx * KTW - 1;
        const int origin = ((b * H + gy0) * W + gx0) * G::XB;   // may be negative; only in-image pixels use it
        // The lane's patch pixel / chunk of every instruction is recomputed per tile from the lane index (a dozen vector instructions each)
        // instead of sitting in a register for the whole kernel: 176 of the 256 registers hold weights.  The empty asm makes the lane index
        // opaque here, or the compiler hoists the arithmetic out of the tile loop again -- and spills it.
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < G::XDMA_PER_WAVE; ++k) {
            if (wave + 8 * k < G::XNDMA) {
                const int P = (wave + 8 * k) * G::XPIX_PER_DMA + ln / G::XCPP;
                const int py = P / KPW, px = P - py * KPW;
                const int lc = (ln % G::XCPP) ^ key_x<CX>(P);
                const bool ok = P < KPH * KPW && (unsigned)(gy0 + py) < (unsigned)H && (unsigned)(gx0 + px) < (unsigned)W;
                const unsigned vo = ok ? (unsigned)(origin + (py * W + px) * G::XB + lc * 16) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(lds + buf * G::X_BYTES + (wave + 8 * k) * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
    };

    // ---- phase-1 addresses (per lane, fixed): task j = 0, 1 -> patch pixel P = 32 pb + fr ------------------------------------------------------
    int t_xa[2], t_mw[2];
    unsigned t_yx[2];          // patch row (bits 8..) and column (bits 0..7) of the lane's pixel; row 255: no such pixel / task
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = wave + 8 * j;
        const int pb = t / G::NCB;
        const int P = 32 * pb + fr;
        const int py = P / KPW, px = P - py * KPW;
        t_yx[j] = ((unsigned)((t < G::NTASK && P < KPIX) ? py : 255) << 8) | (unsigned)px;
        t_xa[j] = P * G::XB + ((fh ^ key_x<CX>(P)) << 4);     // x fragment of k step 0; step s: chunk 2 s + fh -> XOR (2 s) << 4 into the address
        // middle patch: 8-byte piece q (channels 32 cb + 8 q + 4 fh .. + 3) of pixel P: chunk 4 cb + q, second half of the chunk for fh = 1
        t_mw[j] = G::MID_OFF + P * G::MB + 8 * fh;
    }
    // ---- phase-2 fragment addresses: A = middle patch pixel (row wm, column fr + v), k step s; tap row u adds u * KPW * MB (immediate) -----------
    // (k step 0 only; step s reads chunk (2 s + fh) ^ key = chunk(0) ^ 2 s: one XOR at the point of use instead of a register per step)
    int a_addr[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) a_addr[v] = G::MID_OFF + (wm * KPW + fr + v) * G::MB + ((fh ^ key_m<CM>(fr + v)) << 4);
    const int nw = n0 + wn * 32;                 // first output channel of this wave
    const float sc = p.scale[nw + fr], sh = p.shift[nw + fr];
    float *const S = scratch + wave * (16 * 32);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    // epilogue: lane -> (row r16 of a 16-row pass, 8-channel piece pc); shortcut = x patch pixel (wm + 1, 16 hp + r16 + 1), chunk (nw >> 3) + pc
    const int r16 = lane >> 2, pc = lane & 3;
    int sc_addr[2];
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        const int P = (wm + 1) * KPW + 16 * hp + r16 + 1;
        sc_addr[hp] = P * G::XB + ((((nw >> 3) + pc) ^ key_x<CX>(P)) << 4);
    }

    fetch_patch(st, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the first patch is in place
    int buf = 0;
    for (; st < n_spatial; st += sstep, buf ^= 1) {
        const bool more = st + sstep < n_spatial;
        if (more) fetch_patch(st + sstep, buf ^ 1);   // the other buffer: every wave left it at the barrier that opened this tile
        int b, ty, tx;
        tile_coords(st, b, ty, tx);
        const int gy0 = ty * KTH - 1, gx0 = tx * KTW - 1;
        const int xoff = buf * G::X_BYTES;

        // ---- phase 1: the 1x1 conv of the patch -> bf16 middle patch --------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (wave + 8 * j >= G::NTASK) continue;    // (wave-uniform) this wave has no such task
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < G::KS1; ++s) {
                const bf16x8 fx = *reinterpret_cast<const bf16x8 *>(lds + xoff + (t_xa[j] ^ ((2 * s) << 4)));
                const bf16x8 fw = s < KH ? w1frag[s < KH ? s : 0] : *reinterpret_cast<const bf16x8 *>(lds + (w1l ^ ((2 * (s - KH)) << 4)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx, acc, 0, 0, 0);   // operands exchanged: rows = channels, columns = pixels
            }
            const int t_py = (int)(t_yx[j] >> 8), t_px = (int)(t_yx[j] & 255u);
            const bool inside = (unsigned)(gy0 + t_py) < (unsigned)H && (unsigned)(gx0 + t_px) < (unsigned)W;
            const int km = key_m<CM>(t_px);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc1 = *reinterpret_cast<const f32x4 *>(c1 + 8 * q), sh1 = *reinterpret_cast<const f32x4 *>(c1 + CM + 8 * q);
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[4 * q + k] * sc1[k] + sh1[k];
                    if (p.blk.leaky) v[k] = fmaxf(v[k], 0.1f * v[k]);
                }
                u32x2 o;
                o[0] = inside ? pack_bf16_k(v[0], v[1]) : 0u;
                o[1] = inside ? pack_bf16_k(v[2], v[3]) : 0u;
                if (t_py != 255) *reinterpret_cast<u32x2 *>(lds + t_mw[j] + (((4 * cb + q) ^ km) << 4)) = o;   // (rows >= 208 of the last block: no such pixel)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the middle patch is complete (raw: the next patch's loads stay in flight)

        // ---- phase 2: the 3x3 conv from the middle patch -----------------------------------------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
#pragma unroll
                for (int s = 0; s < G::KS2; ++s) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(lds + (a_addr[v] ^ ((2 * s) << 4)) + u * (KPW * G::MB));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, wfrag[(u * 3 + v) * G::KS2 + s], acc, 0, 0, 0);
                }
            }

        // ---- per-wave epilogue, two passes of 16 rows: transpose through the private scratch, + shortcut (from the x patch) in fp32, one rounding ----
        const int oy = ty * KTH + wm;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {           // accumulator elements 8 hp .. 8 hp + 7 = rows 16 hp + 4 fh + (e & 3) + 8 (e >> 2)
                float v = acc[8 * hp + e] * sc + sh;
                if (p.leaky) v = fmaxf(v, 0.1f * v);
                S[(4 * fh + (e & 3) + 8 * (e >> 2)) * 32 + fr] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // lanes read what other lanes of this wave wrote (in-order LDS; pins the compiler)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(S + r16 * 32 + pc * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(S + r16 * 32 + pc * 8 + 4);
            const u32x4 xr = *reinterpret_cast<const u32x4 *>(lds + xoff + sc_addr[hp]);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = __uint_as_float(xr[k] << 16) + v[2 * k];
                v[2 * k + 1] = __uint_as_float(xr[k] & 0xffff0000u) + v[2 * k + 1];
            }
            u32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = pack_bf16_k(v[2 * k], v[2 * k + 1]);
            const int ox = tx * KTW + 16 * hp + r16;
            const bool live = oy < p.Ho && ox < p.Wo;
            // every lane ALWAYS issues its store (a dead pixel gets the out-of-range offset): the counted wait below relies on that
            const unsigned ooff = live ? (unsigned)(((b * p.Ho + oy) * p.Wo + ox) * p.Cout + nw + pc * 8) * 2u : p.dst_bytes;
            __builtin_amdgcn_raw_buffer_store_b128(out, rsd, (int)ooff, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // ... and the next pass's scratch writes stay below these reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // next tile: its x patch (requested at the top: OLDER in this wave's memory queue than this tile's two stores) has landed -- the stores
        // may still be in flight.  The barrier also ends every wave's reads of this tile's patches.
        if (more) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
}  // namespace

bool conv_block_bf16_fits(const ConvArgs &a)
{
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.src1 && (a.Cin == 32 || a.Cin == 64) && a.Cout == 2 * a.Cin && a.Cout == a.CoutPad &&
           a.H == a.Ho && a.W == a.Wo && a.K == 9 * a.Cin && a.residual != nullptr && a.dst != nullptr && a.blk.w != nullptr &&
           a.blk.scale != nullptr && a.blk.shift != nullptr;
}

hipError_t launch_conv_block_bf16(const ConvArgs &a, hipStream_t s)
{
    if (!conv_block_bf16_fits(a)) return hipErrorInvalidValue;
    const int tiles_x = (a.Wo + KTW - 1) / KTW, tiles_y = (a.Ho + KTH - 1) / KTH;
    const int n_spatial = a.B * tiles_y * tiles_x, slices = a.Cout / KSLICE;
    const int cus = a.n_cus > 0 ? a.n_cus : 256;                // read once at plan time (y3_net_plan)
    int per_slice = cus / slices;                               // one persistent workgroup per CU
    if (per_slice < 1) per_slice = 1;
    if (per_slice > n_spatial) per_slice = n_spatial;
    const int grid = per_slice * slices;
    if (a.Cout == 64) {
        static LdsAttrOnce attr;
        if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_block_bf16<64>), BlockGeom<64>::LDS_BYTES, a.device); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_block_bf16<64>, dim3(grid), dim3(KNT), BlockGeom<64>::LDS_BYTES, s, a, tiles_x, tiles_y, n_spatial, slices);
    } else {
        static LdsAttrOnce attr;
        if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_block_bf16<128>), BlockGeom<128>::LDS_BYTES, a.device); e != hipSuccess) return e;
        hipLaunchKernelGGL(conv_block_bf16<128>, dim3(grid), dim3(KNT), BlockGeom<128>::LDS_BYTES, s, a, tiles_x, tiles_y, n_spatial, slices);
    }
    return hipGetLastError();
}

}  // namespace y3
