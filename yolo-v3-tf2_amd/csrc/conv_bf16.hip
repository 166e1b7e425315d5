// bf16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate.
//
// BASELINE config 5 ("bf16 MFMA fused conv+BN+LeakyReLU"): same fused op as conv_f32.hip
// (reference: core/parse_model.py:27-52,72,134,155-156) with bf16 activations/weights in HBM and LDS.
//   * activations NHWC bf16, weights packed [CoutPad][K] bf16 (k = tap*Cin + c), head outputs fp32;
//   * K tile = BK bf16 (BK = 64: one 128-B line per row; BK = 32 for the two Cin = 32 layers); LDS rows are
//     2*BK + 16 bytes (odd number of 16-B slots -> conflict-free ds_read_b128 of 16 different rows), double buffered;
//   * MFMA 32x32x16: lane (r = l & 31, h = l >> 5) feeds A[row r][k = 16s + 8h .. +7] as one ds_read_b128;
//   * epilogue through LDS: accumulators (+scale/shift, leaky) are written as an fp32 [BM][BN+4] tile, then every
//     thread converts 8 consecutive channels (+ bf16 residual) and issues ONE 16-byte store -> full 128-B lines.
#include <type_traits>

#include "decode_box.h"
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

// DMA: operand tiles filled by direct-to-LDS buffer loads (BK = 64 only): unpadded 128-B rows, 16-B chunk index
// XOR-swizzled with (row >> 1) & 7 on the source address and on the fragment reads (see conv_f32.hip).
// DMA with BK = 32 (round 3): 64-B rows, a wave instruction fills 16 rows, chunk index XOR-ed with (row >> 2) & 3.  Half the
// LDS per stage: 128x256 / 256x128 tiles of 8 waves fit TWO workgroups per CU (MINW = 4 caps the registers at 128), so that
// one workgroup's epilogue (27 % of the bf16 conv stack, profiles/r03_ab_bf16_epilogue_probe.txt) runs beside the other's K loop.
// M16 (round 3): the same tile on v_mfma_f32_16x16x32_bf16 -- 2TM x 2TN blocks of 16x16 per wave instead of TM x TN of 32x32.
// Same LDS image, same bytes read per K tile, same MFMA cycles per FLOP; the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md "DVFS give-back" item 7: 1.12-1.15 x the FLOP/s on random data).  Fragment: lane l holds k = 8 (l >> 4) ..
// + 7 of row l & 15; C: acc[mb][nb][j] = row 16 mb + 4 (l >> 4) + j, column 16 nb + (l & 15).
#ifdef Y3_PHASE_STAMPS
// Diagnostic build only (csrc/build.py --variant ... -DY3_PHASE_STAMPS, tools/phase_stamps.py): thread 0 of every workgroup of the
// launches whose K equals y3_dbg_sel_k stores s_memrealtime (100 MHz) at kernel entry, before the first fetch, after the first
// barrier, after the K loop and after the epilogue, plus HW_ID / XCC_ID, into a buffer no other code reads.
__device__ unsigned long long y3_dbg_stamps[8 * 8192];
__device__ int y3_dbg_sel_k = -1;
#define Y3_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && p.K == y3_dbg_sel_k) y3_dbg_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Y3_STAMP(k) do { } while (0)
#endif

template <int TM, int TN, int WR, int WC, int BK, bool CONCAT, bool OUT_F32, bool DMA = false, int MINW = 1, bool M16 = false>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_bf16_mfma(const ConvArgs p)
{
    Y3_STAMP(0);
#ifdef Y3_PHASE_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 8192 && p.K == y3_dbg_sel_k) {
        y3_dbg_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        y3_dbg_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
#endif
    static_assert(!DMA || BK == 64 || BK == 32, "LDS-DMA variant needs 128-byte or 64-byte rows");
    static_assert(!M16 || (DMA && BK == 64), "the 16x16x32 form is built on the 128-byte swizzled rows");
    constexpr int MB = M16 ? 2 * TM : TM, NB = M16 ? 2 * TN : TN;   // accumulator blocks per wave
    using acc_t = typename std::conditional<M16, f32x4, f32x16>::type;
    constexpr int DROWS = 1024 / (2 * BK);   // rows one wave DMA instruction (1 KiB) fills: 8 (BK 64) or 16 (BK 32)
    constexpr int BM = 32 * TM * WR;
    constexpr int BN = 32 * TN * WC;
    constexpr int NT = 64 * WR * WC;
    constexpr int LPR = BK / 8;      // lanes per row (16 B = 8 bf16 each)
    constexpr int RP = NT / LPR;     // rows per load pass
    constexpr int AP = BM / RP, BP = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0 && AP >= 1 && BP >= 1, "tile too small for the thread count");
    constexpr int ROWB = DMA ? 2 * BK : 2 * BK + 16;  // LDS row bytes
    constexpr int STAGE_B = (BM + BN) * ROWB;      // bytes per stage
    constexpr int CROW = BN + 4;                   // floats per row of the epilogue tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tilesN = p.CoutPad / BN;
    const int mt = logical / tilesN, nt = logical - mt * tilesN;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(CONCAT ? p.src1 : p.src0), 0, CONCAT ? p.src1_bytes : p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);
    const unsigned OOB0 = p.src0_bytes, OOB1 = CONCAT ? p.src1_bytes : p.src0_bytes;

    const int lrow = tid / LPR;
    // first bf16 of this lane's 16-B piece inside the K tile (DMA: physical chunk tid & 7 holds logical chunk
    // (tid & 7) ^ ((row >> 1) & 7))
    const int lchunk = DMA ? (BK == 64 ? (((tid % LPR) ^ ((lrow >> 1) & 7)) * 8) : (((tid % LPR) ^ ((lrow >> 2) & 3)) * 8)) : (tid % LPR) * 8;
    int aoff[AP];
    int aoff1[CONCAT ? AP : 1];
    int ahw[AP];
    const int HoWo = p.Ho * p.Wo;
    const int C1 = p.Cin - p.C0;
    const int b0 = m0 / HoWo;
    const int r0 = m0 - b0 * HoWo;
    const int ho0 = r0 / p.Wo;
    const int wo0 = r0 - ho0 * p.Wo;
    const float rcpW = 1.0f / (float)p.Wo, rcpH = 1.0f / (float)p.Ho;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + i * RP + lrow;
        const int x = wo0 + i * RP + lrow;
        const int qx = (int)(((float)x + 0.5f) * rcpW);
        const int wo = x - qx * p.Wo;
        const int y = ho0 + qx;
        const int qy = (int)(((float)y + 0.5f) * rcpH);
        const int ho = y - qy * p.Ho;
        const int b = b0 + qy;
        if (CONCAT) {
            const int H0 = p.up0 ? (p.H >> 1) : p.H, W0 = p.up0 ? (p.W >> 1) : p.W;
            const int h0 = p.up0 ? (ho >> 1) : ho, w0 = p.up0 ? (wo >> 1) : wo;
            aoff[i] = ((b * H0 + h0) * W0 + w0) * p.C0;
            aoff1[i] = ((b * p.H + ho) * p.W + wo) * C1;
            ahw[i] = (m < p.M) ? 0 : (int)0x80000000;
        } else {
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = ((b * p.H + hi0) * p.W + wi0) * p.Cin;
            ahw[i] = (m < p.M) ? ((hi0 << 16) | (wi0 & 0xffff)) : (int)0x80000000;
        }
    }
    unsigned boff[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) boff[j] = (unsigned)((n0 + j * RP + lrow) * p.K + lchunk) * 2u;

    int tap = 0, c0 = 0;
    unsigned avoff[AP];
    unsigned avoff1[CONCAT ? AP : 1];
    auto set_tap = [&]() {
        if (CONCAT) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                avoff[i] = (ahw[i] < 0) ? OOB0 : (unsigned)(aoff[i] + lchunk) * 2u;
                avoff1[i] = (ahw[i] < 0) ? OOB1 : (unsigned)(aoff1[i] + lchunk) * 2u;
            }
        } else {
            const int u = tap / p.ksize, v = tap - u * p.ksize;
            const int toff = (u * p.W + v) * p.Cin + lchunk;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int hi = (ahw[i] >> 16) + u, wi = (int)(short)(ahw[i] & 0xffff) + v;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                avoff[i] = ok ? (unsigned)(aoff[i] + toff) * 2u : OOB0;
            }
        }
    };
    set_tap();

    u32x4 ra[AP], rb[BP];
    int kglob = 0;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch_dma = [&](int buf) {
        unsigned char *sa = smem + buf * STAGE_B + wave * DROWS * ROWB;   // wave w fills rows [pass*RP + DROWS*w, +DROWS)
        unsigned char *sb = sa + BM * ROWB;
        if (CONCAT && c0 >= p.C0) {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sa + i * RP * ROWB), 16, (int)avoff1[i], (c0 - p.C0) * 2, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * ROWB), 16, (int)avoff[i], c0 * 2, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * ROWB), 16, (int)boff[j], kglob * 2, 0, 0);
        kglob += BK;
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            if (!CONCAT) set_tap();
        }
    };
    auto fetch = [&]() {
        if (CONCAT) {
            if (c0 < p.C0) {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = bload16(rs0, avoff[i], c0 * 2);
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = bload16(rs1, avoff1[i], (c0 - p.C0) * 2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) ra[i] = bload16(rs0, avoff[i], c0 * 2);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) rb[j] = bload16(rsw, boff[j], kglob * 2);
        kglob += BK;
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            if (!CONCAT) set_tap();
        }
    };
    auto stage = [&](int buf) {
        unsigned char *sa = smem + buf * STAGE_B;
        unsigned char *sb = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<u32x4 *>(sa + (i * RP + lrow) * ROWB + lchunk * 2) = ra[i];
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<u32x4 *>(sb + (j * RP + lrow) * ROWB + lchunk * 2) = rb[j];
    };

    acc_t acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < (M16 ? 4 : 16); ++e) acc[i][j][e] = 0.0f;

    const int KT = p.K / BK;
    Y3_STAMP(1);
    if (DMA) {
        fetch_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        fetch();
        stage(0);
    }
    __syncthreads();
    Y3_STAMP(2);

    const int fr = M16 ? (lane & 15) : (lane & 31), fh = M16 ? (lane >> 4) : (lane >> 5);   // row in the block, k group
    const int a_frag = (wr * 32 * TM + fr) * ROWB + (DMA ? 0 : fh * 16);
    const int b_frag = BM * ROWB + (wc * 32 * TN + fr) * ROWB + (DMA ? 0 : fh * 16);
    constexpr int KS = M16 ? BK / 32 : BK / 16;      // MFMA k steps per K tile
    constexpr int BR = M16 ? 16 : 32;                // rows per block
    int foff[KS];  // byte offset of this lane's 16-B piece of k-step s inside its row
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
        foff[s_] = M16 ? (((4 * s_ + fh) ^ ((fr >> 1) & 7)) * 16)
                       : DMA ? (BK == 64 ? (((2 * s_ + fh) ^ ((fr >> 1) & 7)) * 16) : (((2 * s_ + fh) ^ ((fr >> 2) & 3)) * 16)) : s_ * 32;

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {
            if (DMA) fetch_dma(cur ^ 1); else fetch();
        }
        const unsigned char *sa = smem + cur * STAGE_B + a_frag;
        const unsigned char *sb = smem + cur * STAGE_B + b_frag;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 fa[MB], fb[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(sa + i * BR * ROWB + foff[s]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const bf16x8 *>(sb + j * BR * ROWB + foff[s]);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (kt + 1 < KT) {
            stage(cur ^ 1);
        }
        __syncthreads();
    }

    Y3_STAMP(3);
    // ---- epilogue through LDS, one 32-row sub-tile of every wave per pass ----------------------------------
    // pass i: wave (wr, wc) writes rows [wr*32, +32) x cols [wc*32*TN, +32*TN) of a [WR*32][BN+4] fp32 tile (its i-th
    // accumulator row block), then all threads convert 8 consecutive channels each and store 16 B.
    if constexpr (!OUT_F32) {
        // ---- bf16 output: per-wave epilogue, no workgroup barrier ------------------------------------------------------
        // Every wave transposes its own 32 x (32 TN) fp32 blocks through a private LDS scratch (the operand tiles are dead
        // after the loop's last barrier) and stores whole 16-byte pieces of 8 channels: rows of 64 TN bytes per wave, full
        // 128-B lines for TN >= 2.  LDS operations of one wave execute in order, so the write -> read hand-off between
        // its lanes needs no barrier, and a wave leaves as soon as ITS stores are issued.  The shortcut operand is added
        // in fp32 before the single rounding to bf16 (the oracle's bf16 mode rounds where the pipeline stores).
        // (round 3: the workgroup-wide fp32 tile + 2 barriers per pass this replaces cost 27 % of the bf16 conv stack,
        // profiles/r03_ab_bf16_epilogue_probe.txt)
        constexpr int CW = 32 * TN;              // floats per scratch row = channels per wave
        constexpr int PPRW = CW / 8;             // 16-byte output pieces per row
        constexpr int NPL = 32 * PPRW / 64;      // pieces per lane per 32-row block
        float *S = reinterpret_cast<float *>(smem) + wave * (32 * CW);
        unsigned short *dstb = static_cast<unsigned short *>(p.dst);
        const unsigned short *res = static_cast<const unsigned short *>(p.residual);
        const int nw = n0 + wc * CW;             // first channel of this wave
        float sc[NB], sh[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            sc[j] = p.scale[nw + j * BR + fr];
            sh[j] = p.shift[nw + j * BR + fr];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mw = m0 + (wr * TM + i) * 32;   // first row of this block
            u32x4 rr[NPL];
            // (round 5: all TM passes' shortcut loads requested up front, so that none queues behind a pass's stores, measured 0.6 % SLOWER,
            // profiles/r05_ab_bf16_shortcut_hoist.txt: 64 loads per CU in flight at once instead of 32 twice)
            if (res) {   // shortcut operand first: its latency hides behind the accumulator write-out
#pragma unroll
                for (int it = 0; it < NPL; ++it) {
                    const int q = lane + it * 64;
                    const int r = q / PPRW, pc = q - r * PPRW;
                    rr[it] = (mw + r < p.M) ? *reinterpret_cast<const u32x4 *>(res + (size_t)(mw + r) * p.Cout + nw + pc * 8)
                                            : u32x4{0u, 0u, 0u, 0u};
                }
            }
            if constexpr (M16) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[2 * i + mb][j][e] * sc[j] + sh[j];
                            if (p.leaky) v = fmaxf(v, 0.1f * v);
                            S[(16 * mb + 4 * fh + e) * CW + j * 16 + fr] = v;
                        }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[i][j][e] * sc[j] + sh[j];
                        if (p.leaky) v = fmaxf(v, 0.1f * v);
                        S[(4 * fh + (e & 3) + 8 * (e >> 2)) * CW + j * 32 + fr] = v;
                    }
                }
            }
            // lanes read what OTHER lanes of this wave wrote: the hardware runs a wave's LDS operations in order, and these three
            // builtins (no instructions) keep the compiler from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < NPL; ++it) {
                const int q = lane + it * 64;
                const int r = q / PPRW, pc = q - r * PPRW;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(S + r * CW + pc * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(S + r * CW + pc * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (res) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] = __uint_as_float(rr[it][k] << 16) + v[2 * k];
                        v[2 * k + 1] = __uint_as_float(rr[it][k] & 0xffff0000u) + v[2 * k + 1];
                    }
                }
                u32x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = pack_bf16(v[2 * k], v[2 * k + 1]);
                if (mw + r < p.M) *reinterpret_cast<u32x4 *>(dstb + (size_t)(mw + r) * p.Cout + nw + pc * 8) = out;
            }
            // ... and the next pass's writes below this pass's reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        Y3_STAMP(4);   // thread 0 = wave 0: its own stores issued (not yet retired)
    } else {
        // ---- fp32 output (head grids, Cout = 255): workgroup-wide fp32 tile, one 32-row block of every wave per pass ----
        constexpr int EROWS = WR * 32;
        float *C = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i > 0) __syncthreads();   // previous pass fully read
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nl = wc * TN * 32 + j * BR + fr;
                const float sc = p.scale[n0 + nl], sh = p.shift[n0 + nl];
                if constexpr (M16) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[2 * i + mb][j][e] * sc + sh;
                            if (p.leaky) v = fmaxf(v, 0.1f * v);
                            C[(wr * 32 + 16 * mb + 4 * fh + e) * CROW + nl] = v;
                        }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[i][j][e] * sc + sh;
                        if (p.leaky) v = fmaxf(v, 0.1f * v);
                        C[(wr * 32 + 4 * fh + (e & 3) + 8 * (e >> 2)) * CROW + nl] = v;
                    }
                }
            }
            __syncthreads();
            float *dst = static_cast<float *>(p.dst);
            if (dst != nullptr) {
                for (int idx = tid; idx < EROWS * BN; idx += NT) {
                    const int r = idx / BN, col = idx - r * BN;
                    const int m = m0 + (r >> 5) * 32 * TM + i * 32 + (r & 31), n = n0 + col;
                    if (m < p.M && n < p.Cout) dst[(size_t)m * p.Cout + n] = C[r * CROW + col];
                }
            }
            // detection head with its decode fused in (y3_net_forward_decode; the launcher guarantees a tile that spans all
            // 3 * (5 + nc) channels): the EROWS pixels of this pass are decoded from the fp32 tile in LDS, same body as decode.hip
            if (p.dec.boxes != nullptr)
                decode_rows_from_lds<NT>(C, CROW, EROWS, [&](int r) { return m0 + (r >> 5) * 32 * TM + i * 32 + (r & 31); }, p.M, p.dec);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round-3 record (code removed again): a PERSISTENT form of this kernel -- workgroups walking tiles bid, bid + G, ...,
// the next tile's first K tile prefetched before the epilogue, a counted s_waitcnt leaving the tile's stores in flight --
// was built to hide the output stores (profiles/r03_ab_bf16_epilogue_probe.txt: the stores alone cost 20 % of the bf16
// conv stack: 10.0 ms with them, 8.0 without).  Bit-identical, 34 bf16 parity tests green, and 10 % SLOWER (11.09 vs
// 10.07 ms, profiles/r03_ab_bf16_persistent.txt).  Two facts defeat it: (1) memory operations retire in issue order, so the
// first wait for a K tile issued after the stores waits for the stores as well -- they can overlap one K iteration
// (~1 us), not a tile; (2) with one workgroup per CU (128 KB of LDS) and equal work per tile all 256 workgroups store at
// the same moment: a 32 MB burst, 4 MB per XCD = the whole L2, which drains at the HBM write rate (~8 us) while HBM idles
// during the K loops.  What would help is out-of-phase workgroups (two per CU), which this tile's LDS and register
// budget do not admit.
// ---------------------------------------------------------------------------------------------------------
// Round-2 record (code removed in round 4): a pipelined 256x256x64 tile (id 20) with the K loop of the guide's "256^2 8-phase
// template" (LDS-DMA loads in flight across raw s_barriers, counted vmcnt, the second M half of the waves one barrier behind
// the first) was correct on the first build and 10-15 % SLOWER than the 16-wave two-phase tile 17
// (profiles/r02_tile_sweep_bf16_pipelined_b128_s416.txt, r02_bf16_pipelined_tile_pmc.txt: MFMA busy 0.45 vs 0.58).
// ---------------------------------------------------------------------------------------------------------
// tile table of the bf16 kernel: {BM, BN, waves, BK}
// Round 5: the table holds exactly the tiles a plan can select -- a packaged tuning table (tuning/bf16_*.json), the library's heuristic or
// the head-decode fallback (choose_tile_bf16 / run_slice in y3_api.cpp) names every one of them (tests/test_abi.py); the other ids are retired.
static const TileInfo kTilesBf16[BF16_TILE_COUNT] = {
    {128, 128, 4, 64}, {0, 0, 0, 64}, {0, 0, 0, 64}, {64, 64, 4, 64}, {128, 32, 4, 64},
    {128, 64, 4, 32}, {64, 64, 4, 32}, {0, 0, 0, 64},
    {128, 128, 4, 64}, {0, 0, 0, 64}, {128, 64, 4, 64}, {64, 64, 4, 64}, {64, 128, 4, 64}, {0, 0, 0, 64},  // 8..13: LDS-DMA
    {0, 0, 0, 64}, {0, 0, 0, 64}, {0, 0, 0, 64},                  // 14..16
    {256, 256, 16, 64}, {0, 0, 0, 64}, {128, 256, 16, 64},        // 17..19: LDS-DMA, 16 waves (64x64 / 32x64 wave tiles)
    {0, 0, 0, 64},                                                // 20
    {0, 0, 0, 32}, {256, 128, 8, 32}, {0, 0, 0, 32},              // 21..23: LDS-DMA with BK = 32, two workgroups per CU
    {256, 256, 16, 64}, {0, 0, 0, 64}, {128, 256, 16, 64}, {128, 128, 4, 64}, {0, 0, 0, 64}, {64, 128, 4, 64},  // 24..29: 16x16x32 MFMAs
    {0, 0, 0, 32}, {0, 0, 0, 32},                                 // 30, 31
    {128, 64, 8, 32},                                             // 32: weight-resident 3x3 / stride 1, Cin = 32 / 64 (conv_res_bf16.hip): 4 x 32 pixels x 64 channels per workgroup tile
    // 33..35 (3x3 / stride 1 with tap-row reuse) and 36 (256x256 on four waves of 128x128, hand-pipelined): parity-green, neutral / slower in the
    // two-lane step (profiles/r04_ab_bf16_rs.txt, r04_tile_sweep_bf16_w4.txt); code in the history (commit 1777145)
    {0, 0, 0, 64}, {0, 0, 0, 64}, {0, 0, 0, 64}, {0, 0, 0, 64},
};

#ifdef Y3_PHASE_STAMPS
extern "C" int y3_dbg_select_k(int K) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(y3_dbg_sel_k), &K, sizeof(int)); }
extern "C" int y3_dbg_copy_stamps(unsigned long long *dst, int n_words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y3_dbg_stamps), (size_t)n_words * sizeof(unsigned long long));
}
#endif

TileInfo conv_bf16_tile_info(int tile) { return kTilesBf16[(tile >= 0 && tile < BF16_TILE_COUNT) ? tile : 0]; }

bool conv_bf16_tile_built(int tile) { return tile >= 0 && tile < BF16_TILE_COUNT && kTilesBf16[tile].bm > 0; }

template <int TM, int TN, int WR, int WC, int BK, bool CONCAT, bool OUT_F32, bool DMA = false, int MINW = 1, bool M16 = false>
static hipError_t launch_kb(const ConvArgs &a, hipStream_t s)
{
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.CoutPad / BN;
    const size_t stages = 2 * (size_t)(BM + BN) * (DMA ? 2 * BK : 2 * BK + 16);
    // epilogue: fp32 output -> one workgroup-wide 32-row block per wave row; bf16 output -> 32 x (32 TN) floats per wave
    const size_t ctile = OUT_F32 ? (size_t)WR * 32 * (BN + 4) * sizeof(float) : (size_t)WR * WC * 32 * 32 * TN * sizeof(float);
    const size_t lds = stages > ctile ? stages : ctile;
    auto k = conv_bf16_mfma<TM, TN, WR, WC, BK, CONCAT, OUT_F32, DMA, MINW, M16>;
    static LdsAttrOnce attr;  // per instantiation
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(k), (int)lds, a.device); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN), dim3(64 * WR * WC), lds, s, a);
    return hipGetLastError();
}

template <int TM, int TN, int WR, int WC, int BK, bool DMA = false, int MINW = 1, bool M16 = false>
static hipError_t launch_tb(const ConvArgs &a, bool out_f32, hipStream_t s)
{
    if (a.src1)
        return out_f32 ? launch_kb<TM, TN, WR, WC, BK, true, true, DMA, MINW, M16>(a, s) : launch_kb<TM, TN, WR, WC, BK, true, false, DMA, MINW, M16>(a, s);
    return out_f32 ? launch_kb<TM, TN, WR, WC, BK, false, true, DMA, MINW, M16>(a, s) : launch_kb<TM, TN, WR, WC, BK, false, false, DMA, MINW, M16>(a, s);
}

hipError_t launch_conv_bf16(const ConvArgs &a, int tile, bool out_f32, hipStream_t s)
{
    if (!conv_bf16_tile_built(tile)) return hipErrorInvalidValue;
    const TileInfo t = kTilesBf16[tile];
    if (a.dec.boxes != nullptr && (!out_f32 || t.bn < a.CoutPad)) return hipErrorInvalidValue;   // a fused head needs all its channels in one tile
    if (a.dst == nullptr && a.dec.boxes == nullptr) return hipErrorInvalidValue;
    if (a.Cin % t.stages || a.CoutPad % t.bn || (a.src1 && a.C0 % t.stages)) return hipErrorInvalidValue;  // .stages holds BK
    if (tile == 32) return (!out_f32 && conv_res_bf16_fits(a)) ? launch_conv_res_bf16(a, s) : hipErrorInvalidValue;
    switch (tile) {
        case 0: return launch_tb<2, 2, 2, 2, 64>(a, out_f32, s);
        case 3: return launch_tb<1, 1, 2, 2, 64>(a, out_f32, s);
        case 4: return launch_tb<1, 1, 4, 1, 64>(a, out_f32, s);
        case 5: return launch_tb<2, 1, 2, 2, 32>(a, out_f32, s);
        case 6: return launch_tb<1, 1, 2, 2, 32>(a, out_f32, s);
        case 8: return launch_tb<2, 2, 2, 2, 64, true>(a, out_f32, s);    // 128x128 LDS-DMA
        case 10: return launch_tb<2, 1, 2, 2, 64, true>(a, out_f32, s);   // 128x64 LDS-DMA
        case 11: return launch_tb<1, 1, 2, 2, 64, true>(a, out_f32, s);   // 64x64 LDS-DMA
        case 12: return launch_tb<1, 2, 2, 2, 64, true>(a, out_f32, s);   // 64x128 LDS-DMA
        case 17: return launch_tb<2, 2, 4, 4, 64, true>(a, out_f32, s);   // 256x256, 16 waves, LDS-DMA
        case 19: return launch_tb<1, 2, 4, 4, 64, true>(a, out_f32, s);   // 128x256, 16 waves, LDS-DMA
        case 22: return launch_tb<2, 2, 4, 2, 32, true, 4>(a, out_f32, s);   // 256x128, 8 waves, BK 32, two workgroups per CU
        case 24: return launch_tb<2, 2, 4, 4, 64, true, 1, true>(a, out_f32, s);   // tile 17 on 16x16x32 MFMAs
        case 26: return launch_tb<1, 2, 4, 4, 64, true, 1, true>(a, out_f32, s);   // tile 19 on 16x16x32
        case 27: return launch_tb<2, 2, 2, 2, 64, true, 1, true>(a, out_f32, s);   // tile 8 (128x128, 4 waves) on 16x16x32
        case 29: return launch_tb<1, 2, 2, 2, 64, true, 1, true>(a, out_f32, s);   // tile 12 (64x128, 4 waves) on 16x16x32
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------------
// First layer in bf16 mode: fp32 image in, fp32 arithmetic (K = 27), bf16 out.  Same structure as
// conv_first_f32; the LDS transpose lets 4 lanes write one pixel's 32 bf16 channels (64 B) as 16-B stores.
// ---------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_first_bf16(const ConvArgs p, const float *__restrict__ w)
{
    constexpr int ROW = COUT + 4;
    __shared__ __attribute__((aligned(16))) float tr[4][64 * ROW];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int mw = blockIdx.x * 256 + wave * 64;
    const int m = mw + lane;
    const int HW = p.H * p.W;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    const int b = mm / HW;
    const int r = mm - b * HW;
    const int ho = r / p.W, wo = r - ho * p.W;
    const float *x = static_cast<const float *>(p.src0);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[COUT / 2];   // packed pairs: v_pk_fma_f32 retires two MACs per VALU instruction
#pragma unroll
    for (int n = 0; n < COUT / 2; ++n) acc2[n] = f32x2{0.0f, 0.0f};
#pragma unroll 1
    for (int u = 0; u < 3; ++u) {
        const int hi = ho - 1 + u;
#pragma unroll 1
        for (int v = 0; v < 3; ++v) {
            const int wi = wo - 1 + v;
            const bool ok = live && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float *xp = x + ((size_t)(b * p.H + (ok ? hi : 0)) * p.W + (ok ? wi : 0)) * 3;
            float xv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) xv[c] = ok ? xp[c] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *wr = w + ((u * 3 + v) * 3 + c) * COUT;
#pragma unroll
                for (int n = 0; n < COUT; n += 2)
                    acc2[n / 2] = __builtin_elementwise_fma(f32x2{xv[c], xv[c]}, f32x2{wr[n], wr[n + 1]}, acc2[n / 2]);
            }
        }
    }
    float *t = tr[wave];
#pragma unroll
    for (int n = 0; n < COUT; n += 4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc2[(n + e) / 2][(n + e) & 1] * p.scale[n + e] + p.shift[n + e];
            if (p.leaky) v = fmaxf(v, 0.1f * v);
            o[e] = v;
        }
        *reinterpret_cast<f32x4 *>(t + lane * ROW + n) = o;
    }
    unsigned short *dst = static_cast<unsigned short *>(p.dst);
    constexpr int CH = COUT / 8;   // 16-B pieces (8 bf16) per pixel
    constexpr int PPI = 64 / CH;   // pixels per store instruction
    const int c8 = lane % CH, pl = lane / CH;
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int px = it * PPI + pl;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(t + px * ROW + c8 * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(t + px * ROW + c8 * 8 + 4);
        u32x4 o;
        o[0] = pack_bf16(v0[0], v0[1]);
        o[1] = pack_bf16(v0[2], v0[3]);
        o[2] = pack_bf16(v1[0], v1[1]);
        o[3] = pack_bf16(v1[2], v1[3]);
        if (mw + px < p.M) *reinterpret_cast<u32x4 *>(dst + (size_t)(mw + px) * COUT + c8 * 8) = o;
    }
}

hipError_t launch_conv_first_bf16(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s)
{
    if (a.Cin != 3 || a.ksize != 3 || a.stride != 1 || a.Cout != 32 || a.residual || a.src1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_first_bf16<32>, dim3((a.M + 255) / 256), dim3(256), 0, s, a, w_hwio_dev);
    return hipGetLastError();
}

// bf16 -> fp32 copy (y3_net_read_tensor in bf16 mode)
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const unsigned short *x, float *y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = __uint_as_float((unsigned)x[i] << 16);
}

hipError_t launch_bf16_to_f32(const void *x, float *y, size_t n, hipStream_t s)
{
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s,
                       static_cast<const unsigned short *>(x), y, n);
    return hipGetLastError();
}

}  // namespace y3
