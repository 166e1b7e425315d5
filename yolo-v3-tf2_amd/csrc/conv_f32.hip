// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces Conv2D -> BatchNormalization -> LeakyReLU(0.1) [-> Add] of the reference graph
// (reference: core/parse_model.py:27-52,155-156) and the UpSampling2D + Concatenate feeding the
// lateral 1x1 convs (reference: core/parse_model.py:72,134) with ONE launch per conv.
//
// GEMM view: M = B*Ho*Wo output pixels (NHWC order), N = Cout, K = taps*Cin with k = tap*Cin + c.
//   A[m][k]  = input pixel (ho*s-pad+u, wo*s-pad+v) channel c, gathered on the fly (zero outside)
//   B[n][k]  = packed weights [CoutPad][K]
// Block tile BM x BN x 32, WR x WC waves (64*WR*WC threads), wave tile (32*TM) x (32*TN).
// Both operand tiles live in LDS as [rows][32+4] floats (K contiguous, one 16-B pad per row:
// row stride 9 x 16 B makes every ds_read_b128 of 16 different rows conflict-free), filled with
// 16-B buffer loads -> ds_write_b128 and double buffered (one barrier per K tile).
// A lane reads 4 consecutive k of its row as one ds_read_b128 and feeds them to 4 MFMAs; lanes
// 0-31 / 32-63 take k = 8q+t / 8q+4+t in MFMA (q,t), identically for A and B, so every k is
// contracted exactly once (the order of k inside a tile is irrelevant to the sum's value up to
// fp32 rounding).
// Epilogue from the accumulators: y = acc*scale[n] + shift[n]; leaky; + residual; store.
#include <algorithm>
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;
static constexpr int LDS_ROW_PADDED = BK + 4;  // floats (register-staged variant)

// 16-byte buffer load: per-lane voffset (range-checked against num_records) + wave-uniform soffset
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}

// DMA != 0: operand tiles are filled by direct-to-LDS buffer loads (no VGPR round trip, no ds_write): measured with
// tools/mfma_probe.hip, a 16-B load to VGPRs costs the SIMD ~8-16 cycles of matrix-pipe time and a ds_write_b128 ~13,
// an LDS-DMA load ~4.  LDS rows are then unpadded 128 B (a wave instruction writes 8 whole rows) and bank conflicts
// are avoided by an XOR swizzle of the 16-B chunk index applied on the SOURCE address and on the fragment reads.
//
// Removed in round 4 (records under profiles/r01_*, r02_*, code in the history): the timing-only PROBE ablations, the
// persistent stream-K schedule (neutral on 3x3, -10 % on 1x1 layers: profiles/r02_tile_sweep_f32_streamk_b64_s416.txt) and the
// residual-prefetch tiles (conv stack -0.3 %: profiles/r02_residual_prefetch_ab.txt).
#ifdef Y3_PHASE_STAMPS
// Diagnostic build only (csrc/build.py --variant ... -DY3_PHASE_STAMPS, tools/phase_stamps.py --dtype f32): thread 0 of every workgroup
// (up to 32768) of the launches whose K equals y3_dbg32_sel_k stores s_memrealtime (100 MHz) at kernel entry, before the first fetch,
// after the first barrier, after the K loop and after the epilogue, plus HW_ID / XCC_ID, into a buffer no other code reads.
__device__ unsigned long long y3_dbg32_stamps[8 * 32768];
__device__ int y3_dbg32_sel_k = -1;
#define Y3_STAMP32(k) do { if (threadIdx.x == 0 && blockIdx.x < 32768 && p.K == y3_dbg32_sel_k) y3_dbg32_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Y3_STAMP32(k) do { } while (0)
#endif

template <int TM, int TN, int WR, int WC, bool CONCAT, int STAGES, int MINW = 1, int DMA = 0>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_f32_mfma(const ConvArgs p)
{
    Y3_STAMP32(0);
#ifdef Y3_PHASE_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 32768 && p.K == y3_dbg32_sel_k) {
        y3_dbg32_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        y3_dbg32_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
#endif
    constexpr int LDS_ROW = DMA ? BK : BK + 4;  // floats per LDS row
    constexpr int BM = 32 * TM * WR;
    constexpr int BN = 32 * TN * WC;
    constexpr int NT = 64 * WR * WC;
    constexpr int RP = NT / 8;   // rows per load pass (8 lanes x 16 B cover one 32-float row)
    constexpr int AP = BM / RP;  // A load passes
    constexpr int BP = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0 && AP >= 1 && BP >= 1, "tile too small for the thread count");
    constexpr int STAGE = (BM + BN) * LDS_ROW;  // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    if (p.clk_stamps != nullptr && blockIdx.x == (gridDim.x >> 1) && tid == 0) {   // measurement launches only (y3_net_measure_sclk): a workgroup of the launch's steady state
        p.clk_stamps[0] = __builtin_amdgcn_s_memtime();
        p.clk_stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
    if (p.clk_stamps != nullptr && blockIdx.x == 0 && tid == 0) p.clk_stamps[4] = __builtin_amdgcn_s_memrealtime();   // the launch's first workgroup: when the kernel began
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    // run of logical tiles so that the N-tiles of one pixel tile and neighbouring pixel tiles meet
    // in one L2.  Bijective for any grid size.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tilesN = p.CoutPad / BN;
    const int KT = p.K / BK;
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(CONCAT ? p.src1 : p.src0), 0, CONCAT ? p.src1_bytes : p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);
    // a voffset equal to num_records is out of range for the buffer's bounds check -> the load returns 0
    const unsigned OOB0 = p.src0_bytes, OOB1 = CONCAT ? p.src1_bytes : p.src0_bytes;

    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.residual ? p.residual : p.dst), 0, p.dst_bytes, 0x00020000);
    const int fr = lane & 31, fh = lane >> 5;

    const int tile = logical;
    int mt = tile / tilesN, nt = tile - mt * tilesN;
    if (p.xcd_gn > 0) {
        // XCD-blocked order (launch_k sizes the grid for it): the 8 XCDs form a (8/gn) x gn grid over the tile matrix;
        // XCD (xm, xn) owns M-tiles [xm*tilesM/gm, (xm+1)*tilesM/gm) x N-tiles [xn*tilesN/gn, +tilesN/gn), N fastest.
        // With gn > 1 an XCD streams only 1/gn of the weight matrix through its 4 MB L2 (the 256->512 and 512->1024
        // 3x3 weights are 4.7 / 18.9 MB) at the price of gn XCDs reading every activation tile.
        const int gn = p.xcd_gn, gm = 8 / gn;
        const int tilesM = (p.M + BM - 1) / BM;
        const int xm = xcd / gn, xn = xcd - xm * gn;
        const int nb = tilesN / gn;
        const int mlo = xm * tilesM / gm, mhi = (xm + 1) * tilesM / gm;
        const int j = bid >> 3;
        const int lm = j / nb;
        mt = mlo + lm;
        nt = xn * nb + (j - lm * nb);
        if (mt >= mhi) return;                 // padding workgroups of an uneven M split
    }
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread gather state -------------------------------------------------------------
    const int lrow = tid >> 3;         // row inside a pass
    // first float of this lane's 16-B piece inside the 32-float K tile (DMA: the piece that lands in physical chunk
    // tid & 7 of the row is logical chunk (tid & 7) ^ ((row >> 1) & 7); rows of a pass differ by multiples of 32)
    const int lchunk = DMA ? (((tid & 7) ^ ((lrow >> 1) & 7)) * 4) : (tid & 7) * 4;
    int aoff[AP];                      // element offset of (b, hi0, wi0, 0) in src0 (may be negative)
    int aoff1[CONCAT ? AP : 1];        // CONCAT: element offset of (b, ho, wo, 0) in src1
    int ahw[AP];                       // hi0 << 16 | (wi0 & 0xffff); row >= M marked by hi0 = -32768
    const int HoWo = p.Ho * p.Wo;
    const int C1 = p.Cin - p.C0;
    // (b, ho, wo) of row m: the tile's first row is decomposed with wave-uniform (scalar) divisions; the lane's
    // displacement (< BM + Wo) is folded in with an exact small float division -- vector integer division costs
    // ~40 VALU instructions each, and VALU time is lost MFMA time for every wave on the SIMD.
    const int b0 = m0 / HoWo;
    const int r0 = m0 - b0 * HoWo;
    const int ho0 = r0 / p.Wo;
    const int wo0 = r0 - ho0 * p.Wo;
    const float rcpW = 1.0f / (float)p.Wo, rcpH = 1.0f / (float)p.Ho;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + i * RP + lrow;
        const int x = wo0 + i * RP + lrow;                       // < BM + Wo <= 1024: (x+0.5)*rcp is exact
        const int qx = (int)(((float)x + 0.5f) * rcpW);
        const int wo = x - qx * p.Wo;
        const int y = ho0 + qx;
        const int qy = (int)(((float)y + 0.5f) * rcpH);
        const int ho = y - qy * p.Ho;
        const int b = b0 + qy;
        if (CONCAT) {
            // 1x1 conv: src0 optionally read through nearest x2 up-sampling
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
    unsigned boff[BP];  // byte offset of this lane's piece of weight row n, k = 0
#pragma unroll
    for (int j = 0; j < BP; ++j) boff[j] = (unsigned)((n0 + j * RP + lrow) * p.K + lchunk) * 4u;

    // walking state of the *next* K tile to fetch.  The fp32 MFMA does not co-execute with VALU work
    // (SQ_VALU_MFMA_COEXEC_CYCLES == 0 in the profile), so the K loop must issue (almost) no vector ALU
    // instructions: per-lane byte offsets are fixed per tap (out-of-image lanes hold the out-of-range
    // sentinel) and everything that changes per K tile goes into the scalar soffset of the buffer load.
    // K order.  Classic (k_chunk == 0): k = tap * Cin + c walked upwards.  Chunked (k_chunk = CK channels, 3x3 convs): for
    // every chunk of CK input channels all taps, then the next chunk -- the SAME products summed in another order (weights stay
    // packed [n][tap * Cin + c]; only the soffsets change).  The 9 reads of a pixel's CK channels (one per tap) then fall into
    // 9 * CK / 32 consecutive K tiles instead of being Cin / 32 tiles apart, so the tap re-reads hit in L2: with the classic
    // order a 512 -> 1024 @13 launch fetched 1.5-1.9 GB from beyond L2 for 85 MB of operands (tools/traffic_per_layer.py).
    const int CK = (!CONCAT && p.k_chunk > 0 && p.k_chunk < p.Cin) ? p.k_chunk : p.Cin;
    int kglob = 0;  // k index of the next tile to fetch
    int tap = 0;
    int c0 = 0;
    int cend = CK;            // end of the current channel chunk (classic order: Cin)
    const int taps = p.ksize * p.ksize;
    unsigned avoff[AP];                  // voffset of this lane's piece for the current tap (or OOB0)
    unsigned okmask[CONCAT ? 1 : AP];    // !CONCAT: bit t = tap t of this row lies inside the image
    unsigned abase4[CONCAT ? 1 : AP];    // !CONCAT: byte offset of this lane's piece of the row at tap (0, 0), channel 0
    unsigned avoff1[CONCAT ? AP : 1];    // CONCAT: same for src1
    auto set_tap = [&]() {
        if (CONCAT) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                avoff[i] = (ahw[i] < 0) ? OOB0 : (unsigned)(aoff[i] + lchunk) * 4u;
                avoff1[i] = (ahw[i] < 0) ? OOB1 : (unsigned)(aoff1[i] + lchunk) * 4u;
            }
        } else {
            // 4 vector instructions per row (bit test, compare, add, select): with the chunk-major K order this runs every
            // CK / 32 K tiles, and vector ALU time is lost MFMA time
            const int u = tap / p.ksize, v = tap - u * p.ksize;
            const unsigned toff4 = (unsigned)((u * p.W + v) * p.Cin) * 4u;
#pragma unroll
            for (int i = 0; i < AP; ++i) avoff[i] = ((okmask[i] >> tap) & 1u) ? abase4[i] + toff4 : OOB0;
        }
    };
    if (!CONCAT) {
        // per row: bit t of okmask = tap t reads inside the image (rows >= M: no bit set); byte offset of the lane's piece at tap 0
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            unsigned mk = 0;
            if (ahw[i] >= 0 || (ahw[i] >> 16) != -32768) {
                const int hi0 = ahw[i] >> 16, wi0 = (int)(short)(ahw[i] & 0xffff);
                if (p.ksize == 3) {
                    // closed form of the loop below: row u contributes bits 3u..3u+2, column v bit v of each row.  With the SIMD full
                    // of 64-cycle MFMAs every vector instruction of a prologue waits ~one MFMA for its issue slot
                    // (tools/phase_stamps.py --dtype f32: 32 us from entry to the first fetch): six compares instead of two nested loops
                    const unsigned H = (unsigned)p.H, W = (unsigned)p.W;
                    const unsigned rm = ((unsigned)hi0 < H ? 7u : 0u) | ((unsigned)(hi0 + 1) < H ? 56u : 0u) | ((unsigned)(hi0 + 2) < H ? 448u : 0u);
                    const unsigned cm = ((unsigned)wi0 < W ? 1u : 0u) | ((unsigned)(wi0 + 1) < W ? 2u : 0u) | ((unsigned)(wi0 + 2) < W ? 4u : 0u);
                    mk = rm & (cm * 73u);
                } else {
                    for (int u = 0; u < p.ksize; ++u)
                        for (int v = 0; v < p.ksize; ++v)
                            if ((unsigned)(hi0 + u) < (unsigned)p.H && (unsigned)(wi0 + v) < (unsigned)p.W) mk |= 1u << (u * p.ksize + v);
                }
            }
            okmask[i] = mk;
            abase4[i] = (unsigned)(aoff[i] + lchunk) * 4u;
        }
    }
    set_tap();

    f32x4 ra[AP], rb[BP];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch_dma = [&](int buf) {
        // wave w fills rows [pass*RP + 8w, +8) of each tile: LDS destination = M0 base + lane*16
        float *sa = smem + buf * STAGE + wave * 8 * LDS_ROW;
        float *sb = sa + BM * LDS_ROW;
        if (CONCAT && c0 >= p.C0) {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sa + i * RP * LDS_ROW), 16, (int)avoff1[i], (c0 - p.C0) * 4, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * LDS_ROW), 16, (int)avoff[i], c0 * 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * LDS_ROW), 16, (int)boff[j], kglob * 4, 0, 0);
        c0 += BK;
        if (c0 == cend) {
            ++tap;
            if (!CONCAT && tap == taps && cend != p.Cin) {   // chunked order: next channel chunk, first tap again
                tap = 0;
                cend += CK;
            }
            c0 = cend - CK;
            if (!CONCAT) set_tap();
        }
        kglob = tap * p.Cin + c0;
    };
    auto fetch = [&]() {
        if (CONCAT) {
            // channels [0,C0) come from src0, [C0,Cin) from src1; a 32-wide K tile never straddles (C0 % 32 == 0)
            if (c0 < p.C0) {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = buf_load16(rs0, avoff[i], c0 * 4);
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = buf_load16(rs1, avoff1[i], (c0 - p.C0) * 4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) ra[i] = buf_load16(rs0, avoff[i], c0 * 4);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) rb[j] = buf_load16(rsw, boff[j], kglob * 4);
        c0 += BK;
        if (c0 == cend) {
            ++tap;
            if (!CONCAT && tap == taps && cend != p.Cin) {   // chunked order: next channel chunk, first tap again
                tap = 0;
                cend += CK;
            }
            c0 = cend - CK;
            if (!CONCAT) set_tap();
        }
        kglob = tap * p.Cin + c0;
    };
    auto stage = [&](int buf) {
        float *sa = smem + buf * STAGE;
        float *sb = sa + BM * LDS_ROW;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4 *>(sa + (i * RP + lrow) * LDS_ROW + lchunk) = ra[i];
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<f32x4 *>(sb + (j * RP + lrow) * LDS_ROW + lchunk) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    Y3_STAMP32(1);
    if (DMA) {
        fetch_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        fetch();
        stage(0);
    }
    __syncthreads();
    Y3_STAMP32(2);

    const int a_frag = (wr * 32 * TM + fr) * LDS_ROW + (DMA ? 0 : fh * 4);
    const int b_frag = BM * LDS_ROW + (wc * 32 * TN + fr) * LDS_ROW + (DMA ? 0 : fh * 4);
    int foff[4];  // float offset of this lane's k-chunk q inside its row
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = DMA ? ((((2 * q + fh) ^ ((fr >> 1) & 7)) * 4)) : q * 8;

    // epilogue geometry
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.Cout);
    const int row_bytes = p.Cout * 4;

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = (STAGES == 2) ? (kt & 1) : 0;
        if (DMA && STAGES == 2) {
            if (kt + 1 < KT) fetch_dma(cur ^ 1);   // every wave passed the barrier that ended tile kt-1: buf cur^1 is free
        } else if (!DMA) {
            if (kt + 1 < KT) fetch();
        }
        const float *sa = smem + cur * STAGE + a_frag;
        const float *sb = smem + cur * STAGE + b_frag;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(sa + i * 32 * LDS_ROW + foff[q]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(sb + j * 32 * LDS_ROW + foff[q]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
            // (s_setprio 1 / 0 around this block: -0.6 % on the conv stack, tools/ab_libs.py, round 2)
        }
        if (DMA && STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt+1 has landed (issued a whole K tile ago)
            __syncthreads();
        } else if (DMA) {
            if (kt + 1 < KT) {
                __syncthreads();   // every wave is done reading the single buffer
                fetch_dma(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        } else if (STAGES == 2) {
            if (kt + 1 < KT) stage(cur ^ 1);
            __syncthreads();
        } else if (kt + 1 < KT) {
            __syncthreads();  // every wave is done reading the tile
            stage(0);
            __syncthreads();
        }
    }

    Y3_STAMP32(3);
    // ---- epilogue ----------------------------------------------------------------------------
    // accumulator element e of lane l: column (n) = l & 31, row (m) = (e & 3) + 8*(e >> 2) + 4*(l >> 5).
    // Straight-line: out-of-tile elements get a voffset == num_records, which the buffer bounds check
    // turns into "load 0 / drop the store"; the 16 residual loads of a sub-tile are issued together.
    // interior tiles (the common case): one per-lane voffset per sub-tile, the row displacement of accumulator
    // element e rides in the scalar soffset -> no per-element address or bounds arithmetic on the VALU
    auto emit = [&](auto leaky_tag, auto res_tag, auto interior_tag) {
        constexpr bool LEAKY = decltype(leaky_tag)::value, RES = decltype(res_tag)::value;
        // shadows the run-time flag: straight-line code per case (the run-time form compiled to a branch around every
        // load and store; A/B r03: conv stack -0.07 %, i.e. neutral -- kept for the shorter instruction stream)
        constexpr bool interior = decltype(interior_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wc * TN + j) * 32 + fr;
            const float sh = p.shift[n];   // the BN scale is folded into the packed weights (y3_api.cpp)
            const bool n_ok = n < p.Cout;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mbase = m0 + (wr * TM + i) * 32 + 4 * fh;
                const unsigned vbase = (unsigned)(mbase * p.Cout + n) * 4u;
                unsigned off[16];
                if (!interior) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mbase + (e & 3) + 8 * (e >> 2);
                        off[e] = (n_ok && m < p.M) ? (unsigned)(m * p.Cout + n) * 4u : p.dst_bytes;
                    }
                }
                float r[16];
                if (RES) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int so = ((e & 3) + 8 * (e >> 2)) * row_bytes;
                        r[e] = __builtin_bit_cast(float, interior ? __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)vbase, so, 0)
                                                                  : __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)off[e], 0, 0));
                    }
                }
                // element pairs as 2-vectors: the compiler selects v_pk_add_f32 / v_pk_mul_f32 (one instruction per pair, the same
                // IEEE operation per element): with the SIMD full of 64-cycle MFMAs every vector instruction of an epilogue
                // waits ~one MFMA for its issue slot, so the count of instructions is what the epilogue costs
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 v2 = f32x2{acc[i][j][e], acc[i][j][e + 1]} + f32x2{sh, sh};
                    if (LEAKY) {
                        const f32x2 t2 = v2 * f32x2{0.1f, 0.1f};
                        v2 = f32x2{fmaxf(v2[0], t2[0]), fmaxf(v2[1], t2[1])};   // == (v >= 0 ? v : 0.1 v) for every finite v
                    }
                    if (RES) v2 = f32x2{r[e], r[e + 1]} + v2;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ee = e + h;
                        const int so = ((ee & 3) + 8 * (ee >> 2)) * row_bytes;
                        if (interior)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v2[h]), rsd, (int)vbase, so, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v2[h]), rsd, (int)off[ee], 0, 0);
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto emit2 = [&](auto leaky_tag, auto res_tag) {
        if (interior) emit(leaky_tag, res_tag, T_{}); else emit(leaky_tag, res_tag, F_{});
    };
    if (p.residual) {
        if (p.leaky) emit2(T_{}, T_{}); else emit2(F_{}, T_{});
    } else {
        if (p.leaky) emit2(T_{}, F_{}); else emit2(F_{}, F_{});
    }
    Y3_STAMP32(4);   // thread 0 = wave 0: its own stores issued (not yet retired)
    if (p.clk_stamps != nullptr && blockIdx.x == (gridDim.x >> 1) && tid == 0) {
        p.clk_stamps[2] = __builtin_amdgcn_s_memtime();
        p.clk_stamps[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// tile table: {BM, BN, waves, LDS stages}; ids are stable (tuning files refer to them).  Ids 20..22 and 25 were the
// timing-only probes of rounds 1-2 (removed in round 4; the ids stay reserved), 33..45 the stream-K / residual-prefetch tiles.
// Round 5: the table holds exactly the tiles a plan can select -- a packaged tuning table (tuning/f32_*.json) or the library's heuristic
// (choose_tile in y3_api.cpp) names every one of them (tests/test_abi.py).  The other ids of rounds 1-4 (two-stage forms of 0..5, the 8- and
// 16-wave 128x128 / 256x128 tiles, the register-budget variant 24, the two-stage LDS-DMA tiles 28..30, the ablations 20..22, 25) are retired:
// {0,0,0,0}, y3_tile_built answers 0; the kernel template itself is general and the sweeps that retired them are under profiles/.
static const TileInfo kTiles[TILE_COUNT] = {
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},                                  // 0..5
    {128, 128, 4, 1}, {0, 0, 0, 0}, {256, 32, 4, 1}, {128, 64, 4, 1}, {64, 128, 4, 1}, {64, 64, 4, 1},                 // 6..11: single LDS stage
    {128, 128, 8, 1}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},                                                          // 12..15
    {0, 0, 0, 0}, {128, 64, 8, 1}, {0, 0, 0, 0}, {0, 0, 0, 0},                                                           // 16..19
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},                                                                            // 20..22
    {128, 128, 4, 1}, {0, 0, 0, 0},                      // 23: 128x128 with the register budget of 3 waves per SIMD
    {0, 0, 0, 0},                                        // 25
    {64, 128, 4, 2}, {64, 64, 4, 2}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},   // 26, 27: LDS-DMA loads, two stages
    {64, 128, 4, 1}, {64, 64, 4, 1},                     // 31, 32: LDS-DMA, single stage
    {128, 64, 8, 2},                                     // 33: weight-resident 3x3 / stride 1 / Cin = 32 (conv_res_f32.hip): 8 x 16 pixels x 64 channels per workgroup tile
};

#ifdef Y3_PHASE_STAMPS
extern "C" int y3_dbg32_select_k(int K) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(y3_dbg32_sel_k), &K, sizeof(int)); }
extern "C" int y3_dbg32_copy_stamps(unsigned long long *dst, int n_words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y3_dbg32_stamps), (size_t)n_words * sizeof(unsigned long long));
}
#endif

TileInfo conv_tile_info(int tile) { return kTiles[(tile >= 0 && tile < TILE_COUNT) ? tile : 0]; }

bool conv_tile_built(int tile) { return tile >= 0 && tile < TILE_COUNT && kTiles[tile].bm > 0; }

template <int TM, int TN, int WR, int WC, bool CONCAT, int STAGES, int MINW = 1, int DMA = 0>
static hipError_t launch_k(const ConvArgs &a_in, hipStream_t s)
{
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    ConvArgs a = a_in;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.CoutPad / BN;
    const size_t lds = STAGES * (size_t)(BM + BN) * (DMA ? BK : LDS_ROW_PADDED) * sizeof(float);
    auto k = conv_f32_mfma<TM, TN, WR, WC, CONCAT, STAGES, MINW, DMA>;
    static LdsAttrOnce attr;  // per instantiation
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(k), (int)lds, a.device); e != hipSuccess) return e;
    int grid = tilesM * tilesN;
    if (a.xcd_gn > 0) {
        if (8 % a.xcd_gn || tilesN % a.xcd_gn) return hipErrorInvalidValue;
        const int gm = 8 / a.xcd_gn;
        int rows = 0;                                           // largest M block
        for (int xm = 0; xm < gm; ++xm) rows = std::max(rows, (xm + 1) * tilesM / gm - xm * tilesM / gm);
        grid = 8 * rows * (tilesN / a.xcd_gn);
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WR * WC), lds, s, a);
    return hipGetLastError();
}

// single LDS stage, register-staged operand loads.  MINW: register budget (waves per SIMD) -- with 4, the two accumulators of the 32x64 wave
// tile stay in architectural VGPRs and the epilogue needs no v_accvgpr_read (nor the prologue 32-64 v_accvgpr_write: with the SIMD full of
// 64-cycle MFMAs every vector instruction outside the K loop waits ~one MFMA for its issue slot, profiles/r03_ab_f32_prologue.txt)
template <int TM, int TN, int WR, int WC, int MINW = 1>
static hipError_t launch_s1(const ConvArgs &a, hipStream_t s)
{
    return a.src1 ? launch_k<TM, TN, WR, WC, true, 1, MINW>(a, s) : launch_k<TM, TN, WR, WC, false, 1, MINW>(a, s);
}

hipError_t launch_conv_f32(const ConvArgs &a, int tile, hipStream_t s)
{
    if (!conv_tile_built(tile)) return hipErrorInvalidValue;
    if (tile == 33) return conv_res_f32_fits(a) ? launch_conv_res_f32(a, s) : hipErrorInvalidValue;
    switch (tile) {
        case 6: return launch_s1<2, 2, 2, 2>(a, s);        // 128x128, 4 waves
        case 8: return launch_s1<2, 1, 4, 1>(a, s);        // 256x32
        case 9: return launch_s1<1, 2, 4, 1, 4>(a, s);     // 128x64
        case 10: return launch_s1<1, 2, 2, 2, 4>(a, s);    // 64x128
        case 11: return launch_s1<1, 1, 2, 2, 4>(a, s);    // 64x64
        case 12: return launch_s1<2, 1, 2, 4>(a, s);       // 128x128, 8 waves
        case 17: return launch_s1<1, 1, 4, 2>(a, s);       // 128x64, 8 waves
        case 23: return launch_s1<2, 2, 2, 2, 3>(a, s);    // 128x128 within 3 waves per SIMD of registers
        // direct-to-LDS operand loads
        case 26: return a.src1 ? launch_k<1, 2, 2, 2, true, 2, 4, 1>(a, s) : launch_k<1, 2, 2, 2, false, 2, 4, 1>(a, s);  // 64x128, two stages
        case 27: return a.src1 ? launch_k<1, 1, 2, 2, true, 2, 4, 1>(a, s) : launch_k<1, 1, 2, 2, false, 2, 4, 1>(a, s);  // 64x64, two stages
        case 31: return a.src1 ? launch_k<1, 2, 2, 2, true, 1, 4, 1>(a, s) : launch_k<1, 2, 2, 2, false, 1, 4, 1>(a, s);  // 64x128, 1 stage
        case 32: return a.src1 ? launch_k<1, 1, 2, 2, true, 1, 4, 1>(a, s) : launch_k<1, 1, 2, 2, false, 1, 4, 1>(a, s);  // 64x64, 1 stage
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer: 3x3 / stride 1 / Cin = 3 / Cout = 32 (K = 27 is too thin for the MFMA tile and the
// layer is bound by its 4*Cout bytes of output per pixel).  One thread = one output pixel x all
// output channels; weights are wave-uniform (scalar loads), accumulation order (u,v,c) like the
// GEMM kernel's k order.
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_first_f32(const ConvArgs p, const float *__restrict__ w)
{
    // per-wave transpose buffer: 64 pixels x (COUT + 4) floats (row stride 9 x 16 B -> conflict-free b128 access)
    constexpr int ROW = COUT + 4;
    __shared__ __attribute__((aligned(16))) float tr[4][64 * ROW];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int mw = blockIdx.x * 256 + wave * 64;  // first pixel of this wave
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
                const float *wr = w + ((u * 3 + v) * 3 + c) * COUT;  // HWIO, wave-uniform address -> scalar loads
#pragma unroll
                for (int n = 0; n < COUT; n += 2)
                    acc2[n / 2] = __builtin_elementwise_fma(f32x2{xv[c], xv[c]}, f32x2{wr[n], wr[n + 1]}, acc2[n / 2]);
            }
        }
    }
    // epilogue into LDS (lane = pixel), then 16-B stores with 8 lanes per pixel: every wave store instruction
    // writes 1 KiB of contiguous NHWC output (the lane-per-pixel store wrote 16 B per 128-B line and cost 2.2x
    // the bytes at the memory side: WRITE_SIZE in profiles/r01_derived.txt)
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
    // same wave wrote and reads: no barrier needed, only the LDS counter (compiler inserts the wait)
    float *dst = static_cast<float *>(p.dst);
    constexpr int CH = COUT / 4;           // 16-B chunks per pixel
    constexpr int PPI = 64 / CH;           // pixels per store instruction
    const int c4 = lane % CH, pl = lane / CH;
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int px = it * PPI + pl;
        const f32x4 o = *reinterpret_cast<const f32x4 *>(t + px * ROW + c4 * 4);
        if (mw + px < p.M) *reinterpret_cast<f32x4 *>(dst + (size_t)(mw + px) * COUT + c4 * 4) = o;
    }
}

hipError_t launch_conv_first_f32(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s)
{
    if (a.Cin != 3 || a.ksize != 3 || a.stride != 1 || a.Cout != 32 || a.residual || a.src1)
        return hipErrorInvalidValue;
    dim3 grid((a.M + 255) / 256), block(256);
    hipLaunchKernelGGL(conv_first_f32<32>, grid, block, 0, s, a, w_hwio_dev);
    return hipGetLastError();
}

}  // namespace y3
