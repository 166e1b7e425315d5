// bf16 3x3 / stride-1 convolution with TAP-ROW REUSE of the activation tile (round 4).
//
// Same fused op as conv_bf16_mfma (reference: core/parse_model.py:27-52, :155-156), same block tiles, same LDS-DMA double
// buffering, same per-wave epilogue.  What changes is how often the activations travel from L2 to LDS.  The implicit-GEMM A tile
// of tap (u, v) holds, for output pixel m, the input pixel (ho + u - 1, wo + v - 1): for the three taps of one kernel ROW u these
// are the SAME pixels shifted by one -- row m of tap v + 1 is row m + 1 of tap v, except where wo + v - 1 leaves the image row
// (zero padding).  So the kernel fetches, per kernel row u and 64-channel chunk, ONE activation tile of BM + 2 pixel rows (the
// centre-column pixels of output rows m0 - 1 .. m0 + BM) and reads the fragments of tap v from LDS rows shifted by v, zeroing the
// lanes whose pixel sits at the left (v = 0) / right (v = 2) image border with a per-lane mask (the bf16 MFMA co-issues with vector
// work).  Activation fetches: 3 per 9 K tiles instead of 9.
// MEASURED (profiles/r04_ab_bf16_rs.txt, r04_tile_sweep_bf16_rs_*.txt): one layer alone on the chip, the 16-wave tile gains
// 0 .. +10 % over tile 24 of the same shape (52^2: 0 / +2.5 %, 26^2: +1.3 / +3.7 %, 13^2: +5.7 / +9.9 %); in the running pipeline
// (two sub-batch lanes, 128 x 416^2) the conv stack does not move: 9.204 vs 9.186 ms.  NOT selected by the heuristic or the packaged
// tables; the tiles stay selectable (y3_net_set_tile_bf16, Y3_TUNING_FILE) and parity-tested.  The timing-only build that had
// promised more (activations of ONE tap of nine: 9.06 -> 7.80 ms, profiles/r04_ab_bf16_probe_a1.txt) also dropped the other two
// kernel ROWS' fetches, which no reuse removes, and most of the conv's L2 footprint with them: it bounded the wrong thing.
// K order: (kernel row u, 64-channel chunk, kernel column v) -- the same products as conv_bf16_mfma's (tap, channel) order summed
// in another sequence: results agree to fp32-summation rounding before the one rounding to bf16 (not bit-identical); the WHOLE
// family (256x256 / 128x128 / 64x128) uses this order, so a plan's results do not depend on the batch or lane a call runs.
// Needs Cin % 128 == 0 (an even number of (u, chunk) groups: the loop is unrolled over two groups = 6 K tiles so that every LDS
// buffer index is a compile-time constant), CoutPad % BN == 0, bf16 output.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ unsigned pack_bf16_rs(float lo, float hi)
{
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

template <int TM, int TN, int WR, int WC, bool M16>
__global__ __launch_bounds__(64 * WR * WC, 1) void conv_bf16_rs(const ConvArgs p)
{
    constexpr int BK = 64, ROWB = 128, DROWS = 8;   // K tile (bf16), bytes per LDS row, rows per wave DMA instruction
    constexpr int MB = M16 ? 2 * TM : TM, NB = M16 ? 2 * TN : TN;   // accumulator blocks per wave
    using acc_t = typename std::conditional<M16, f32x4, f32x16>::type;
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC, NT = 64 * WR * WC;
    constexpr int RP = NT / 8;                  // rows per load pass
    constexpr int AP = BM / RP, BP = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0 && AP >= 1 && BP >= 1, "tile too small for the thread count");
    constexpr int BMX = BM + DROWS;             // activation rows in LDS: pixels m0 - 1 .. m0 + BM + 6 (BM + 2 are used)
    constexpr int A_BYTES = BMX * ROWB, B_BYTES = BN * ROWB;
    constexpr int KS = M16 ? BK / 32 : BK / 16; // MFMA k steps per K tile
    constexpr int BR = M16 ? 16 : 32;           // rows per accumulator block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const Bbuf = smem;                       // two weight tiles
    unsigned char *const Abuf = smem + 2 * B_BYTES;         // two activation tiles (after the weights: every fragment read then
                                                            // reaches its buffer and block through the 16-bit offset field)

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

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);
    const unsigned OOB0 = p.src0_bytes;

    // ---- activation rows of this thread: LDS row j = pass * RP + lrow  <->  output pixel m0 + j - 1 (its centre-column input pixel) ----
    const int lrow = tid >> 3;
    const int lchunk = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;   // logical 16-B chunk (in bf16) that lands in physical chunk tid & 7
    const int HoWo = p.Ho * p.Wo;
    const float rcpW = 1.0f / (float)p.Wo, rcpH = 1.0f / (float)p.Ho;
    const int shift = m0 == 0 ? 1 : 0;          // origin of the decomposition: pixel m0 - 1 (pixel 0 for the first tile)
    const int mo = m0 - 1 + shift;
    const int r0 = mo % HoWo;
    const int ho0 = r0 / p.Wo;
    const int wo0 = r0 - ho0 * p.Wo;
    // stride 1, "same" padding: the centre pixel of output pixel m is input pixel m, byte offset m * Cin * 2 -- one base per thread, the
    // passes a uniform RP * Cin apart; per pass three validity bits (kernel rows 0..2: the row above / itself / below exists)
    const int abase = ((m0 - 1 + lrow) * p.Cin + lchunk) * 2;
    unsigned arow = 0;
#pragma unroll
    for (int i = 0; i <= AP; ++i) {
        const int j = i * RP + lrow;
        const int m = mo + j - shift;
        const int x = wo0 + j - shift;          // >= 0 for every valid row
        const int xs = x < 0 ? 0 : x;
        const int qx = (int)(((float)xs + 0.5f) * rcpW);
        const int y = ho0 + qx;
        const int qy = (int)(((float)y + 0.5f) * rcpH);
        const int ho = y - qy * p.Ho;
        const bool valid = m >= 0 && m < p.M && (i < AP || lrow < DROWS);
        if (valid) arow |= ((ho > 0 ? 1u : 0u) | 2u | (ho < p.H - 1 ? 4u : 0u)) << (3 * i);
    }
    const bool extra_pass = wave == 0;          // rows BM .. BM + 7: one more DMA instruction, wave 0 only
    const unsigned boff = (unsigned)((n0 + lrow) * p.K + lchunk) * 2u;   // weight row of pass 0; the passes RP rows apart (scalar offset)

    // ---- K walk: (kernel row u, channel chunk c0, kernel column v), v fastest -------------------------------------------------------
    int u = 0, c0 = 0;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch_a = [&](int abuf) {   // the activation tile of (u, c0): BM + 8 rows
        unsigned char *sa = Abuf + abuf * A_BYTES + wave * DROWS * ROWB;
        const int urow = (u - 1) * p.W * p.Cin * 2;
#pragma unroll
        for (int i = 0; i <= AP; ++i) {
            const bool ok = (arow >> (3 * i + u)) & 1u;
            const unsigned vo = ok ? (unsigned)(abase + urow + i * RP * p.Cin * 2) : OOB0;
            if (i < AP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * ROWB), 16, (int)vo, c0 * 2, 0, 0);
            else if (extra_pass) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + AP * RP * ROWB), 16, (int)vo, c0 * 2, 0, 0);
        }
    };
    auto fetch_b = [&](int bbuf, int v) {   // the weight tile of tap (u, v), channels [c0, c0 + 64)
        unsigned char *sb = Bbuf + bbuf * B_BYTES + wave * DROWS * ROWB;
        const int k = ((3 * u + v) * p.Cin + c0) * 2;
#pragma unroll
        for (int j = 0; j < BP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * ROWB), 16, (int)boff, k + j * RP * p.K * 2, 0, 0);
    };
    auto next_group = [&]() {   // (u, c0) -> the next (kernel row, chunk)
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++u;
        }
    };

    acc_t acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < (M16 ? 4 : 16); ++e) acc[i][j][e] = 0.0f;

    // ---- fragment addresses and border masks -------------------------------------------------------------------------------------------
    const int fr = M16 ? (lane & 15) : (lane & 31), fh = M16 ? (lane >> 4) : (lane >> 5);
    // k step 0 only: step s reads chunk (SSTEP * s + fh) ^ key = chunk(0) ^ (SSTEP * s), i.e. the step-0 address with (SSTEP * 16 * s)
    // XOR-ed in -- one vector instruction per step in the loop (free beside the bf16 MFMAs) instead of a register per step
    constexpr int SSTEP = M16 ? 4 : 2;
    int a_addr[3];      // tap column v: LDS row (wave's first row + fr + v), chunk swizzled by that row's key
#pragma unroll
    for (int v = 0; v < 3; ++v) a_addr[v] = 2 * B_BYTES + (wr * 32 * TM + fr + v) * ROWB + ((fh ^ (((fr + v) >> 1) & 7)) << 4);
    const int b_addr = (wc * 32 * TN + fr) * ROWB + ((fh ^ ((fr >> 1) & 7)) << 4);
    // output pixel of the lane's row in block i: m0 + wr * 32 * TM + i * BR + fr; its column decides which taps fall off the image row
    bool keepL[MB], keepR[MB];   // block i's pixel has a left neighbour (v = 0 stays) / a right neighbour (v = 2 stays): lane masks in SGPR pairs
    {
        const int b0m = m0 / HoWo;
        const int r0m = m0 - b0m * HoWo;
        const int wo0m = r0m - (r0m / p.Wo) * p.Wo;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int x = wo0m + wr * 32 * TM + i * BR + fr;
            const int qx = (int)(((float)x + 0.5f) * rcpW);
            const int wo = x - qx * p.Wo;
            keepL[i] = wo > 0;
            keepR[i] = wo < p.Wo - 1;
        }
    }

    auto mma = [&](auto v_tag, auto abuf_tag, auto bbuf_tag) {
        constexpr int V = decltype(v_tag)::value, AB = decltype(abuf_tag)::value, BB = decltype(bbuf_tag)::value;
        const unsigned char *sa = smem + AB * A_BYTES;
        const unsigned char *sb = smem + BB * B_BYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 fa[MB], fb[NB];
            int ao = a_addr[V], bo = b_addr;
            if (s > 0) {   // volatile: computed here, not hoisted into (and spilled from) loop-invariant registers
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ao) : "v"(a_addr[V]), "n"(SSTEP * 16 * s));
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(bo) : "v"(b_addr), "n"(SSTEP * 16 * s));
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                u32x4 w = *reinterpret_cast<const u32x4 *>(sa + ao + i * BR * ROWB);
                if (V != 1) {   // four v_cndmask_b32 on the lane mask
                    const bool k = V == 0 ? keepL[i] : keepR[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = k ? w[e] : 0u;
                }
                fa[i] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const bf16x8 *>(sb + bo + j * BR * ROWB);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // K tiles in pairs of groups: positions 0..5 = (group 2 gg, v 0..2), (group 2 gg + 1, v 0..2); activation buffer = group parity,
    // weight buffer = position parity -- all compile-time.  Each position requests the operands of the next one first.
    const int pairs = (3 * p.Cin / BK) / 2;      // (u, chunk) groups / 2
    fetch_a(0);
    fetch_b(0, 0);
    sync();
    for (int gg = 0; gg < pairs; ++gg) {
        const bool more = gg + 1 < pairs;
        fetch_b(1, 1);
        mma(I0{}, I0{}, I0{});
        sync();
        fetch_b(0, 2);
        mma(I1{}, I0{}, I1{});
        sync();
        next_group();               // the second group of the pair: its activation tile and its first weight tile
        fetch_a(1);
        fetch_b(1, 0);
        mma(I2{}, I0{}, I0{});
        sync();
        fetch_b(0, 1);
        mma(I0{}, I1{}, I1{});
        sync();
        fetch_b(1, 2);
        mma(I1{}, I1{}, I0{});
        sync();
        if (more) {
            next_group();
            fetch_a(0);
            fetch_b(0, 0);
        }
        mma(I2{}, I1{}, I1{});
        sync();
    }

    // ---- per-wave epilogue (conv_bf16_mfma's): transpose through a private LDS scratch, shortcut in fp32, one rounding, 16-B stores ----
    constexpr int CW = 32 * TN, PPRW = CW / 8, NPL = 32 * PPRW / 64;
    float *S = reinterpret_cast<float *>(smem) + wave * (32 * CW);
    unsigned short *dstb = static_cast<unsigned short *>(p.dst);
    const unsigned short *res = static_cast<const unsigned short *>(p.residual);
    const int nw = n0 + wc * CW;
    float sc[NB], sh[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        sc[j] = p.scale[nw + j * BR + fr];
        sh[j] = p.shift[nw + j * BR + fr];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mw = m0 + (wr * TM + i) * 32;
        u32x4 rr[NPL];
        if (res) {
#pragma unroll
            for (int it = 0; it < NPL; ++it) {
                const int q = lane + it * 64;
                const int r = q / PPRW, pc = q - r * PPRW;
                rr[it] = (mw + r < p.M) ? *reinterpret_cast<const u32x4 *>(res + (size_t)(mw + r) * p.Cout + nw + pc * 8) : u32x4{0u, 0u, 0u, 0u};
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
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] * sc[j] + sh[j];
                    if (p.leaky) v = fmaxf(v, 0.1f * v);
                    S[(4 * fh + (e & 3) + 8 * (e >> 2)) * CW + j * 32 + fr] = v;
                }
        }
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
            for (int k = 0; k < 4; ++k) out[k] = pack_bf16_rs(v[2 * k], v[2 * k + 1]);
            if (mw + r < p.M) *reinterpret_cast<u32x4 *>(dstb + (size_t)(mw + r) * p.Cout + nw + pc * 8) = out;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int TM, int TN, int WR, int WC, bool M16>
hipError_t launch_rs(const ConvArgs &a, hipStream_t s)
{
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    if (a.CoutPad % BN) return hipErrorInvalidValue;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.CoutPad / BN;
    const size_t stages = 2 * (size_t)(BM + 8 + BN) * 128;
    const size_t scratch = (size_t)WR * WC * 32 * 32 * TN * sizeof(float);
    const size_t lds = stages > scratch ? stages : scratch;
    auto k = conv_bf16_rs<TM, TN, WR, WC, M16>;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN), dim3(64 * WR * WC), lds, s, a);
    return hipGetLastError();
}
}  // namespace

bool conv_bf16_rs_fits(const ConvArgs &a)
{
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.src1 && a.Cin % 128 == 0 && a.H == a.Ho && a.W == a.Wo && a.K == 9 * a.Cin &&
           a.Cout == a.CoutPad && a.dst != nullptr;
}

hipError_t launch_conv_bf16_rs(const ConvArgs &a, int tile, hipStream_t s)
{
    if (!conv_bf16_rs_fits(a)) return hipErrorInvalidValue;
    switch (tile) {
        case 33: return launch_rs<2, 2, 4, 4, true>(a, s);    // 256x256, 16 waves, 16x16x32 (the shape of tile 24)
        case 34: return launch_rs<2, 2, 2, 2, true>(a, s);    // 128x128, 4 waves, 16x16x32 (tile 27)
        case 35: return launch_rs<1, 2, 2, 2, true>(a, s);    // 64x128, 4 waves, 16x16x32 (tile 29)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace y3
