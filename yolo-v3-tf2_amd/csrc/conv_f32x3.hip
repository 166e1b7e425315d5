// fp32-accurate convolution on the bf16 matrix cores: every fp32 value is carried as THREE bf16 planes
// (hi, mid, lo: x = hi + mid + lo exactly, 3 x 8 significand bits = the 24 bits of fp32) and every product
// a*b is formed from the six leading partial products
//      hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid          (dropped terms are <= 2^-24 |a*b|)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 x bf16 products are exact in fp32, so the result has
// fp32-level accuracy (the parity bar of the fp32 path -- 1e-4 on boxes/scores against the fp32 oracle -- is the
// bar of this path, same tests) at 1/6 of the bf16 MFMA rate = 2.67x the rate of v_mfma_f32_32x32x2_f32.
//
// A second scheme of the same kernel (NPL = 2, "f32x2") carries a value as TWO fp16 planes, x = h + l' * 2^-11 with
// h = fp16(x) and l' = fp16((x - h) * 2^11) (the power-of-two scale keeps l' a normal number whenever |x| > 2^-13),
// and forms a*b from three products on v_mfma_f32_32x32x16_f16:
//      ha*hb -> accumulator 0;   ha*lb' + la'*hb -> accumulator 1;   result = acc0 + acc1 * 2^-11
// (dropped: la*lb <= 2^-22 |a*b|; fp16 x fp16 products are exact in fp32).  Half the MFMAs and two thirds of the bytes
// of the three-plane scheme -- it matters because both run at the socket power limit -- for a
// representation error of 2^-22 instead of 2^-24 and a value range of |x| < 65504.
//
// Same fused op as conv_f32.hip / conv_bf16.hip (reference: core/parse_model.py:27-52,72,134,155-156).
// Layout: activations [pixel][plane 0..2][C] bf16, weights [CoutPad][plane][K] bf16, head outputs fp32.
// Operand tiles go HBM/L2 -> LDS by direct-to-LDS buffer loads (one tile per plane), double buffered; LDS rows are
// 2*BK bytes with the 16-B chunk index XOR-swizzled on the source address and on the fragment reads.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

// x -> (hi, mid, lo) bf16 bit patterns with hi + mid + lo == x (fp32 subtractions of nearby values are exact)
__device__ __forceinline__ void split3(float x, unsigned short &hi, unsigned short &mid, unsigned short &lo)
{
    const float h = bf16_round(x);
    const float r1 = x - h;
    const float m = bf16_round(r1);
    const float r2 = r1 - m;
    hi = bf16_bits(h);
    mid = bf16_bits(m);
    lo = bf16_bits(r2);
}

// x -> (h, l') fp16 bit patterns with h + l' * 2^-11 == x up to 2^-22 |x|
__device__ __forceinline__ void split2(float x, unsigned short &hi, unsigned short &lo)
{
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ float f16lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float f16hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

// eight consecutive channels: fp32 -> NPL packed planes (o[plane] = 8 x 16-bit)
template <int NPL>
__device__ __forceinline__ void split_planes(const float (&v)[8], u32x4 (&o)[NPL])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (NPL == 3) {
            unsigned short h0, m0_, l0, h1, m1, l1;
            split3(v[2 * k], h0, m0_, l0);
            split3(v[2 * k + 1], h1, m1, l1);
            o[0][k] = (unsigned)h0 | ((unsigned)h1 << 16);
            o[1][k] = (unsigned)m0_ | ((unsigned)m1 << 16);
            o[NPL - 1][k] = (unsigned)l0 | ((unsigned)l1 << 16);
        } else {
            unsigned short h0, l0, h1, l1;
            split2(v[2 * k], h0, l0);
            split2(v[2 * k + 1], h1, l1);
            o[0][k] = (unsigned)h0 | ((unsigned)h1 << 16);
            o[1][k] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
    }
}

// packed planes of two adjacent channels -> their fp32 values
template <int NPL>
__device__ __forceinline__ void join_planes(const u32x4 (&q)[NPL], int k, float &a0, float &a1)
{
    if (NPL == 3) {
        a0 = (__uint_as_float(q[0][k] << 16) + __uint_as_float(q[1][k] << 16)) + __uint_as_float(q[NPL - 1][k] << 16);
        a1 = (__uint_as_float(q[0][k] & 0xffff0000u) + __uint_as_float(q[1][k] & 0xffff0000u)) +
             __uint_as_float(q[NPL - 1][k] & 0xffff0000u);
    } else {
        a0 = f16lo(q[0][k]) + f16lo(q[1][k]) * (1.0f / 2048.0f);
        a1 = f16hi(q[0][k]) + f16hi(q[1][k]) * (1.0f / 2048.0f);
    }
}

// VAR: schedule variants of the K loop (one kernel body, so that they stay comparable):
//   V_BURST      the LDS-DMA instructions of the next K tile are issued together at the top of the iteration (default)
//   V_ILV        ... issued one or two at a time between the MFMA groups of the current tile (measured: no gain)
//   V_ILV_PINNED ... and pinned there with sched_group_barrier (no gain)
// (the timing-only probes of round 1 -- A for one tap, split K, no fetches, fetches only, one accumulator set -- were removed in
// round 4; their records are profiles/r01_probe_*.txt)
enum { V_BURST = 0, V_ILV = 1, V_ILV_PINNED = 2 };
template <int NPL, int TM, int TN, int WR, int WC, int BK, bool CONCAT, bool OUT_F32, int STAGES = 2, int VAR = V_BURST>
__global__ __launch_bounds__(64 * WR * WC) void conv_f32x3_mfma(const ConvArgs p)
{
    static_assert(NPL == 2 || NPL == 3, "two fp16 planes or three bf16 planes");
    constexpr int NACC = NPL == 2 ? 2 : 1;   // accumulator sets (two-plane scheme: cross terms carry a 2^11 scale)
    constexpr int MPG = NPL == 2 ? 3 : 6;    // MFMAs per (i, j, k-step) group
    constexpr int BM = 32 * TM * WR;
    constexpr int BN = 32 * TN * WC;
    constexpr int NT = 64 * WR * WC;
    constexpr int LPR = BK / 8;        // 16-B chunks (8 bf16) per row: 2, 4 or 8
    constexpr int RPI = 64 / LPR;      // rows written by one wave-wide LDS-DMA instruction
    constexpr int RP = NT / LPR;       // rows per load pass of the whole workgroup
    constexpr int AP = (BM + RP - 1) / RP, BP = (BN + RP - 1) / RP;   // a pass may cover fewer rows than RP: whole
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must be whole wave instructions");   // waves then skip it
    constexpr int ROWB = 2 * BK;                      // LDS row bytes (64 or 128)
    constexpr int PLANE_B = (BM + BN) * ROWB;         // one plane of one stage: A rows then B rows
    constexpr int STAGE_B = NPL * PLANE_B;
    constexpr int CROW = BN + 4;
    // swizzle key (row >> SWZ_SHIFT) & (LPR - 1): 16 consecutive rows then cover all 16 of the 16-B slots of a 256-B bank row
    constexpr int SWZ_SHIFT = (LPR == 8) ? 1 : (LPR == 4) ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    constexpr bool ILV = VAR == V_ILV || VAR == V_ILV_PINNED;
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
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
    const int lchunk = (((tid % LPR) ^ ((lrow >> SWZ_SHIFT) & (LPR - 1))) * 8);  // logical chunk landing in physical chunk tid % LPR
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
            aoff[i] = ((b * H0 + h0) * W0 + w0) * NPL * p.C0;
            aoff1[i] = ((b * p.H + ho) * p.W + wo) * NPL * C1;
            ahw[i] = (m < p.M) ? 0 : (int)0x80000000;
        } else {
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = ((b * p.H + hi0) * p.W + wi0) * NPL * p.Cin;
            ahw[i] = (m < p.M) ? ((hi0 << 16) | (wi0 & 0xffff)) : (int)0x80000000;
        }
    }
    unsigned boff[BP];  // byte offset of plane 0 of weight row n, this lane's chunk
#pragma unroll
    for (int j = 0; j < BP; ++j) boff[j] = (unsigned)((n0 + j * RP + lrow) * NPL * p.K + lchunk) * 2u;

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
            const int toff = (u * p.W + v) * NPL * p.Cin + lchunk;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int hi = (ahw[i] >> 16) + u, wi = (int)(short)(ahw[i] & 0xffff) + v;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                avoff[i] = ok ? (unsigned)(aoff[i] + toff) * 2u : OOB0;
            }
        }
    };
    set_tap();

    int kglob = 0;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch_dma = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            unsigned char *sa = smem + buf * STAGE_B + pl * PLANE_B + wave * RPI * ROWB;
            unsigned char *sb = sa + BM * ROWB;
            if (CONCAT && c0 >= p.C0) {
#pragma unroll
                for (int i = 0; i < AP; ++i)
                    if (i * RP + wave * RPI < BM)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sa + i * RP * ROWB), 16, (int)avoff1[i],
                                                                 (pl * C1 + c0 - p.C0) * 2, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i)
                    if (i * RP + wave * RPI < BM)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * ROWB), 16, (int)avoff[i],
                                                                 (pl * (CONCAT ? p.C0 : p.Cin) + c0) * 2, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < BP; ++j)
                if (j * RP + wave * RPI < BN)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * ROWB), 16, (int)boff[j],
                                                             (pl * p.K + kglob) * 2, 0, 0);
        }
        kglob += BK;
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            if (!CONCAT) set_tap();
        }
    };

    // one LDS-DMA instruction of the next tile (pc = plane * (AP + BP) + row pass); state advances in dma_advance()
    auto dma_piece = [&](int buf, int pc) {
        const int pl = pc / (AP + BP), r = pc - pl * (AP + BP);
        unsigned char *sa = smem + buf * STAGE_B + pl * PLANE_B + wave * RPI * ROWB;
        unsigned char *sb = sa + BM * ROWB;
        if (r < AP) {
            if ((r + 1) * RP <= BM || r * RP + wave * RPI < BM) {
                if (CONCAT && c0 >= p.C0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sa + r * RP * ROWB), 16, (int)avoff1[r],
                                                             (pl * C1 + c0 - p.C0) * 2, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + r * RP * ROWB), 16, (int)avoff[r],
                                                             (pl * (CONCAT ? p.C0 : p.Cin) + c0) * 2, 0, 0);
            }
        } else {
            const int j = r - AP;
            if ((j + 1) * RP <= BN || j * RP + wave * RPI < BN)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * ROWB), 16, (int)boff[j],
                                                         (pl * p.K + kglob) * 2, 0, 0);
        }
    };
    auto dma_advance = [&]() {
        kglob += BK;
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            if (!CONCAT) set_tap();
        }
    };

    // three stages: while tile kt is multiplied, tiles kt+1 and kt+2 are in flight; "tile kt+1 has landed" is then
    // "at most this wave's own DMA instructions of tile kt+2 are outstanding" (vmcnt retires in order)
    int ndma = 0;
    if (STAGES == 3) {
#pragma unroll
        for (int i = 0; i < AP; ++i) ndma += (i * RP + wave * RPI < BM) ? NPL : 0;
#pragma unroll
        for (int j = 0; j < BP; ++j) ndma += (j * RP + wave * RPI < BN) ? NPL : 0;
    }
    auto wait_all_but_newest_tile = [&]() {
        // s_waitcnt takes an immediate: pick the matching one (wave-uniform branches)
        if (ndma >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ndma >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (ndma >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ndma >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ndma >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ndma >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (ndma >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.0f;

    const int KT = p.K / BK;
    fetch_dma(0);
    if (STAGES == 3 && KT > 1) {
        fetch_dma(1);
        wait_all_but_newest_tile();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int a_frag = (wr * 32 * TM + fr) * ROWB;
    const int b_frag = BM * ROWB + (wc * 32 * TN + fr) * ROWB;
    int foff[BK / 16];
#pragma unroll
    for (int s_ = 0; s_ < BK / 16; ++s_) foff[s_] = (((2 * s_ + fh) ^ ((fr >> SWZ_SHIFT) & (LPR - 1))) * 16);

    int cur3 = 0;   // STAGES == 3: stage of tile kt
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = (STAGES == 3) ? cur3 : (STAGES == 2) ? (kt & 1) : 0;
        const bool more = kt + 1 < KT;
        if (STAGES == 2 && !ILV && more) fetch_dma(cur ^ 1);
        if (STAGES == 3 && kt + 2 < KT) fetch_dma(cur3 == 0 ? 2 : cur3 - 1);   // stage of tile kt-1, free since the last barrier
        constexpr int NP = NPL * (AP + BP);             // DMA instructions per tile
        constexpr int NG = (BK / 16) * TM * TN;         // MFMA groups (MPG MFMAs each) per tile
        int grp = 0;
        const unsigned char *st = smem + cur * STAGE_B;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            typedef typename std::conditional<NPL == 3, bf16x8, f16x8>::type frag_t;
            frag_t fa[NPL][TM], fb[NPL][TN];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[pl][i] = *reinterpret_cast<const frag_t *>(st + pl * PLANE_B + a_frag + i * 32 * ROWB + foff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[pl][j] = *reinterpret_cast<const frag_t *>(st + pl * PLANE_B + b_frag + j * 32 * ROWB + foff[s]);
            }
            // products outermost, (i, j) innermost: consecutive MFMAs never wait on each other's accumulator
            if constexpr (NPL == 3) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][i], fb[PB[q]][j], acc[0][i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q)    // h*l' and l'*h into the scaled accumulator set, then h*h
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            f32x16 &c = acc[q == 2 ? 0 : NACC - 1][i][j];
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q == 1 ? 1 : 0][i], fb[q == 0 ? 1 : 0][j], c, 0, 0, 0);
                        }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ILV && STAGES == 2) {
                        // issued on the last tile too (branch-free): it lands in the idle stage, and every address
                        // is range-checked by its buffer descriptor
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            if (pc * NG / NP == s * TM * TN + i * TN + j) dma_piece(cur ^ 1, pc);
                    }
                    ++grp;
                }
        }
        (void)grp;
        if (VAR == V_ILV_PINNED && STAGES == 2) {
            // pin the issue order: a few MFMAs, then one DMA instruction, repeated
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, (NG * MPG) / (NP + 1), 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        if (ILV && STAGES == 2) dma_advance();
        if (STAGES == 3) {
            if (kt + 2 < KT)
                wait_all_but_newest_tile();
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur3 = cur3 == 2 ? 0 : cur3 + 1;
        } else if (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            __syncthreads();                       // every wave is done reading the single stage
            if (kt + 1 < KT) {
                fetch_dma(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }

    // ---- epilogue through LDS, one 32-row block of every wave per pass (as conv_bf16.hip) ------------------
    constexpr int EROWS = WR * 32;
    constexpr int PPR = BN / 8;
    constexpr int NPC = (EROWS * PPR + NT - 1) / NT;
    float *C = reinterpret_cast<float *>(smem);
    unsigned short *dstb = static_cast<unsigned short *>(p.dst);
    const unsigned short *res = static_cast<const unsigned short *>(p.residual);
    const size_t prow = (size_t)NPL * p.Cout;  // 16-bit elements per pixel
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        u32x4 rr[OUT_F32 ? 1 : NPC][NPL];
        if (!OUT_F32 && res) {
#pragma unroll
            for (int it = 0; it < NPC; ++it) {
                const int pc = tid + it * NT;
                const int r = pc / PPR, ch = (pc - r * PPR) * 8;
                const int m = m0 + (r >> 5) * 32 * TM + i * 32 + (r & 31);
                const bool ok = pc < EROWS * PPR && m < p.M && n0 + ch < p.Cout;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    rr[it][pl] = ok ? *reinterpret_cast<const u32x4 *>(res + (size_t)m * prow + pl * p.Cout + n0 + ch)
                                    : u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (i > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wc * TN + j) * 32 + fr;
            const float sc = p.scale[n0 + nl], sh = p.shift[n0 + nl];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // three planes: BN scale applied here; two planes: scale already folded into the packed weights
                float v = NPL == 3 ? acc[0][i][j][e] * sc + sh
                                   : (acc[0][i][j][e] + acc[NACC - 1][i][j][e] * (1.0f / 2048.0f)) + sh;
                if (p.leaky) v = fmaxf(v, 0.1f * v);
                C[(wr * 32 + 4 * fh + (e & 3) + 8 * (e >> 2)) * CROW + nl] = v;
            }
        }
        __syncthreads();
        if (OUT_F32) {
            float *dst = static_cast<float *>(p.dst);
            for (int idx = tid; idx < EROWS * BN; idx += NT) {
                const int r = idx / BN, col = idx - r * BN;
                const int m = m0 + (r >> 5) * 32 * TM + i * 32 + (r & 31), n = n0 + col;
                if (m < p.M && n < p.Cout) dst[(size_t)m * p.Cout + n] = C[r * CROW + col];
            }
        } else {
#pragma unroll
            for (int it = 0; it < NPC; ++it) {
                const int pc = tid + it * NT;
                const int r = pc / PPR, ch = (pc - r * PPR) * 8;
                const int m = m0 + (r >> 5) * 32 * TM + i * 32 + (r & 31);
                if (pc >= EROWS * PPR || m >= p.M || n0 + ch >= p.Cout) continue;   // Cout < BN only for the 32-channel conv
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(C + r * CROW + ch);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(C + r * CROW + ch + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (res) {
                    // shortcut operand rebuilt from its planes, Add([from, x]) = from + x
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float a0, a1;
                        join_planes<NPL>(rr[it], k, a0, a1);
                        v[2 * k] = a0 + v[2 * k];
                        v[2 * k + 1] = a1 + v[2 * k + 1];
                    }
                }
                u32x4 o[NPL];
                split_planes<NPL>(v, o);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *reinterpret_cast<u32x4 *>(dstb + (size_t)m * prow + pl * p.Cout + n0 + ch) = o[pl];
            }
        }
    }
}

// tile table: {BM, BN, waves, BK}
// Round 5: only the tiles a plan can select are instantiated (tuning/f32x3_*.json, f32x2_*.json, choose_tile_x3 / choose_tile_x2 in
// y3_api.cpp; tests/test_abi.py); the schedule variants of rounds 1-2 (BK 16 / 64, interleaved / pinned DMA issue, three stages) are retired.
static const TileInfo kTilesX3[X3_TILE_COUNT] = {
    {128, 128, 4, 32}, {128, 64, 4, 32}, {64, 64, 4, 32}, {64, 128, 4, 32}, {256, 128, 8, 32}, {0, 0, 0, 32},
    {0, 0, 0, 64}, {0, 0, 0, 64}, {128, 256, 8, 32},
    {128, 128, 4, 32}, {0, 0, 0, 32}, {0, 0, 0, 32}, {128, 128, 8, 32}, {128, 64, 4, 32}, {64, 128, 4, 32}, {0, 0, 0, 32},   // 9, 13, 14: single LDS stage
    {0, 0, 0, 16}, {0, 0, 0, 16}, {0, 0, 0, 16}, {0, 0, 0, 16},
    {0, 0, 0, 32}, {0, 0, 0, 32}, {0, 0, 0, 32}, {0, 0, 0, 32},
    {0, 0, 0, 32}, {0, 0, 0, 32},
    {256, 128, 16, 32}, {128, 256, 16, 32},                                       // 26..27: 16 waves, 64x32 wave tiles (two-plane mode)
    {0, 0, 0, 32}, {0, 0, 0, 32},
    {0, 0, 0, 32}, {0, 0, 0, 32}, {0, 0, 0, 32}, {0, 0, 0, 32},
};

bool conv_x3_tile_built(int tile);
TileInfo conv_x3_tile_info(int tile) { return kTilesX3[(tile >= 0 && tile < X3_TILE_COUNT) ? tile : 0]; }

template <int NPL, int TM, int TN, int WR, int WC, int BK, bool CONCAT, bool OUT_F32, int STAGES = 2, int VAR = V_BURST>
static hipError_t launch_kx(const ConvArgs &a, hipStream_t s)
{
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.CoutPad / BN;
    const size_t stages = STAGES * (size_t)NPL * (BM + BN) * (2 * BK);
    const size_t ctile = (size_t)WR * 32 * (BN + 4) * sizeof(float);
    const size_t lds = stages > ctile ? stages : ctile;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = conv_f32x3_mfma<NPL, TM, TN, WR, WC, BK, CONCAT, OUT_F32, STAGES, VAR>;
    static LdsAttrOnce attr;  // per instantiation
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(k), (int)lds, a.device); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN), dim3(64 * WR * WC), lds, s, a);
    return hipGetLastError();
}

template <int NPL, int TM, int TN, int WR, int WC, int BK, int STAGES = 2, int VAR = V_BURST>
static hipError_t launch_tp(const ConvArgs &a, bool out_f32, hipStream_t s)
{
    if (a.src1)
        return out_f32 ? launch_kx<NPL, TM, TN, WR, WC, BK, true, true, STAGES, VAR>(a, s)
                       : launch_kx<NPL, TM, TN, WR, WC, BK, true, false, STAGES, VAR>(a, s);
    return out_f32 ? launch_kx<NPL, TM, TN, WR, WC, BK, false, true, STAGES, VAR>(a, s)
                   : launch_kx<NPL, TM, TN, WR, WC, BK, false, false, STAGES, VAR>(a, s);
}

template <int TM, int TN, int WR, int WC, int BK, int STAGES = 2, int VAR = V_BURST>
static hipError_t launch_tx(const ConvArgs &a, bool out_f32, hipStream_t s)
{
    return launch_tp<3, TM, TN, WR, WC, BK, STAGES, VAR>(a, out_f32, s);
}

hipError_t launch_conv_f32x3(const ConvArgs &a, int tile, bool out_f32, hipStream_t s)
{
    if (tile < 0 || tile >= X3_TILE_COUNT) return hipErrorInvalidValue;
    const TileInfo t = kTilesX3[tile];
    if (t.bm == 0 || !conv_x3_tile_built(tile)) return hipErrorInvalidValue;   // retired id / two-plane only
    if (a.Cin % t.stages || a.CoutPad % t.bn || (a.src1 && a.C0 % t.stages)) return hipErrorInvalidValue;  // .stages holds BK
    switch (tile) {
        case 0: return launch_tx<2, 2, 2, 2, 32>(a, out_f32, s);
        case 1: return launch_tx<2, 1, 2, 2, 32>(a, out_f32, s);
        case 2: return launch_tx<1, 1, 2, 2, 32>(a, out_f32, s);
        case 3: return launch_tx<1, 2, 2, 2, 32>(a, out_f32, s);
        case 4: return launch_tx<2, 2, 4, 2, 32>(a, out_f32, s);
        case 8: return launch_tx<2, 2, 2, 4, 32>(a, out_f32, s);
        case 9: return launch_tx<2, 2, 2, 2, 32, 1>(a, out_f32, s);    // 128x128 w4, single stage (3 workgroups / CU)
        case 12: return launch_tx<2, 1, 2, 4, 32>(a, out_f32, s);      // 128x128 w8 (64x32 wave tile)
        case 13: return launch_tx<2, 1, 2, 2, 32, 1>(a, out_f32, s);   // 128x64 w4, single stage
        case 14: return launch_tx<1, 2, 2, 2, 32, 1>(a, out_f32, s);   // 64x128 w4, single stage
        default: return hipErrorInvalidValue;
    }
}

// two fp16 planes: the same tile ids (a subset: the schedules that won or came close in the three-plane sweeps)
hipError_t launch_conv_f32x2(const ConvArgs &a, int tile, bool out_f32, hipStream_t s)
{
    if (tile < 0 || tile >= X3_TILE_COUNT) return hipErrorInvalidValue;
    const TileInfo t = kTilesX3[tile];
    if (t.bm == 0) return hipErrorInvalidValue;   // retired id
    if (a.Cin % t.stages || a.CoutPad % t.bn || (a.src1 && a.C0 % t.stages)) return hipErrorInvalidValue;
    switch (tile) {
        case 0: return launch_tp<2, 2, 2, 2, 2, 32>(a, out_f32, s);       // 128x128 w4
        case 1: return launch_tp<2, 2, 1, 2, 2, 32>(a, out_f32, s);       // 128x64 w4
        case 2: return launch_tp<2, 1, 1, 2, 2, 32>(a, out_f32, s);       // 64x64 w4
        case 3: return launch_tp<2, 1, 2, 2, 2, 32>(a, out_f32, s);       // 64x128 w4
        case 4: return launch_tp<2, 2, 2, 4, 2, 32>(a, out_f32, s);       // 256x128 w8
        case 8: return launch_tp<2, 2, 2, 2, 4, 32>(a, out_f32, s);       // 128x256 w8
        case 12: return launch_tp<2, 2, 1, 2, 4, 32>(a, out_f32, s);      // 128x128 w8
        case 26: return launch_tp<2, 2, 1, 4, 4, 32>(a, out_f32, s);      // 256x128 w16
        case 27: return launch_tp<2, 2, 1, 2, 8, 32>(a, out_f32, s);      // 128x256 w16
        default: return hipErrorInvalidValue;
    }
}

bool conv_x3_tile_built(int tile)
{
    switch (tile) {
        case 0: case 1: case 2: case 3: case 4: case 8: case 9: case 12: case 13: case 14: return true;
        default: return false;
    }
}

bool conv_x2_tile_built(int tile)
{
    switch (tile) {
        case 0: case 1: case 2: case 3: case 4: case 8: case 12: case 26: case 27: return true;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------
// First layer: fp32 image in, fp32 arithmetic (K = 27), plane-split output (NPL = 3: bf16 x 3, NPL = 2: fp16 x 2).
// ---------------------------------------------------------------------------------------------------------
template <int COUT, int NPL>
__global__ __launch_bounds__(256) void conv_first_f32x3(const ConvArgs p, const float *__restrict__ w)
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
    f32x2 acc2[COUT / 2];
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
    constexpr int CH = COUT / 8;
    constexpr int PPI = 64 / CH;
    const int c8 = lane % CH, pl_ = lane / CH;
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int px = it * PPI + pl_;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(t + px * ROW + c8 * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(t + px * ROW + c8 * 8 + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        u32x4 o[NPL];
        split_planes<NPL>(v, o);
        if (mw + px < p.M) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                *reinterpret_cast<u32x4 *>(dst + (size_t)(mw + px) * NPL * COUT + pl * COUT + c8 * 8) = o[pl];
        }
    }
}

hipError_t launch_conv_first_f32x3(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s)
{
    if (a.Cin != 3 || a.ksize != 3 || a.stride != 1 || a.Cout != 32 || a.residual || a.src1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_first_f32x3<32, 3>), dim3((a.M + 255) / 256), dim3(256), 0, s, a, w_hwio_dev);
    return hipGetLastError();
}

hipError_t launch_conv_first_f32x2(const ConvArgs &a, const float *w_hwio_dev, hipStream_t s)
{
    if (a.Cin != 3 || a.ksize != 3 || a.stride != 1 || a.Cout != 32 || a.residual || a.src1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_first_f32x3<32, 2>), dim3((a.M + 255) / 256), dim3(256), 0, s, a, w_hwio_dev);
    return hipGetLastError();
}

// two-plane -> fp32 (y3_net_read_tensor)
__global__ __launch_bounds__(256) void x2_to_f32_kernel(const unsigned short *x, float *y, size_t npix, int C)
{
    const size_t n = npix * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t px = i / C;
        const int c = (int)(i - px * C);
        const unsigned short *q = x + px * 2 * C + c;
        y[i] = f16lo(q[0]) + f16lo(q[C]) * (1.0f / 2048.0f);
    }
}

hipError_t launch_x2_to_f32(const void *x, float *y, size_t npix, int C, hipStream_t s)
{
    const size_t n = npix * C;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(x2_to_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s,
                       static_cast<const unsigned short *>(x), y, npix, C);
    return hipGetLastError();
}

// three-plane -> fp32 (y3_net_read_tensor)
__global__ __launch_bounds__(256) void x3_to_f32_kernel(const unsigned short *x, float *y, size_t npix, int C)
{
    const size_t n = npix * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t px = i / C;
        const int c = (int)(i - px * C);
        const unsigned short *q = x + px * 3 * C + c;
        y[i] = (__uint_as_float((unsigned)q[0] << 16) + __uint_as_float((unsigned)q[C] << 16)) +
               __uint_as_float((unsigned)q[2 * C] << 16);
    }
}

hipError_t launch_x3_to_f32(const void *x, float *y, size_t npix, int C, hipStream_t s)
{
    const size_t n = npix * C;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(x3_to_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s,
                       static_cast<const unsigned short *>(x), y, npix, C);
    return hipGetLastError();
}

}  // namespace y3
