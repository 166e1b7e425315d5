// bf16 convolution on a 256 x 256 block tile with FOUR waves of 128 x 128 (one wave per SIMD), software-pipelined by hand (round 4).
//
// Same fused op as conv_bf16_mfma (reference: core/parse_model.py:27-52, :155-156): implicit-GEMM conv + folded BN / bias + LeakyReLU(0.1)
// + optional shortcut, bf16 in / bf16 out, fp32 accumulation, operands by LDS-DMA into XOR-swizzled 128-byte rows, per-wave epilogue.
// Why another shape: the 16-wave tile (64 x 64 per wave) reads 32 B of LDS per SIMD and clock to keep its MFMAs fed -- 128 B/clk per
// CU, the whole LDS port, so the matrix pipe and the LDS port must both run flat out at the same time
// (profiles/r04_ab_bf16_nofetch.txt: with every operand fetch removed that kernel still tops out at ~1050 TFLOP/s).  A 128 x 128
// wave tile needs half of that (8 fragment reads of 1 KiB per 16 MFMAs of 32 cycles = 16 B/clk per SIMD), at the price of 256
// accumulator registers -- one wave per SIMD, so nothing but the wave's own instruction order hides latency: the fragments of k-step
// s + 1 are read while the MFMAs of k-step s issue (two register sets), the LDS-DMA requests of K tile t + 2 are issued between the
// MFMAs of the last k-step of tile t, and there is one barrier per K tile.
// K order: tap-major, 64 channels per K tile, 16 per MFMA (32x32x16) = the order of the 32x32x16 tiles of conv_bf16_mfma (0..19):
// bit-identical to tile 17 (test_bf16_four_wave_tile_bit_identical_to_the_16_wave_tile).
// MEASURED (profiles/r04_tile_sweep_bf16_w4.txt): per K tile 1.23 us against tile 24's 1.31 (and +20 % over the compiler's own schedule of the
// same wave tile), but 21 us instead of 12 us of fixed cost per tile (four waves alone on a CU through prologue, first fetch and a 4-pass
// epilogue): 775-950 against 915-1070 TFLOP/s per layer, behind everywhere but K >= 4608.  Both K loops run at the ~1700 TFLOP/s-equivalent
// the matrix pipes sustain on toggling data (the power wall), so halving the LDS traffic bought 5 %, not 30.  Selectable
// (y3_net_set_tile_bf16 / Y3_TUNING_FILE), not selected.
// Needs: no concat source, bf16 output, Cin % 64 == 0, CoutPad % 256 == 0 and unpadded, an even number of K tiles.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ unsigned pack_bf16_w4(float lo, float hi)
{
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

__global__ __launch_bounds__(256, 1) void conv_bf16_w4(const ConvArgs p)
{
    constexpr int BM = 256, BN = 256, BK = 64, NT = 256, ROWB = 128, DROWS = 8;
    constexpr int RP = NT / 8;                 // 32 rows per load pass
    constexpr int AP = BM / RP, BP = BN / RP;  // 8 passes each
    constexpr int STAGE_B = (BM + BN) * ROWB;  // 64 KB
    constexpr int MB = 4, NB = 4, KS = 4;      // 32-row blocks per wave, k-steps of 16 per K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tilesN = p.CoutPad / BN;
    const int mt = logical / tilesN, nt = logical - mt * tilesN;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);
    const unsigned OOB0 = p.src0_bytes;

    // ---- gather state: this thread stages row (tid >> 3) + 32 * pass of both operands, physical 16-B chunk tid & 7 -------------------
    const int lrow = tid >> 3;
    const int lchunk = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;
    int aoff[AP], ahw[AP];
    {
        const int HoWo = p.Ho * p.Wo;
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
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            aoff[i] = ((b * p.H + hi0) * p.W + wi0) * p.Cin;
            ahw[i] = (m < p.M) ? ((hi0 << 16) | (wi0 & 0xffff)) : (int)0x80000000;
        }
    }
    const unsigned boff = (unsigned)((n0 + lrow) * p.K + lchunk) * 2u;   // weight row of pass 0; pass j: + j * RP rows (scalar offset)
    const int bstep = RP * p.K * 2;

    int tap = 0, c0 = 0, kglob = 0;
    unsigned avoff[AP];
    auto set_tap = [&]() {
        const int u = tap / p.ksize, v = tap - u * p.ksize;
        const int toff = (u * p.W + v) * p.Cin + lchunk;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int hi = (ahw[i] >> 16) + u, wi = (int)(short)(ahw[i] & 0xffff) + v;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            avoff[i] = ok ? (unsigned)(aoff[i] + toff) * 2u : OOB0;
        }
    };
    set_tap();
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch = [&](int stage) {   // 16 LDS-DMA requests of this wave: rows [32 pass + 8 wave, + 8) of the next K tile's operands
        unsigned char *sa = smem + stage * STAGE_B + wave * DROWS * ROWB;
        unsigned char *sb = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AP; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * ROWB), 16, (int)avoff[i], c0 * 2, 0, 0);
#pragma unroll
        for (int j = 0; j < BP; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * ROWB), 16, (int)boff, kglob * 2 + j * bstep, 0, 0);
    };
    auto advance = [&]() {   // the K walk's state -> the next K tile (kept out of the block that interleaves requests and MFMAs: it branches)
        kglob += BK;
        c0 += BK;
        if (c0 == p.Cin) {
            c0 = 0;
            ++tap;
            set_tap();
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- fragments: lane (fr, fh) reads row fr of each 32-row block, the 16-B chunk (2 s + fh) of k-step s, swizzled by the row -------
    const int fr = lane & 31, fh = lane >> 5;
    const int key = (fr >> 1) & 7;
    const int a_base = (wr * 128 + fr) * ROWB;
    const int b_base = BM * ROWB + (wc * 128 + fr) * ROWB;
    int foff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) foff[s] = ((2 * s + fh) ^ key) << 4;
    bf16x8 fa[2][MB], fb[2][NB];
    auto rd = [&](auto stage_tag, auto s_tag, auto f_tag) {
        constexpr int ST = decltype(stage_tag)::value, S = decltype(s_tag)::value, F = decltype(f_tag)::value;
        const unsigned char *sa = smem + ST * STAGE_B + a_base + foff[S];
        const unsigned char *sb = smem + ST * STAGE_B + b_base + foff[S];
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[F][i] = *reinterpret_cast<const bf16x8 *>(sa + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[F][j] = *reinterpret_cast<const bf16x8 *>(sb + j * 32 * ROWB);
    };
    auto mm = [&](auto f_tag) {
        constexpr int F = decltype(f_tag)::value;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[F][i], fb[F][j], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // 16 MFMAs with 8 fragment reads between them (and, at a tile boundary, the 16 DMA requests): two MFMAs, one read (, two requests)
    auto interleave = [&](bool with_dma) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read (LDS-DMA)
        }
    };

    const int KT = p.K / BK;   // even
    // prologue: tiles 0 and 1 requested, tile 0 landed, its first fragments read
    fetch(0);
    advance();
    fetch(1);
    advance();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // requests retire in order: the 16 of tile 0 are in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd(I0{}, I0{}, I0{});

    // One K tile (operands in stage ST, k-step 0 already in fragment set 0).  `more`: a tile t + 1 exists (in stage ST ^ 1); `more2`: a tile
    // t + 2 exists and is requested into stage ST once every wave has read its last fragments of tile t.
    auto tile = [&](auto stage_tag, auto more_tag, auto more2_tag) {
        constexpr int ST = decltype(stage_tag)::value;
        constexpr bool more = decltype(more_tag)::value, more2 = decltype(more2_tag)::value;
        using STt = std::integral_constant<int, ST>;
        using SNt = std::integral_constant<int, ST ^ 1>;
        rd(STt{}, I1{}, I1{});
        mm(I0{});
        interleave(false);
        rd(STt{}, I2{}, I0{});
        mm(I1{});
        interleave(false);
        rd(STt{}, I3{}, I1{});
        mm(I0{});
        interleave(false);
        // tile boundary: this wave's reads of stage ST are done (lgkmcnt), its requests for tile t + 1 -- issued a whole K tile ago -- have
        // landed (vmcnt), and after the barrier that holds for every wave: stage ST may be overwritten, stage ST ^ 1 may be read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (more2) fetch(ST);
        if constexpr (more) rd(SNt{}, I0{}, I0{});
        mm(I1{});
        interleave(more2);
        if constexpr (more2) advance();
    };
    using T = std::true_type;
    using F = std::false_type;
    for (int t = 0; t + 2 < KT; t += 2) {
        tile(I0{}, T{}, T{});
        tile(I1{}, T{}, T{});
    }
    tile(I0{}, T{}, F{});
    tile(I1{}, F{}, F{});
    // (the last boundary's barrier has passed: every wave is done with the operand stages)

    // ---- per-wave epilogue (conv_bf16_mfma's): 32-row blocks through a private LDS scratch, shortcut in fp32, one rounding, 16-B stores ----
    constexpr int CW = 128, PPRW = CW / 8, NPL = 32 * PPRW / 64;   // 8 pieces of 8 channels per lane and block
    float *S = reinterpret_cast<float *>(smem) + wave * (32 * CW);
    unsigned short *dstb = static_cast<unsigned short *>(p.dst);
    const unsigned short *res = static_cast<const unsigned short *>(p.residual);
    const int nw = n0 + wc * CW;
    float sc[NB], sh[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        sc[j] = p.scale[nw + j * 32 + fr];
        sh[j] = p.shift[nw + j * 32 + fr];
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mw = m0 + wr * 128 + i * 32;
        u32x4 rr[NPL];
        if (res) {
#pragma unroll
            for (int it = 0; it < NPL; ++it) {
                const int q = lane + it * 64;
                const int r = q / PPRW, pc = q - r * PPRW;
                rr[it] = (mw + r < p.M) ? *reinterpret_cast<const u32x4 *>(res + (size_t)(mw + r) * p.Cout + nw + pc * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] * sc[j] + sh[j];
                if (p.leaky) v = fmaxf(v, 0.1f * v);
                S[(4 * fh + (e & 3) + 8 * (e >> 2)) * CW + j * 32 + fr] = v;
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
            for (int k = 0; k < 4; ++k) out[k] = pack_bf16_w4(v[2 * k], v[2 * k + 1]);
            if (mw + r < p.M) *reinterpret_cast<u32x4 *>(dstb + (size_t)(mw + r) * p.Cout + nw + pc * 8) = out;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
}  // namespace

bool conv_bf16_w4_fits(const ConvArgs &a)
{
    return !a.src1 && a.Cin % 64 == 0 && a.CoutPad % 256 == 0 && a.Cout == a.CoutPad && a.K % 128 == 0 && a.K == a.ksize * a.ksize * a.Cin &&
           a.dst != nullptr && a.dec.boxes == nullptr;
}

hipError_t launch_conv_bf16_w4(const ConvArgs &a, hipStream_t s)
{
    if (!conv_bf16_w4_fits(a)) return hipErrorInvalidValue;
    const int tilesM = (a.M + 255) / 256, tilesN = a.CoutPad / 256;
    const size_t lds = 2 * (size_t)(256 + 256) * 128;   // two operand stages (the epilogue's 4 x 16 KB of scratch fit inside)
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_bf16_w4), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_bf16_w4, dim3(tilesM * tilesN), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace y3
