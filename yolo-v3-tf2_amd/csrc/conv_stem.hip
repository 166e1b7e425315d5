// Fused network stem on the gfx950 matrix cores: conv0 (3x3 / 1, 3 -> 32, BN, LeakyReLU) feeding conv1 (3x3 / 2, 32 -> 64,
// BN, LeakyReLU) in ONE kernel -- the 416^2 x 32 tensor between them (1.42 GB for 64 images, the largest tensor of the
// network) is never written to or read from HBM.
//
// Replaces the first two Conv2D -> BatchNormalization -> LeakyReLU groups of the backbone
// (reference: core/parse_model.py:27-52 applied to config/models/yolov3/backbone.yaml layers 1 and 2; the stride-2 conv
// pads top/left only, reference: core/parse_model.py:34-35).
//
// One persistent workgroup (8 waves) per CU walks output tiles of 8 x 16 conv1 pixels.  Per tile:
//   phase 1  conv0 for the 17 x 33 conv0 pixels the tile's 3x3/2 windows cover, on v_mfma_f32_32x32x2_f32 (K = 27 padded
//            to 28) from a 19 x 35 x 3 image patch in LDS; result (+shift, leaky) goes to an LDS patch [pixel][32 ch].
//            conv0 positions outside the image (row / column -1: conv1's zero padding) are written as 0.
//   phase 2  conv1 as an implicit GEMM 128 pixels x 64 channels x K = 288 read ENTIRELY from LDS: A fragments from the
//            patch, B fragments from the conv1 weights, which stay LDS-resident for the whole kernel.  No global load, no
//            barrier and (all addresses = per-lane base + immediate) no vector ALU instruction inside the K loop.
//   phase 3  (optional) the 1x1 conv that follows conv1 in the backbone (64 -> 32, reference backbone.yaml layer 3): conv1's
//            tile is also written to LDS (the patch region, dead after phase 2) and contracted with the 1x1 weights held in
//            registers on v_mfma_f32_16x16x4_f32 -- the 708 MB read of conv1's output by a separate launch disappears.
// Recompute: 561 conv0 pixels per 512 consumed = 1.10 x the conv0 FLOPs (0.45 % of the network's).
//
// LDS (153,516 B of the CU's 160 KB; one workgroup per CU):
//   patch  17*33 pixels x 128 B; pixel (y, x) at index y*33 + (x >> 1) + (x & 1)*17 (even columns first: the stride-2
//          windows of 16 consecutive output pixels then touch consecutive indices), 16-B chunk c stored at chunk
//          c ^ (((x >> 1) + (y >> 1)) & 7): every ds_read_b128 of phase 2 is bank-conflict free (checked exhaustively over
//          taps, sub-tiles and k chunks with the lane groups of MI355X_MICROARCH.md, LDS section).
//   w1     64 rows x 288 floats, chunk kc of row n stored at (kc & ~7) | ((kc & 7) ^ ((n >> 1) & 7)).
//   img    19 x 35 x 3 floats.
// Summation order per output element: conv0 in the step order of stem::k_of_step (pairs of taps x channels);
// conv1 identical to conv_f32_mfma (tap-major, 8q+t / 8q+4+t).  The order is the same for every pixel of every image.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace stem {
constexpr int TH = 8, TW = 16;            // conv1 output pixels per tile
constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1;   // conv0 patch: 17 x 33
constexpr int IH = PH + 2, IW = PW + 2;   // image patch: 19 x 35
constexpr int C0 = 32, C1 = 64, K1 = 9 * C0;
constexpr int PATCH_F = PH * PW * C0;     // 17952 floats
constexpr int W1_F = C1 * K1;             // 18432 floats
constexpr int IMG_F = IH * IW * 3;        // 1995 floats
constexpr int NT = 512;
constexpr int LDS_BYTES = (PATCH_F + W1_F + IMG_F + 1) * 4;
constexpr int IMG_PER_THREAD = (IMG_F + NT - 1) / NT;   // 4

// (conv0's zero-weight padding operand reads one float past the image patch: the LDS region carries one pad float)
// conv0's K = 27 (+1 zero) as 14 MFMA steps of two operands (lane halves h = 0 / 1).  Operand k = (u*3 + v)*3 + c sits at
// float offset 105 u + 3 v + c from the pixel's patch origin: three contiguous runs of nine.  Steps 0..8 pair run 0 with run
// 1 (offsets s and s + 105), steps 9..13 pair run 2's values i and i + 5 (offsets 210 + i + 5 h; the last step's second
// operand is the zero-weight pad).  Both halves of a step differ by a constant, so a lane needs TWO address registers
// (+ immediates) for all 14 reads instead of one per step.
__host__ __device__ constexpr int k_of_step(int s, int h) { return s < 9 ? (h ? 9 + s : s) : (h ? (s == 13 ? 27 : 14 + s) : 9 + s); }
}  // namespace stem


// shader-clock stamps (measurement launches only: StemArgs.clk_stamps != nullptr)
__device__ __forceinline__ void stem_stamp(unsigned long long *out, int slot)
{
    if (out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        out[slot] = __builtin_amdgcn_s_memtime();
        out[slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

__global__ __launch_bounds__(stem::NT, 1) void conv_stem_f32(const StemArgs p)
{
    stem_stamp(p.clk_stamps, 0);
    using namespace stem;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *w1s = smem + PATCH_F;
    float *imgs = smem + PATCH_F + W1_F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // ---- once per workgroup: conv1 weights into LDS (swizzled), conv0 weight fragments and shifts into registers ----
    for (int g = tid; g < C1 * (K1 / 4); g += NT) {
        const int n = g / (K1 / 4), kc = g - n * (K1 / 4);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(p.w1) + n * K1 + kc * 4);
        const int kp = (kc & ~7) | ((kc & 7) ^ ((n >> 1) & 7));
        *reinterpret_cast<f32x4 *>(w1s + n * K1 + kp * 4) = v;
    }
    float b0[14];            // conv0 B fragments: lane (n = fr, half fh) holds w0[k_of_step(s, fh)][n]
#pragma unroll
    for (int s = 0; s < 14; ++s) b0[s] = p.w0[(fh ? k_of_step(s, 1) : k_of_step(s, 0)) * C0 + fr];
    const float sh0 = p.shift0[fr];
    // phase-2 role of this wave: rows [2*wm, 2*wm + 2) of the tile x channels [32*wn, +32)
    const int wm = wave >> 1, wn = wave & 1;
    const float sh1 = p.shift1[wn * 32 + fr];

    // ---- per-lane address pieces (all loop invariant) ----
    // phase 1, row sub-tiles: this wave computes patch rows y = 2*wave and 2*wave + 1 (wave 0 also row 16), 32 pixels
    // x = 0..31 each (lane row = fr); wave 1 also the column x = 32 (17 pixels, lane row = y).
    // A operand: image patch float (y*IW + x)*3 + offset(step, half): two per-lane registers, steps and rows by immediates.
    const int a1lo = (PATCH_F + W1_F + 2 * wave * IW * 3 + fr * 3 + 105 * fh) * 4;        // steps 0..8:  + s*4
    const int a1hi = (PATCH_F + W1_F + 2 * wave * IW * 3 + fr * 3 + 210 + 5 * fh) * 4;    // steps 9..13: + (s - 9)*4
    // phase 1 write: accumulator element e of lane (fr = channel n, fh): pixel x = xe + 4*fh, xe = (e & 3) + 8*(e >> 2).
    //   index = y*33 + (x >> 1) + (x & 1)*17 = y*33 + 2*fh + [(xe >> 1) + (xe & 1)*17]
    //   chunk = (n >> 2) ^ (((x >> 1) + (y >> 1)) & 7) = (n >> 2) ^ ((ce + 2*fh + wave) & 7), ce = xe >> 1, y >> 1 == wave
    //   (row 16 of wave 0: y >> 1 = 8 == 0 mod 8).  xr[k] = float offset of channel n inside a pixel whose swizzle key is
    //   (k + 2*fh + wave) & 7 -- element e uses xr[ce & 7], a compile-time index.
    int xr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[k] = ((((fr >> 2) ^ ((k + 2 * fh + wave) & 7)) << 2) | (fr & 3)) * 4;
    // phase 2 A operand: lane row fr -> tile pixel (r = 2*wm + (fr >> 4), c = fr & 15); tap (u, v):
    //   y = 2r + u, x = 2c + v; index = 66 r + c + [33 u + (v >> 1) + 17 (v & 1)];  key = (c + r + (u >> 1) + (v >> 1)) & 7
    const int pr = 2 * wm + (fr >> 4), pc = fr & 15;
    int a2[3][4];            // [tap key offset t = (u >> 1) + (v >> 1)][k chunk q]: byte address without the tap's immediate
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            a2[t][q] = ((66 * pr + pc) * C0 + (((2 * q + fh) ^ ((pc + pr + t) & 7)) << 2)) * 4;
    int bw[4];               // phase 2 B operand: row n = 32*wn + fr of w1, k chunk q of a tap (tap * 128 B as immediate)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bw[q] = (PATCH_F + (wn * 32 + fr) * K1 + (((2 * q + fh) ^ (((wn * 32 + fr) >> 1) & 7)) << 2)) * 4;

    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.img), 0, p.img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const int S = p.S, So = p.S >> 1;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const char *lds = reinterpret_cast<const char *>(smem);

    // ---- phase 3 (1x1 conv on conv1's tile), wave-uniform switch ----
    const bool with2 = p.w2 != nullptr;
    // conv1 tile in LDS: [128 pixels][64 channels] fp32, 256 B per pixel, 16-B chunk c at c ^ (pixel & 15).
    // Written from the phase-2 accumulators: pixel = 32 wm + m, m = (e & 3) + 8 (e >> 2) + 4 fh -> key (m & 15) =
    // cm + 4 fh with cm = (e & 3) + 8 ((e >> 2) & 1); xt[j], j = (cm & 3) + 4 (cm >> 3): byte offset of channel n in such a pixel
    int xt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cm = (j & 3) + 8 * (j >> 2);
        xt[j] = ((((wn * 32 + fr) >> 2) ^ ((cm + 4 * fh) & 15)) << 4) + (fr & 3) * 4 + (wm * 32 + 4 * fh) * 256;
    }
    // read as A operand of v_mfma_f32_16x16x4_f32: lane (row = l & 15, q = l >> 4) of wave w holds pixel 16 w + row and, in
    // step s, chunk 4 s + q = k 16 s + 4 q .. + 3 (MFMA j of the step contracts k = 16 s + 4 q + j over the four quarters)
    const int l15 = lane & 15, lq = lane >> 4;
    int a3[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a3[s] = (16 * wave + l15) * 256 + (((4 * s + lq) ^ l15) << 4);
    f32x4 w2f[2][4];     // B fragments: channel n = l15 + 16 t, k = 16 s + 4 q .. + 3
    float sh2[2] = {0.0f, 0.0f};
    if (with2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) w2f[t][s] = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(p.w2) + (l15 + 16 * t) * C1 + 16 * s + 4 * lq);
            sh2[t] = p.shift2[l15 + 16 * t];
        }
    }
    const __amdgpu_buffer_rsrc_t rsd2 = __builtin_amdgcn_make_buffer_rsrc(with2 ? p.dst2 : p.dst, 0, with2 ? p.dst2_bytes : p.dst_bytes, 0x00020000);

    // image patch of a tile: element i of the 19 x 35 x 3 patch <- image (iy0 + i / 105, ix0 + (i % 105) / 3, channel);
    // positions outside the image read as 0 (conv0's 'same' padding) through the buffer range check
    auto img_voff = [&](int tile, unsigned (&vo)[IMG_PER_THREAD]) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
        const int iy0 = 2 * ty * TH - 2, ix0 = 2 * tx * TW - 2;
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i) {
            const int e = tid + i * NT;
            const int iy = e / (IW * 3), rem = e - iy * (IW * 3);
            const int ix = rem / 3, c = rem - ix * 3;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = e < IMG_F && (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S;
            vo[i] = ok ? (unsigned)(((b * S + gy) * S + gx) * 3 + c) * 4u : p.img_bytes;
        }
    };
    float rimg[IMG_PER_THREAD];
    auto img_fetch = [&](const unsigned (&vo)[IMG_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i)
            rimg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, (int)vo[i], 0, 0));
    };
    auto img_stage = [&]() {
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i)
            if (tid + i * NT < IMG_F) imgs[tid + i * NT] = rimg[i];
    };

    // the zero-weight 28th operand of conv0 reads one float past the image patch for the patch's last pixel: it must be
    // a finite number (0 x NaN would poison the sum), so the pad float is cleared once
    if (tid == 0) imgs[IMG_F] = 0.0f;
    int tile = blockIdx.x;
    if (tile < p.n_tiles) {
        unsigned vo[IMG_PER_THREAD];
        img_voff(tile, vo);
        img_fetch(vo);
        img_stage();
    }
    __syncthreads();

    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;

        // ================= phase 1: conv0 into the LDS patch =================
        // one row sub-tile: patch row y (wave-uniform), pixels x = 0..31
        auto conv0_row = [&](auto ytag) {
            constexpr int YD = decltype(ytag)::value;       // row relative to the wave's first row (0, 1, or 16 for wave 0)
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            float a[14];
#pragma unroll
            for (int s = 0; s < 14; ++s)
                a[s] = *reinterpret_cast<const float *>(lds + (s < 9 ? a1lo + s * 4 : a1hi + (s - 9) * 4) + YD * IW * 3 * 4);
#pragma unroll
            for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc, 0, 0, 0);
            const int y = 2 * wave + YD;
            // conv0 positions outside the image = conv1's zero padding: patch row 0 of the first tile row, column 0 of
            // the first tile column
            const bool zrow = (ty == 0) && (y == 0);
            const int base = ((y * PW + 2 * fh) * C0) * 4;
            int wa[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wa[k] = base + xr[k];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xe = (e & 3) + 8 * (e >> 2);
                float v = acc[e] + sh0;
                if (p.leaky0) v = fmaxf(v, 0.1f * v);
                const bool zero = zrow || (tx == 0 && xe == 0 && fh == 0);
                v = zero ? 0.0f : v;
                const int imm = (((xe >> 1) + (xe & 1) * 17) * C0) * 4;
                *reinterpret_cast<float *>(const_cast<char *>(lds) + wa[(xe >> 1) & 7] + imm) = v;
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I16 = std::integral_constant<int, 16>;
        conv0_row(I0{});
        conv0_row(I1{});
        if (wave == 0) conv0_row(I16{});
        if (wave == 1) {
            // column sub-tile: x = 32, lane row = y (17 of 32 rows used)
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const int yl = fr < PH ? fr : PH - 1;
            const int abase = (PATCH_F + W1_F + (yl * IW + 32) * 3) * 4;
            float a[14];
#pragma unroll
            for (int s = 0; s < 14; ++s)
                a[s] = *reinterpret_cast<const float *>(lds + abase + (s < 9 ? (105 * fh + s) * 4 : (210 + 5 * fh + s - 9) * 4));
#pragma unroll
            for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int y = (e & 3) + 8 * (e >> 2) + 4 * fh;        // accumulator row = patch row
                float v = acc[e] + sh0;
                if (p.leaky0) v = fmaxf(v, 0.1f * v);
                v = (ty == 0 && y == 0) ? 0.0f : v;
                // x = 32: index y*33 + 16, key (16 + (y >> 1)) & 7
                const int off = ((y * PW + 16) * C0 + ((((fr >> 2) ^ ((y >> 1) & 7)) << 2) | (fr & 3))) * 4;
                if (y < PH) *reinterpret_cast<float *>(const_cast<char *>(lds) + off) = v;
            }
        }
        __syncthreads();   // patch complete; the image patch is free

        // prefetch the next tile's image patch: in flight during phase 2, staged before the closing barrier
        const int next = tile + gridDim.x;
        if (next < p.n_tiles) {
            unsigned vo[IMG_PER_THREAD];
            img_voff(next, vo);
            img_fetch(vo);
        }

        // ================= phase 2: conv1 from LDS =================
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int t = (u >> 1) + (v >> 1);
                const int aimm = ((33 * u + (v >> 1) + 17 * (v & 1)) * C0) * 4;
                const int bimm = ((u * 3 + v) * C0) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 fa = *reinterpret_cast<const f32x4 *>(lds + a2[t][q] + aimm);
                    const f32x4 fb = *reinterpret_cast<const f32x4 *>(lds + bw[q] + bimm);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k], fb[k], acc, 0, 0, 0);
                }
            }

        // ---- epilogue: accumulator element e = tile pixel row (e & 3) + 8*(e >> 2) + 4*fh of this wave's 32, channel fr ----
        {
            const int oy0 = ty * TH + 2 * wm, ox0 = tx * TW;
            const int n = wn * 32 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2);               // + 4*fh: pixel (row + 4 fh) >> 4, & 15
                const int rr = (row >> 4), cc = (row & 15) + 4 * fh;  // row & 15 + 4 fh < 16: rows 0-3, 8-11 (+4) only
                float v = acc[e] + sh1;
                if (p.leaky1) v = fmaxf(v, 0.1f * v);
                const unsigned off = (unsigned)((((b * So + oy0 + rr) * So) + ox0 + cc) * C1 + n) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, (int)off, 0, 0);
                acc[e] = v;      // kept for phase 3
            }
        }
        if (with2) {
            __syncthreads();   // every wave is done reading the patch: it becomes conv1's tile
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cm = (e & 3) + 8 * ((e >> 2) & 1);
                const int imm = ((e & 3) + 8 * (e >> 2)) * 256;
                *reinterpret_cast<float *>(const_cast<char *>(lds) + xt[(cm & 3) + 4 * (cm >> 3)] + imm) = acc[e];
            }
            __syncthreads();   // tile complete
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            f32x4v c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 fa = *reinterpret_cast<const f32x4 *>(lds + a3[s]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], w2f[0][s][j], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], w2f[1][s][j], c1, 0, 0, 0);
                }
            }
            // C layout of 16x16: column = l & 15 (channel), rows 4 (l >> 4) + i (pixel 16 w + 4 q + i = tile row w, col 4 q + i)
            const int oy = ty * TH + wave, ox = tx * TW + 4 * lq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v0 = c0[i] + sh2[0], v1 = c1[i] + sh2[1];
                if (p.leaky2) {
                    v0 = fmaxf(v0, 0.1f * v0);
                    v1 = fmaxf(v1, 0.1f * v1);
                }
                const unsigned off = (unsigned)(((b * So + oy) * So + ox + i) * 32 + l15) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rsd2, (int)off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rsd2, (int)off, 64, 0);
            }
        }
        if (next < p.n_tiles) img_stage();
        __syncthreads();   // patch free for the next tile; its image patch is in place
    }
    stem_stamp(p.clk_stamps, 2);
}

hipError_t launch_conv_stem_f32(const StemArgs &a, hipStream_t s)
{
    using namespace stem;
    if (a.S % 32 || a.B <= 0 || !a.img || !a.w0 || !a.w1 || !a.dst) return hipErrorInvalidValue;
    StemArgs p = a;
    p.tiles_y = (a.S / 2) / TH;
    p.tiles_x = (a.S / 2) / TW;
    p.n_tiles = a.B * p.tiles_y * p.tiles_x;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_stem_f32), LDS_BYTES, a.device); e != hipSuccess) return e;
    const int cus = a.n_cus > 0 ? a.n_cus : 256;                // read once at plan time (y3_net_plan)
    const int grid = p.n_tiles < cus ? p.n_tiles : cus;         // one persistent workgroup per CU
    hipLaunchKernelGGL(conv_stem_f32, dim3(grid), dim3(NT), LDS_BYTES, s, p);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------------------
// bf16 variant (BASELINE config 5): same tiling and phase structure.  conv0 runs on the BF16 matrix cores from split
// operands (x = hi + lo, products hi*hi + hi*lo + lo*hi with fp32 accumulation: ~2^-16 relative error per product -- close to,
// but NOT, the fp32 arithmetic of the stand-alone bf16 first-layer launch; see phase 1 below), then y = acc*scale + shift, leaky,
// and the result is rounded to bf16 exactly where the two-launch form stores it -- here into the LDS patch; conv1 runs on
// v_mfma_f32_32x32x16_bf16 from LDS (bf16 weights, unscaled; y = acc*scale + shift, leaky, bf16 out).
// LDS: patch 561 x 64 B (chunk c of pixel (y, x) at c ^ ((x >> 2) & 3)), w1 64 x 576 B (chunk c of a tap's four at
// c ^ ((n >> 2) & 3)), image patch 7,980 B: 80,748 B -> TWO workgroups per CU, one computing while the other stores.
// The output tile leaves through LDS (the patch region, dead after phase 2): every store instruction writes whole
// 128-B pixels (64 channels x 2 B), 16 pixels of a row contiguous.
// -----------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

namespace stemb {
using namespace stem;
constexpr int PATCH_B = PH * PW * C0 * 2;     // 35904 bytes
constexpr int W1_B = C1 * K1 * 2;             // 36864 bytes
constexpr int IMG_B = (IMG_F + 1) * 4;        // 7984 bytes (one pad float: conv0's zero-weight operand)
constexpr int LDS_BYTES_B = PATCH_B + W1_B + IMG_B;   // 80752 <= 81920: two workgroups per CU
}  // namespace stemb

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

__global__ __launch_bounds__(stem::NT, 4) void conv_stem_bf16(const StemArgs p)
{
    stem_stamp(p.clk_stamps, 0);
    using namespace stemb;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    char *lds = reinterpret_cast<char *>(smemb);
    float *imgs = reinterpret_cast<float *>(smemb + PATCH_B + W1_B);

    // conv1 weights (bf16 [64][288]) into LDS: 16-B chunk kc (8 channels; 4 chunks per tap) of row n
    {
        const unsigned short *w1 = static_cast<const unsigned short *>(p.w1);
        for (int g = tid; g < C1 * (K1 / 8); g += NT) {
            const int n = g / (K1 / 8), kc = g - n * (K1 / 8);
            const u32x4s v = *reinterpret_cast<const u32x4s *>(w1 + n * K1 + kc * 8);
            const int kp = (kc & ~3) | ((kc & 3) ^ ((n >> 2) & 3));
            *reinterpret_cast<u32x4s *>(lds + PATCH_B + n * (K1 * 2) + kp * 16) = v;
        }
    }
    // conv0 on the bf16 matrix cores with fp32-class accuracy: image values and weights are split x = hi + lo (two bf16, 16
    // significand bits together) and a product is formed as hi*hi + hi*lo + lo*hi with fp32 accumulation (the dropped lo*lo is
    // <= 2^-18 of the product): 6 v_mfma_f32_32x32x16_bf16 (K = 27 padded to 32) instead of 14 v_mfma_f32_32x32x2_f32 -- 192
    // instead of 896 matrix-pipe cycles per 32 x 32 block.  The result is rounded to bf16 into the patch as before.
    // K slots (group g = 2 s + h of MFMA step s, lane half h; 8 slots j each), k = 9 u + i = float offset 105 u + i in the patch:
    //   g0: k = j              g1: k = 9 + j              g2: k = 18 + j              g3: j = 0, 1, 2 -> k = 8, 17, 26; else zero weight
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = s == 0 ? j : 18 + j;                                  // lane half 0
            const int k1 = s == 0 ? 9 + j : (j < 3 ? 8 + 9 * j : -1);            // lane half 1
            const float w0v = p.w0[k0 * C0 + fr];
            const float w1v = k1 >= 0 ? p.w0[k1 * C0 + fr] : 0.0f;
            const float wv = fh ? w1v : w0v;
            const __bf16 hi = (__bf16)wv;
            bh[s][j] = hi;
            bl[s][j] = (__bf16)(wv - (float)hi);
        }
    const float sc0 = p.scale0[fr], sh0 = p.shift0[fr];
    const int wm = wave >> 1, wn = wave & 1;
    const float sc1 = p.scale1[wn * 32 + fr], sh1 = p.shift1[wn * 32 + fr];
    const bool with2 = p.w2 != nullptr;      // wave-uniform: the 1x1 conv after conv1 runs from the staging tile (phase 3)
    const float sc2 = with2 ? p.scale2[fr] : 0.0f, sh2 = with2 ? p.shift2[fr] : 0.0f;
    constexpr int OUT2_OFF = 128 * 128;      // second staging tile, behind the first (both inside the dead patch region)
    static_assert(OUT2_OFF + 128 * 64 <= PATCH_B, "staging tiles must fit the patch region");

    // A operand of a conv0 block: 16 floats of the image patch per lane, relative to the pixel's patch origin o (bytes):
    //   step 0: o + (105 h + j) 4;   step 1, h = 0: o + (210 + j) 4;   h = 1: j = 0, 1, 2 -> o + (8 + 105 j) 4, j >= 3 -> anything finite
    // (zero weight; o + (8 + j) 4 is taken).  Four per-lane displacements + immediates cover all 16 reads.
    const int d0 = 420 * fh, d1 = fh ? 32 : 840, d2 = fh ? 452 - 4 : 840, d3 = fh ? 872 - 8 : 840;
    const int a1o = PATCH_B + W1_B + (2 * wave * IW * 3 + fr * 3) * 4;      // origin of pixel (y = 2 wave, x = fr)
    auto conv0_block = [&](int o) -> f32x16 {
        float xs[16];
        const int v0 = o + d0, v1 = o + d1, v2 = o + d2, v3 = o + d3;
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = *reinterpret_cast<const float *>(lds + v0 + j * 4);
        xs[8] = *reinterpret_cast<const float *>(lds + v1);
        xs[9] = *reinterpret_cast<const float *>(lds + v2 + 4);
        xs[10] = *reinterpret_cast<const float *>(lds + v3 + 8);
#pragma unroll
        for (int j = 3; j < 8; ++j) xs[8 + j] = *reinterpret_cast<const float *>(lds + v1 + j * 4);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            bf16x8 ah, al;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = xs[8 * s_ + j];
                const __bf16 hi = (__bf16)xv;
                ah[j] = hi;
                al[j] = (__bf16)(xv - (float)hi);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s_], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s_], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s_], acc, 0, 0, 0);
        }
        return acc;
    };
    // phase 1 write: element e -> pixel x = xe + 4 fh of row y; channel n = fr: chunk (n >> 3) ^ (((xe >> 2) + fh) & 3)
    int xr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xr[k] = (((fr >> 3) ^ ((k + fh) & 3)) << 4) + (fr & 7) * 2;
    // phase 2 A: pixel (r = 2 wm + (fr >> 4), c = fr & 15), tap (u, v): x = 2c + v, key (x >> 2) & 3 =
    // ((c >> 1) + ((c & 1) & (v >> 1))) & 3 -> two variants (v < 2, v == 2)
    const int pr = 2 * wm + (fr >> 4), pc = fr & 15;
    int a2[2][2];
#pragma unroll
    for (int vk = 0; vk < 2; ++vk)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            a2[vk][s] = (66 * pr + pc) * 64 + (((2 * s + fh) ^ (((pc >> 1) + ((pc & 1) & vk)) & 3)) << 4);
    int bw[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) bw[s] = PATCH_B + (wn * 32 + fr) * (K1 * 2) + (((2 * s + fh) ^ (((wn * 32 + fr) >> 2) & 3)) << 4);

    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.img), 0, p.img_bytes, 0x00020000);
    const int S = p.S, So = p.S >> 1;
    const int tiles_per_img = p.tiles_y * p.tiles_x;

    auto img_voff = [&](int tile, unsigned (&vo)[IMG_PER_THREAD]) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
        const int iy0 = 2 * ty * TH - 2, ix0 = 2 * tx * TW - 2;
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i) {
            const int e = tid + i * NT;
            const int iy = e / (IW * 3), rem = e - iy * (IW * 3);
            const int ix = rem / 3, c = rem - ix * 3;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = e < IMG_F && (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S;
            vo[i] = ok ? (unsigned)(((b * S + gy) * S + gx) * 3 + c) * 4u : p.img_bytes;
        }
    };
    float rimg[IMG_PER_THREAD];
    auto img_fetch = [&](const unsigned (&vo)[IMG_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i)
            rimg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, (int)vo[i], 0, 0));
    };
    auto img_stage = [&]() {
#pragma unroll
        for (int i = 0; i < IMG_PER_THREAD; ++i)
            if (tid + i * NT < IMG_F) imgs[tid + i * NT] = rimg[i];
    };

    // the zero-weight 28th operand of conv0 reads one float past the image patch for the patch's last pixel: it must be
    // a finite number (0 x NaN would poison the sum), so the pad float is cleared once
    if (tid == 0) imgs[IMG_F] = 0.0f;
    int tile = blockIdx.x;
    if (tile < p.n_tiles) {
        unsigned vo[IMG_PER_THREAD];
        img_voff(tile, vo);
        img_fetch(vo);
        img_stage();
    }
    __syncthreads();

    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t2 = tile - b * tiles_per_img;
        const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;

        // ---- phase 1: conv0 (fp32 MFMA) -> bf16 patch ----
        auto conv0_row = [&](auto ytag) {
            constexpr int YD = decltype(ytag)::value;
            const f32x16 acc = conv0_block(a1o + YD * IW * 3 * 4);
            const int y = 2 * wave + YD;
            const bool zrow = (ty == 0) && (y == 0);
            const int base = (y * PW + 2 * fh) * 64;
            int wa[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wa[k] = base + xr[k];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xe = (e & 3) + 8 * (e >> 2);
                float v = acc[e] * sc0 + sh0;
                if (p.leaky0) v = fmaxf(v, 0.1f * v);
                const bool zero = zrow || (tx == 0 && xe == 0 && fh == 0);
                v = zero ? 0.0f : v;
                const int imm = ((xe >> 1) + (xe & 1) * 17) * 64;
                *reinterpret_cast<unsigned short *>(lds + wa[(xe >> 2) & 3] + imm) = bf16_bits(v);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I16 = std::integral_constant<int, 16>;
        conv0_row(I0{});
        conv0_row(I1{});
        if (wave == 0) conv0_row(I16{});
        if (wave == 1) {   // column x = 32: lane row = patch row y
            const int yl = fr < PH ? fr : PH - 1;
            const f32x16 acc = conv0_block(PATCH_B + W1_B + ((yl * IW + 32) * 3) * 4);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int y = (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc[e] * sc0 + sh0;
                if (p.leaky0) v = fmaxf(v, 0.1f * v);
                v = (ty == 0 && y == 0) ? 0.0f : v;
                // x = 32: index y*33 + 16, key (32 >> 2) & 3 = 0
                const int off = (y * PW + 16) * 64 + ((fr >> 3) << 4) + (fr & 7) * 2;
                if (y < PH) *reinterpret_cast<unsigned short *>(lds + off) = bf16_bits(v);
            }
        }
        __syncthreads();   // (1) patch complete, image patch free

        const int next = tile + gridDim.x;
        if (next < p.n_tiles) {
            unsigned vo[IMG_PER_THREAD];
            img_voff(next, vo);
            img_fetch(vo);
        }

        // ---- phase 2: conv1 on the bf16 matrix cores, everything from LDS ----
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int aimm = (33 * u + (v >> 1) + 17 * (v & 1)) * 64;
                const int bimm = (u * 3 + v) * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(lds + a2[v >> 1][s] + aimm);
                    const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(lds + bw[s] + bimm);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                }
            }
        __syncthreads();   // (2) every wave is done reading the patch: it becomes the output staging tile

        // ---- epilogue: [128 pixels][64 channels] bf16 through LDS, then whole 128-B pixels to HBM ----
        {
            const int n = wn * 32 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * fh;      // pixel of this wave's 32
                float v = acc[e] * sc1 + sh1;
                if (p.leaky1) v = fmaxf(v, 0.1f * v);
                *reinterpret_cast<unsigned short *>(lds + (wm * 32 + m) * 128 + n * 2) = bf16_bits(v);
            }
        }
        __syncthreads();   // (3) staging tile complete
        {
            unsigned short *dst = static_cast<unsigned short *>(p.dst);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int g = tid + i * NT;                         // 1024 chunks of 16 B
                const int P = g >> 3, ch = g & 7;
                const u32x4s v = *reinterpret_cast<const u32x4s *>(lds + P * 128 + ch * 16);
                const int oy = ty * TH + (P >> 4), ox = tx * TW + (P & 15);
                *reinterpret_cast<u32x4s *>(dst + ((size_t)(b * So + oy) * So + ox) * C1 + ch * 8) = v;
            }
        }
        if (with2) {
            // ---- phase 3: the 1x1 conv that follows conv1 (64 -> 32), from the staging tile = conv1's output as stored (bf16) ----
            // waves 0..3: one 32-pixel block each, K = 64 = four steps of v_mfma_f32_32x32x16_bf16 in the k grouping of
            // conv_bf16_mfma (lane half h takes k = 16 s + 8 h .. + 7): bit-identical to the separate launch.  Weights
            // ([32][64] bf16, 4 KB, L1-resident) are fetched per tile: the kernel has no registers to keep them (118 of 128).
            if (wave < 4) {
                f32x16 acc2;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
                const unsigned short *w2 = static_cast<const unsigned short *>(p.w2);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(lds + (wave * 32 + fr) * 128 + (2 * s + fh) * 16);
                    const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(w2 + fr * C1 + (2 * s + fh) * 8);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * fh;
                    float v = acc2[e] * sc2 + sh2;
                    if (p.leaky2) v = fmaxf(v, 0.1f * v);
                    *reinterpret_cast<unsigned short *>(lds + OUT2_OFF + (wave * 32 + m) * 64 + fr * 2) = bf16_bits(v);
                }
            }
            __syncthreads();   // (3b) second staging tile ([128 pixels][32 channels] bf16) complete
            {
                unsigned short *dst2 = static_cast<unsigned short *>(p.dst2);
                const int P = tid >> 2, ch = tid & 3;               // 512 chunks of 16 B
                const u32x4s v = *reinterpret_cast<const u32x4s *>(lds + OUT2_OFF + P * 64 + ch * 16);
                const int oy = ty * TH + (P >> 4), ox = tx * TW + (P & 15);
                *reinterpret_cast<u32x4s *>(dst2 + ((size_t)(b * So + oy) * So + ox) * 32 + ch * 8) = v;
            }
        }
        if (next < p.n_tiles) img_stage();
        __syncthreads();   // (4) staging tile read, next image patch in place
    }
    stem_stamp(p.clk_stamps, 2);
}

hipError_t launch_conv_stem_bf16(const StemArgs &a, hipStream_t s)
{
    using namespace stemb;
    if (a.S % 32 || a.B <= 0 || !a.img || !a.w0 || !a.w1 || !a.dst || !a.scale0 || !a.scale1) return hipErrorInvalidValue;
    StemArgs p = a;
    p.tiles_y = (a.S / 2) / TH;
    p.tiles_x = (a.S / 2) / TW;
    p.n_tiles = a.B * p.tiles_y * p.tiles_x;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(conv_stem_bf16), LDS_BYTES_B, a.device); e != hipSuccess) return e;
    const int cus = a.n_cus > 0 ? a.n_cus : 256;                // read once at plan time (y3_net_plan)
    const int grid = p.n_tiles < 2 * cus ? p.n_tiles : 2 * cus;   // two persistent workgroups per CU
    hipLaunchKernelGGL(conv_stem_bf16, dim3(grid), dim3(NT), LDS_BYTES_B, s, p);
    return hipGetLastError();
}

}  // namespace y3
