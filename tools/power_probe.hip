// Sustained-MFMA power probe: how fast can the matrix pipe run for SECONDS, on operands that toggle like real data?
// A loop of independent MFMAs cycling through 8 register-resident operand sets (all zeros / a fixed small pattern /
// random), fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16), 4 waves per SIMD, no memory traffic in
// the loop.  Reports TFLOP/s over ~3 s per variant and the shader clock the kernel itself saw
// (s_memtime ticks / s_memrealtime ticks x 100 MHz).  Study tool; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF16>
__global__ __launch_bounds__(256) void probe(float *out, const unsigned *in, int iters, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    // 8 operand sets per lane: fp32 uses 1 dword per operand, bf16 4 dwords
    unsigned a[8][4], b[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            a[s][w] = in[((s * 4 + w) * 2 + 0) * 256 + tid];
            b[s][w] = in[((s * 4 + w) * 2 + 1) * 256 + tid];
        }
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (BF16) {
                    bf16x8 av, bv;
                    __builtin_memcpy(&av, a[(s + k) & 7], 16);
                    __builtin_memcpy(&bv, b[s], 16);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[k], 0, 0, 0);
                } else {
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[(s + k) & 7][0]), __uint_as_float(b[s][0]),
                                                                  acc[k], 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

template <bool BF16>
static void run(const char *name, const std::vector<unsigned> &host, float *out, unsigned *in, unsigned long long *clk,
                double seconds)
{
    hipMemcpy(in, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * 4, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<BF16>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<BF16>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms1;
    hipEventElapsedTime(&ms1, e0, e1);
    int n = (int)(seconds * 1000 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < n; ++r) hipLaunchKernelGGL(probe<BF16>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop_per_mfma = BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double flops = (double)n * blocks * 4 * (double)iters * 32 * flop_per_mfma;
    printf("%-34s first launch %7.1f TF/s | sustained %5.1f s: %7.1f TF/s, in-kernel shader clock %6.0f MHz\n", name,
           (double)blocks * 4 * iters * 32 * flop_per_mfma / ms1 / 1e9, ms / 1e3, flops / ms / 1e9,
           (double)h[0] / (double)h[1] * 100.0);
    fflush(stdout);
}

static unsigned bf16_bits(float f)
{
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    float *out;
    unsigned *in;
    unsigned long long *clk;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&in, 8 * 4 * 2 * 256 * 4);
    hipMalloc(&clk, 16);
    const size_t n = 8 * 4 * 2 * 256;
    std::vector<unsigned> zeros(n, 0u), patt(n), rnd32(n), rnd16(n);
    srand(7);
    for (size_t i = 0; i < n; ++i) {
        float p = (float)(i & 7) * 0.125f, r = (float)rand() / RAND_MAX * 2.f - 1.f, r2 = (float)rand() / RAND_MAX * 2.f - 1.f;
        __builtin_memcpy(&patt[i], &p, 4);
        __builtin_memcpy(&rnd32[i], &r, 4);
        rnd16[i] = bf16_bits(r) | (bf16_bits(r2) << 16);
    }
    run<false>("fp32 MFMA, zeros", zeros, out, in, clk, seconds);
    run<false>("fp32 MFMA, 8-value pattern", patt, out, in, clk, seconds);
    run<false>("fp32 MFMA, random in [-1,1]", rnd32, out, in, clk, seconds);
    run<true>("bf16 MFMA, zeros", zeros, out, in, clk, seconds);
    run<true>("bf16 MFMA, random in [-1,1]", rnd16, out, in, clk, seconds);
    return 0;
}
