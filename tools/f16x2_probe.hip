// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs, and how accurate is an fp32 product formed from two fp16
// planes (x = h + l, products hh + hl + lh, fp32 accumulate)?  Prints the error of a 32x32x(16*T) GEMM against a
// double reference for operands of several magnitudes.  Study tool; not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A [32][K] row-major, B [32][K] (n-major), C [32][32]; one wave
__global__ __launch_bounds__(64) void gemm(const float *A, const float *B, float *C, int K, int mode)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int e = 0; e < 8; ++e) {
            const float a = A[r * K + k0 + h * 8 + e], b = B[r * K + k0 + h * 8 + e];
            ah[e] = (_Float16)a;
            al[e] = (_Float16)(a - (float)ah[e]);
            bh[e] = (_Float16)b;
            bl[e] = (_Float16)(b - (float)bh[e]);
        }
        if (mode >= 1) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        }
        if (mode >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) C[(4 * h + (e & 3) + 8 * (e >> 2)) * 32 + r] = acc[e];
}

int main()
{
    const int K = 1152;
    std::vector<float> A(32 * K), B(32 * K), C(1024);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dC, 4096);
    srand(3);
    const float amag[] = {1.0f, 1.0f, 0.05f, 1e-3f, 30.0f}, bmag[] = {1.0f, 0.03f, 0.03f, 0.03f, 0.03f};
    for (int c = 0; c < 5; ++c) {
        for (auto &v : A) v = amag[c] * ((float)rand() / RAND_MAX * 2.f - 1.f);
        for (auto &v : B) v = bmag[c] * ((float)rand() / RAND_MAX * 2.f - 1.f);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        double scale = 0;
        std::vector<double> ref(1024);
        std::vector<float> f32(1024);
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = 0;
                float sf = 0;
                for (int k = 0; k < K; ++k) {
                    s += (double)A[m * K + k] * (double)B[n * K + k];
                    sf = fmaf(A[m * K + k], B[n * K + k], sf);
                }
                ref[m * 32 + n] = s;
                f32[m * 32 + n] = sf;
                scale += s * s;
            }
        scale = sqrt(scale / 1024);
        double e32 = 0;
        for (int i = 0; i < 1024; ++i) e32 = fmax(e32, fabs(f32[i] - ref[i]));
        printf("|a|<=%g |b|<=%g  rms(C)=%.3g  max|err|/rms: fp32 fma chain %.2e", amag[c], bmag[c], scale, e32 / scale);
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double e = 0;
            for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - ref[i]));
            printf(" | %s %.2e", mode == 0 ? "hh" : mode == 1 ? "hh+hl+lh" : "all four", e / scale);
        }
        printf("\n");
    }
    // subnormal check: a = 2^-20 (fp16 subnormal), b = 1: sum over K=16 of a*b = 16 * 2^-20
    for (auto &v : A) v = ldexpf(1.f, -20);
    for (auto &v : B) v = 1.f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, 16, 0);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    printf("subnormal fp16 input 2^-20 x 1, K=16: got %.6e, expected %.6e\n", C[0], 16 * ldexp(1.0, -20));
    return 0;
}
