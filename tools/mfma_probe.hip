// Micro-benchmark: what keeps v_mfma_f32_32x32x2_f32 from issuing every 64 cycles?
// Variants of a loop of 32 MFMAs per "K tile" with optional LDS reads / barriers / LDS writes / global loads,
// at 1..4 waves per SIMD.  Prints TFLOP/s per variant.  (Study tool; not part of the product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// GLD: 0 none, 1 streaming (HBM) loads to VGPRs, 2 L2-resident loads to VGPRs, 3 L2-resident LDS-DMA loads
template <int NACC, bool LDSR, bool BAR, bool LDSW, int GLD>
__global__ __launch_bounds__(256) void probe(float *out, const float *in, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[192 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 192 * 36; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    const float *base = lds + (lane & 31) * 36 + (lane >> 5) * 4;
    f32x4 fa = *reinterpret_cast<const f32x4 *>(base), fb = *reinterpret_cast<const f32x4 *>(base + 64 * 36);
    f32x4 g[6];
    const f32x4 *gin = reinterpret_cast<const f32x4 *>(in) + (size_t)blockIdx.x * 256 * 8 + tid;
    const f32x4 *gl2 = reinterpret_cast<const f32x4 *>(in) + (size_t)(blockIdx.x & 7) * 256 * 8 + tid;  // 8 x 32 KB hot set
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 1u << 30, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int it = 0; it < iters; ++it) {
        if (GLD == 1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) g[k] = gin[(size_t)k * 256 + (size_t)(it & 63) * 256 * 8 * 64];
        } else if (GLD == 2) {
#pragma unroll
            for (int k = 0; k < 6; ++k) g[k] = gl2[(size_t)k * 256 + (size_t)(it & 7) * 256 * 8 * 8];
        } else if (GLD == 3) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + (k * 4 + wave) * 256),
                                                         16, ((blockIdx.x & 7) * 256 * 8 + tid + k * 256) * 16, (it & 7) * 256 * 8 * 8 * 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a0 = fa, b0 = fb, b1 = fb;
            if (LDSR) {
                a0 = *reinterpret_cast<const f32x4 *>(base + q * 8);
                b0 = *reinterpret_cast<const f32x4 *>(base + 64 * 36 + q * 8);
                b1 = *reinterpret_cast<const f32x4 *>(base + 96 * 36 + q * 8);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0], 0, 0, 0);
                acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[1 % NACC], 0, 0, 0);
            }
        }
        if (BAR) __syncthreads();
        if (LDSW) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                f32x4 v = (GLD == 1 || GLD == 2) ? g[k] : fa;
                *reinterpret_cast<f32x4 *>(lds + ((k * 32 + (tid >> 3)) * 36 + (tid & 7) * 4)) = v;
            }
        } else if (GLD == 1 || GLD == 2) {
#pragma unroll
            for (int k = 0; k < 6; ++k) asm volatile("" ::"v"(g[k]));
        } else if (GLD == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int NACC, bool LDSR, bool BAR, bool LDSW, int GLD>
static void run(const char *name, int blocks, float *out, float *in, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, LDSR, BAR, LDSW, GLD>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<NACC, LDSR, BAR, LDSW, GLD>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-44s blocks=%5d  %8.3f ms  %7.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main()
{
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    size_t in_bytes = (size_t)4096 * 256 * 8 * 16 + (size_t)64 * 256 * 8 * 64 * 16;
    hipMalloc(&in, in_bytes);
    hipMemset(in, 0, in_bytes);
    const int it = 2000;
    for (int wps = 2; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        printf("-- %d wave(s) per SIMD\n", wps);
        run<1, false, false, false, 0>("mfma only, 1 acc (every MFMA waits for the one before)", blocks, out, in, it);
        run<2, false, false, false, 0>("mfma only, 2 acc", blocks, out, in, it);
        run<2, true, true, false, 0>("+ ds_read + barriers", blocks, out, in, it);
        run<2, true, true, true, 0>("+ ds_read + barriers + 6 ds_write_b128", blocks, out, in, it);
        run<2, true, true, false, 1>("+ ds_read + barriers + 6 HBM loads->VGPR", blocks, out, in, it);
        run<2, true, true, false, 2>("+ ds_read + barriers + 6 L2 loads->VGPR", blocks, out, in, it);
        run<2, true, true, true, 2>("+ all, L2 loads staged through VGPR+ds_write", blocks, out, in, it);
        run<2, true, true, false, 3>("+ ds_read + barriers + 6 L2 LDS-DMA loads", blocks, out, in, it);
    }
    return 0;
}
