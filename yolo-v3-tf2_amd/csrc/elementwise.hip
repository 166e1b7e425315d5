// Stand-alone Add / UpSampling2D(2) / Concatenate(axis=3) (reference: core/parse_model.py:155-156,72,134).
// YOLOv3's own graph never launches these (the lowering folds all of them into conv launches); they exist
// so that a model description whose pattern does not fold still runs.  HBM-bound, 16-B accesses.
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void add_kernel(const f32x4 *a, const f32x4 *b, f32x4 *y, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) y[i] = a[i] + b[i];
}

hipError_t launch_add(const float *a, const float *b, float *y, size_t n, hipStream_t s)
{
    if (n & 3) return hipErrorInvalidValue;
    const size_t n4 = n >> 2;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(add_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, (const f32x4 *)a, (const f32x4 *)b,
                       (f32x4 *)y, n4);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const f32x4 *x, int B, int H, int W, int C4, f32x4 *y)
{
    const size_t total = (size_t)B * 2 * H * 2 * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t pix = i / C4;
        const int w2 = (int)(pix % (2 * W));
        pix /= (2 * W);
        const int h2 = (int)(pix % (2 * H));
        const int b = (int)(pix / (2 * H));
        y[i] = x[(((size_t)b * H + (h2 >> 1)) * W + (w2 >> 1)) * C4 + c];
    }
}

hipError_t launch_upsample2x(const float *x, int B, int H, int W, int C, float *y, hipStream_t s)
{
    if (C & 3) return hipErrorInvalidValue;
    const size_t total = (size_t)B * 4 * H * W * (C >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, (const f32x4 *)x, B, H, W,
                       C >> 2, (f32x4 *)y);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void concat_kernel(const f32x4 *a, int Ca4, const f32x4 *b, int Cb4, size_t npix,
                                                     f32x4 *y)
{
    const int C4 = Ca4 + Cb4;
    const size_t total = npix * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const size_t pix = i / C4;
        y[i] = (c < Ca4) ? a[pix * Ca4 + c] : b[pix * Cb4 + (c - Ca4)];
    }
}

hipError_t launch_concat(const float *a, int Ca, const float *b, int Cb, size_t npix, float *y, hipStream_t s)
{
    if ((Ca | Cb) & 3) return hipErrorInvalidValue;
    const size_t total = npix * ((Ca + Cb) >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(concat_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, (const f32x4 *)a, Ca >> 2,
                       (const f32x4 *)b, Cb >> 2, npix, (f32x4 *)y);
    return hipGetLastError();
}

}  // namespace y3
