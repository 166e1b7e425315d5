// Image input stage on the GPU (SURVEY.md 8f "n2"): uint8 HWC image -> float32 in [0,1] -> bilinear resize to
// S x S, written straight into slot b of the NHWC batch the conv program reads.
// Replaces, for the image_file / images_dir sources of reference inference.py:157-158,
//   tf.image.decode_image(..., channels=3, dtype=tf.float32)   (uint8 -> float: cast * (1/255), alpha dropped)
//   tf.image.resize(image, (S, S))                              (bilinear, antialias=False, half-pixel centres)
// and, for the tfrecords source (reference core/load_tfrecords.py:46-48),
//   tf.image.resize(decode_jpeg(...), (S, S)) / 255            (uint8 values resized as floats, then a true divide)
// which is mode 2 below.
// Arithmetic restated from TF's ResizeBilinear CPU kernel: in = (out + 0.5) * (in_size / out_size) - 0.5,
// lower = max(floor(in), 0), upper = min(ceil(in), in_size - 1), lerp = in - floor(in); interpolate along x first
// (top, bottom) then along y; fp32, no contraction.  HBM-bound and tiny; one thread per output pixel.
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

template <typename T>
__device__ __forceinline__ float px(const T *p);
template <>
__device__ __forceinline__ float px<unsigned char>(const unsigned char *p) { return (float)(*p) * (1.0f / 255.0f); }
template <>
__device__ __forceinline__ float px<float>(const float *p) { return *p; }

struct RawU8 { unsigned char v; };   // uint8 taken as 0..255, divided by 255 after the resize (mode 2)
template <>
__device__ __forceinline__ float px<RawU8>(const RawU8 *p) { return (float)p->v; }

template <typename T>
__global__ __launch_bounds__(256) void resize_kernel(const T *__restrict__ src, int H, int W, int pix_stride,
                                                     float *__restrict__ dst, int S)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int oy = i / S, ox = i - oy * S;
    const float sy = (float)H / (float)S, sx = (float)W / (float)S;
    const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), H - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), W - 1);
    const float ly = fy - fly, lx = fx - flx;
    const T *r0 = src + (size_t)y0 * W * pix_stride, *r1 = src + (size_t)y1 * W * pix_stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tl = px<T>(r0 + x0 * pix_stride + c), tr = px<T>(r0 + x1 * pix_stride + c);
        const float bl = px<T>(r1 + x0 * pix_stride + c), br = px<T>(r1 + x1 * pix_stride + c);
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        const float v = top + (bot - top) * ly;
        dst[(size_t)i * 3 + c] = std::is_same<T, RawU8>::value ? v / 255.0f : v;
    }
}

hipError_t launch_resize(const void *src, int is_u8, int H, int W, int pix_stride, float *dst, int S, hipStream_t s)
{
    dim3 grid((S * S + 255) / 256), block(256);
    if (is_u8 == 2)
        hipLaunchKernelGGL(resize_kernel<RawU8>, grid, block, 0, s, static_cast<const RawU8 *>(src), H, W, pix_stride, dst, S);
    else if (is_u8)
        hipLaunchKernelGGL(resize_kernel<unsigned char>, grid, block, 0, s, static_cast<const unsigned char *>(src), H, W,
                           pix_stride, dst, S);
    else
        hipLaunchKernelGGL(resize_kernel<float>, grid, block, 0, s, static_cast<const float *>(src), H, W, pix_stride, dst, S);
    return hipGetLastError();
}

}  // namespace y3
