// yolo_decode (+ the class arg-max / score of yolo_nms) on gfx950.
//
// Replaces reference core/yolo_decode_layer.py:4-36 (split, sigmoid x3, exp*anchor, meshgrid add,
// divide, +-wh/2, concat, reshape, concat) and core/yolo_nms.py:18-24 (argmax, reduce_max, multiply).
// HBM-bound: 4*(5+nc) bytes read per box.  A workgroup copies 128 boxes (contiguous 128*(5+nc)
// floats of one scale) into LDS with 16-B coalesced loads, then each lane owns one box and walks
// its 5+nc values from LDS (row stride 5+nc = 85 dwords is odd -> conflict-free ds_read_b32);
// class probabilities are written back through LDS so that the [B,N,nc] store is coalesced.
// Arithmetic: sigmoid(x) = 1/(1+exp(-x)), true divisions, fp32, no contraction (-ffp-contract=off).
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int DEC_BOXES = 128;  // boxes per workgroup == threads per workgroup

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct DecodeLaunch {
    DecodeArgs a;
    int blk_start[4];  // first workgroup of each scale (prefix sums), [3] = total
};

template <bool WRITE_PROBS, bool WRITE_SCORES>
__global__ __launch_bounds__(DEC_BOXES) void decode_kernel(const DecodeLaunch L, float *__restrict__ bboxes,
                                                           float *__restrict__ conf, float *__restrict__ probs,
                                                           int64_t *__restrict__ cls, float *__restrict__ scores)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int F = 5 + L.a.nc;
    int s = 0;
    if ((int)blockIdx.x >= L.blk_start[1]) s = 1;
    if ((int)blockIdx.x >= L.blk_start[2]) s = 2;
    const int g = L.a.g[s];
    const int per_img = g * g * 3;
    const long long total = (long long)L.a.B * per_img;            // boxes of this scale over the batch
    const long long box0 = (long long)((int)blockIdx.x - L.blk_start[s]) * DEC_BOXES;
    const int nbox = (int)((total - box0) < DEC_BOXES ? (total - box0) : DEC_BOXES);
    const float *src = L.a.grid[s] + box0 * F;
    const int nfl = nbox * F;
    const int tid = threadIdx.x;
    // coalesced copy (16 B per lane; box0*F*4 is a multiple of 16 because DEC_BOXES*4 is)
    const int nvec = nfl >> 2;
    for (int i = tid; i < nvec; i += DEC_BOXES) reinterpret_cast<f32x4 *>(lds)[i] = reinterpret_cast<const f32x4 *>(src)[i];
    for (int i = (nvec << 2) + tid; i < nfl; i += DEC_BOXES) lds[i] = src[i];
    __syncthreads();

    long long out_row = 0;
    if (tid < nbox) {
        const long long gi = box0 + tid;
        const int b = (int)(gi / per_img);
        const int r = (int)(gi - (long long)b * per_img);
        const int a = r % 3;
        const int cell = r / 3;
        const int row = cell / g, col = cell - row * g;
        out_row = (long long)b * L.a.N + L.a.off[s] + r;
        float *t = lds + tid * F;
        // grid = meshgrid(range(W), range(H)): (...,0) = col, (...,1) = row; divisor cast([H,W]) (square grid)
        const float x = (sigmoidf_(t[0]) + (float)col) / (float)g;
        const float y = (sigmoidf_(t[1]) + (float)row) / (float)g;
        const float w = expf(t[2]) * L.a.anchors[s][a][0];
        const float h = expf(t[3]) * L.a.anchors[s][a][1];
        f32x4 bb;
        bb[0] = x - w / 2;
        bb[1] = y - h / 2;
        bb[2] = x + w / 2;
        bb[3] = y + h / 2;
        *reinterpret_cast<f32x4 *>(bboxes + out_row * 4) = bb;
        const float c = sigmoidf_(t[4]);
        if (conf) conf[out_row] = c;
        float best = 0.0f;
        int besti = 0;
        for (int k = 0; k < L.a.nc; ++k) {
            const float pk = sigmoidf_(t[5 + k]);
            if (WRITE_PROBS) t[5 + k] = pk;
            if (k == 0 || pk > best) {  // first maximum wins, like tf.argmax
                best = pk;
                besti = k;
            }
        }
        if (WRITE_SCORES) {
            cls[out_row] = (int64_t)besti;
            scores[out_row] = c * best;
        }
    }
    if (WRITE_PROBS) {
        __syncthreads();
        // the block's boxes may straddle images; rows of one image are contiguous in probs, so resolve per box
        const int nc = L.a.nc;
        const int nout = nbox * nc;
        for (int i = tid; i < nout; i += DEC_BOXES) {
            const int j = i / nc, k = i - j * nc;
            const long long gi = box0 + j;
            const int b = (int)(gi / per_img);
            const int r = (int)(gi - (long long)b * per_img);
            const long long orow = (long long)b * L.a.N + L.a.off[s] + r;
            probs[orow * nc + k] = lds[j * F + 5 + k];
        }
    }
}

hipError_t launch_decode(const DecodeArgs &a, float *bboxes, float *conf, float *probs, int64_t *cls, float *scores,
                         hipStream_t s)
{
    DecodeLaunch L;
    L.a = a;
    int acc = 0;
    for (int i = 0; i < 3; ++i) {
        L.blk_start[i] = acc;
        const long long total = (long long)a.B * a.g[i] * a.g[i] * 3;
        acc += (int)((total + DEC_BOXES - 1) / DEC_BOXES);
    }
    L.blk_start[3] = acc;
    const size_t lds = (size_t)DEC_BOXES * (5 + a.nc) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(acc), block(DEC_BOXES);
    const bool wp = probs != nullptr, wsc = scores != nullptr;
    auto go = [&](auto k) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, block, lds, s, L, bboxes, conf, probs, cls, scores);
        return hipGetLastError();
    };
    if (wp && wsc) return go(decode_kernel<true, true>);
    if (wp) return go(decode_kernel<true, false>);
    if (wsc) return go(decode_kernel<false, true>);
    return go(decode_kernel<false, false>);
}

// class_indices = argmax(probs, -1) (int64, first max), scores = conf * reduce_max(probs, -1)
// reference: core/yolo_nms.py:18-24.  Stand-alone form for callers that hold [B,N,nc] probabilities.
__global__ __launch_bounds__(256) void class_scores_kernel(const float *__restrict__ conf,
                                                           const float *__restrict__ probs, size_t n, int nc,
                                                           int64_t *__restrict__ cls, float *__restrict__ scores)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = probs + i * nc;
    float best = p[0];
    int besti = 0;
    for (int k = 1; k < nc; ++k) {
        const float v = p[k];
        if (v > best) {
            best = v;
            besti = k;
        }
    }
    cls[i] = besti;
    scores[i] = conf[i] * best;
}

hipError_t launch_class_scores(const float *conf, const float *probs, size_t n, int nc, int64_t *cls, float *scores,
                               hipStream_t s)
{
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(class_scores_kernel, grid, block, 0, s, conf, probs, n, nc, cls, scores);
    return hipGetLastError();
}

}  // namespace y3
