// yolo_decode (+ the class arg-max / score of yolo_nms) on gfx950.
//
// Replaces reference core/yolo_decode_layer.py:4-36 (split, sigmoid x3, exp*anchor, meshgrid add,
// divide, +-wh/2, concat, reshape, concat) and core/yolo_nms.py:18-24 (argmax, reduce_max, multiply).
// HBM-bound: 4*(5+nc) bytes read per box.  A workgroup of 256 lanes copies 64 boxes (contiguous 64*(5+nc)
// floats of one scale) into LDS with 16-B coalesced loads; four lanes then share one box: lane q takes field q of
// (x, y, w, h) and the classes q, q+4, q+8, ... (a quarter of the sigmoids each), and the four partial
// (first-maximum, index) pairs are merged with two lane exchanges.  22 KB of LDS per workgroup keeps 7 workgroups =
// 28 waves per CU in flight, so the copy of one workgroup overlaps the sigmoids of the others.  Class probabilities are
// written back through LDS so that the [B,N,nc] store is coalesced.
// Arithmetic: sigmoid(x) = 1/(1+exp(-x)), true divisions, fp32, no contraction (-ffp-contract=off).
#include "y3_kernels.h"
#include "decode_box.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int DEC_BOXES = 64;    // boxes per workgroup
static constexpr int DEC_LANES = 4;     // lanes per box
static constexpr int DEC_THREADS = DEC_BOXES * DEC_LANES;

struct DecodeLaunch {
    DecodeArgs a;
    int blk_start[4];  // first workgroup of each scale (prefix sums), [3] = total
};

template <bool WRITE_PROBS, bool WRITE_SCORES>
__global__ __launch_bounds__(DEC_THREADS) void decode_kernel(const DecodeLaunch L, float *__restrict__ bboxes,
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
    for (int i = tid; i < nvec; i += DEC_THREADS) reinterpret_cast<f32x4 *>(lds)[i] = reinterpret_cast<const f32x4 *>(src)[i];
    for (int i = (nvec << 2) + tid; i < nfl; i += DEC_THREADS) lds[i] = src[i];
    __syncthreads();

    const int j = tid >> 2, q = tid & 3;          // box within the workgroup, lane within the box
    const bool live = j < nbox;
    const int jj = live ? j : 0;
    const long long gi = box0 + jj;
    const int b = (int)(gi / per_img);
    const int r = (int)(gi - (long long)b * per_img);
    const int a = r % 3;
    const int cell = r / 3;
    const int row = cell / g, col = cell - row * g;
    const long long out_row = (long long)b * L.a.N + L.a.off[s] + r;
    float *t = lds + jj * F;
    // lane 0: x, lane 1: y, lane 2: w, lane 3: h; classes q, q + 4, ... (decode_box.h: the body the fused head convs share)
    f32x4 bb;
    float c, best;
    int besti;
    decode_box_lanes<WRITE_PROBS, WRITE_SCORES>(t, q, tid & 63, L.a.nc, g, row, col, q >= 2 ? L.a.anchors[s][a][q - 2] : 0.0f, live, bb, c, best, besti);
    if (q == 0 && live) {
        *reinterpret_cast<f32x4 *>(bboxes + out_row * 4) = bb;
        if (conf) conf[out_row] = c;
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
        for (int i = tid; i < nout; i += DEC_THREADS) {
            const int jb = i / nc, k = i - jb * nc;
            const long long gj = box0 + jb;
            const int bj = (int)(gj / per_img);
            const int rj = (int)(gj - (long long)bj * per_img);
            const long long orow = (long long)bj * L.a.N + L.a.off[s] + rj;
            probs[orow * nc + k] = lds[jb * F + 5 + k];
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
    dim3 grid(acc), block(DEC_THREADS);
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
