// The per-box arithmetic of yolo_decode + the class arg-max / score of yolo_nms, shared by the stand-alone decode kernel
// (decode.hip) and the head convs that decode their own output tile (conv_head.hip, conv_bf16.hip): ONE body, so that the fused
// route of y3_net_detect and the composed route y3_net_forward -> y3_yolo_decode_scores give the same bits.
// reference: core/yolo_decode_layer.py:15-36 (sigmoid x3, exp * anchor, meshgrid add, divide, +-wh/2), core/yolo_nms.py:18-24
// (argmax: first maximum; score = conf * max prob).  fp32, true divisions, no contraction (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace y3 {

typedef float dec_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Four consecutive lanes (q = 0..3, aligned to 4 inside their wave) share the box whose 5 + nc logits lie at t (LDS):
// lane q takes field q of (x, y, w, h) and the classes q, q + 4, ...; the four partial (first-maximum, index) pairs are merged
// with two lane exchanges (ties -> lowest index, like tf.argmax over the whole row).  EVERY lane of the wave must call this.
// anchor_wh: anchors[a][q - 2] for q >= 2 (ignored otherwise).  On return: bb / conf are valid in lane q == 0, best / besti in
// all four lanes (MERGE) or per lane (no MERGE).  WRITE_PROBS stores the lane's class probabilities back over their logits.
template <bool WRITE_PROBS, bool MERGE>
__device__ __forceinline__ void decode_box_lanes(float *t, int q, int wave_lane, int nc, int g, int row, int col, float anchor_wh,
                                                 bool live, dec_f32x4 &bb, float &conf, float &best, int &besti)
{
    // grid = meshgrid(range(W), range(H)): (...,0) = col, (...,1) = row; divisor cast([H,W]) (square grid)
    float v;
    if (q < 2)
        v = (sigmoidf_(t[q]) + (float)(q == 0 ? col : row)) / (float)g;
    else
        v = expf(t[q]) * anchor_wh;
    const int base = wave_lane & ~3;             // first lane of this box inside the wave
    const float x = __shfl(v, base + 0), y = __shfl(v, base + 1), w = __shfl(v, base + 2), h = __shfl(v, base + 3);
    conf = 0.0f;
    if (q == 0) {
        bb[0] = x - w / 2;
        bb[1] = y - h / 2;
        bb[2] = x + w / 2;
        bb[3] = y + h / 2;
        conf = sigmoidf_(t[4]);
    }
    best = -INFINITY;
    besti = 0x7fffffff;
    for (int k = q; k < nc; k += 4) {
        const float pk = sigmoidf_(t[5 + k]);
        if (WRITE_PROBS && live) t[5 + k] = pk;
        if (k == q || pk > best) {
            best = pk;
            besti = k;
        }
    }
    if (MERGE) {
#pragma unroll
        for (int m = 1; m < 4; m <<= 1) {
            const float ob = __shfl_xor(best, m);
            const int oi = __shfl_xor(besti, m);
            if (ob > best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
    }
}

// A head conv's view of the decode: where its scale's boxes go.  boxes == nullptr: no fused decode.
struct DecodeHead {
    float *boxes;        // [B, N, 4] of the (sub-)batch this launch covers
    int64_t *cls;        // [B, N]
    float *scores;       // [B, N]
    int g, off, N, nc;   // grid size of this scale, first box index of the scale, boxes per image, classes
    float anchors[3][2];
};

// Decode the boxes of `nrows` consecutive output pixels whose logits sit in LDS as C[r * stride + channel] (3 x (5 + nc)
// channels per pixel), pixel r being row m_of(r) = (image, cell) of this launch; NT threads, all of them must call.
template <int NT, typename MOF>
__device__ __forceinline__ void decode_rows_from_lds(float *C, int stride, int nrows, MOF m_of, int M, const DecodeHead &h)
{
    const int tid = threadIdx.x;
    const int F = 5 + h.nc;
    const int cells = h.g * h.g;
    const int tasks = nrows * 12;                      // 3 boxes per pixel x 4 lanes per box
    for (int base = 0; base < tasks; base += NT) {
        const int idx = base + tid;
        const int box = idx >> 2, q = idx & 3;
        int r = box / 3;
        const int a = box - r * 3;
        const bool in = idx < tasks;
        if (!in) r = 0;
        const int m = m_of(r);
        const bool live = in && m < M;
        const int mm = live ? m : 0;
        const int b = mm / cells;
        const int cell = mm - b * cells;
        const int row = cell / h.g, col = cell - row * h.g;
        float *t = C + r * stride + a * F;
        dec_f32x4 bb;
        float conf, best;
        int besti;
        decode_box_lanes<false, true>(t, q, tid & 63, h.nc, h.g, row, col, q >= 2 ? h.anchors[a][q - 2] : 0.0f, live, bb, conf, best, besti);
        if (q == 0 && live) {
            const long long out_row = (long long)b * h.N + h.off + cell * 3 + a;
            *reinterpret_cast<dec_f32x4 *>(h.boxes + out_row * 4) = bb;
            h.cls[out_row] = (int64_t)besti;
            h.scores[out_row] = conf * best;
        }
    }
}

}  // namespace y3
