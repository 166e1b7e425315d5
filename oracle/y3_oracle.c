/*
 * oracle/y3_oracle.c -- CPU restatement of the reference's YOLOv3 inference path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path is the HIP library in
 * yolo-v3-tf2_amd/csrc and never calls into this file.
 *
 * PARITY UNPINNED: the reference's arithmetic lives in tensorflow==2.8.1 / keras==2.8.0
 * (requirements.txt:16,41), which is not installed here and whose source is not vendored
 * in /root/reference; the reference ships no golden outputs, no weights and no hot-path
 * tests.  Each function below restates the published semantics of the
 * TF/Keras op the reference calls and cites the call site it follows.  An independent
 * PyTorch-CPU implementation (tests/test_oracle.py) and a literal NumPy
 * restatement of TF's tiled NMS (oracle/nms_tiled_ref.py) cross-check it.
 *
 * Arithmetic is plain fp32 in a fixed, documented order; build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define Y3O_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------
 * Conv2D, NHWC x HWIO -> NHWC.   reference: core/parse_model.py:27-43
 *   stride 1: padding='same'  (pad (k-1)/2 on every side)
 *   stride 2: ZeroPadding2D(((1,0),(1,0))) then padding='valid'  (core/parse_model.py:34-35)
 * Accumulation order: taps row-major (u,v), then input channel, fp32 (acc64=0) or fp64.
 * ------------------------------------------------------------------------- */
Y3O_API void y3o_conv2d(const float *x, int B, int H, int W, int Cin, const float *w, int k, int stride,
                        int Cout, float *y, int acc64)
{
    const int pad = (stride == 1) ? (k - 1) / 2 : 1; /* stride 2: top/left only; bottom/right never read */
    const int Ho = (stride == 1) ? H : (H + 1 - k) / stride + 1;
    const int Wo = (stride == 1) ? W : (W + 1 - k) / stride + 1;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)Cout);
        double *accd = (double *)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    if (acc64)
                        memset(accd, 0, sizeof(double) * (size_t)Cout);
                    else
                        memset(acc, 0, sizeof(float) * (size_t)Cout);
                    for (int u = 0; u < k; ++u) {
                        const int hi = ho * stride - pad + u;
                        if (hi < 0 || hi >= H) continue;
                        for (int v = 0; v < k; ++v) {
                            const int wi = wo * stride - pad + v;
                            if (wi < 0 || wi >= W) continue;
                            const float *xp = x + (((size_t)b * H + hi) * W + wi) * Cin;
                            const float *wp = w + ((size_t)(u * k + v) * Cin) * Cout;
                            for (int c = 0; c < Cin; ++c) {
                                const float xv = xp[c];
                                const float *wr = wp + (size_t)c * Cout;
                                if (acc64) {
                                    for (int n = 0; n < Cout; ++n) accd[n] += (double)xv * (double)wr[n];
                                } else {
                                    for (int n = 0; n < Cout; ++n) acc[n] += xv * wr[n];
                                }
                            }
                        }
                    }
                    float *yp = y + (((size_t)b * Ho + ho) * Wo + wo) * Cout;
                    if (acc64)
                        for (int n = 0; n < Cout; ++n) yp[n] = (float)accd[n];
                    else
                        memcpy(yp, acc, sizeof(float) * (size_t)Cout);
                }
        free(acc);
        free(accd);
    }
}

/* BatchNormalization() inference with Keras defaults (eps passed by the caller = 1e-3), then
 * LeakyReLU(alpha=0.1).   reference: core/parse_model.py:45-52
 *   scale = gamma * rsqrt(var+eps); y = x*scale + (beta - mean*scale)  (TF fused/non-fused
 *   inference both evaluate this form)                                                    */
Y3O_API void y3o_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                         int C, float *scale, float *shift)
{
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + eps);
        scale[c] = inv * gamma[c];
        shift[c] = beta[c] - mean[c] * scale[c];
    }
}

/* y = act(x*scale + shift); for bias convs scale==1, shift==bias (use_bias = not BN, parse_model.py:41) */
Y3O_API void y3o_affine_act(float *y, size_t npix, int C, const float *scale, const float *shift, int leaky)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p) {
        float *yp = y + p * C;
        for (int c = 0; c < C; ++c) {
            float v = yp[c] * scale[c] + shift[c];
            if (leaky) v = (v >= 0.0f) ? v : 0.1f * v;
            yp[c] = v;
        }
    }
}

/* bias only: y = x + b (Conv2D use_bias=True, linear) */
Y3O_API void y3o_bias(float *y, size_t npix, int C, const float *bias)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c) y[p * C + c] += bias[c];
}

/* Add()([from, x])   reference: core/parse_model.py:155-156 */
Y3O_API void y3o_add(const float *a, const float *b, float *y, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = a[i] + b[i];
}

/* UpSampling2D(size=2), nearest   reference: core/parse_model.py:72 */
Y3O_API void y3o_upsample2x(const float *x, int B, int H, int W, int C, float *y)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < 2 * H; ++h)
            for (int w2 = 0; w2 < 2 * W; ++w2)
                memcpy(y + (((size_t)b * 2 * H + h) * 2 * W + w2) * C,
                       x + (((size_t)b * H + h / 2) * W + w2 / 2) * C, sizeof(float) * (size_t)C);
}

/* Concatenate(axis=3)([a, b])   reference: core/parse_model.py:134 */
Y3O_API void y3o_concat(const float *a, int Ca, const float *b, int Cb, size_t npix, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p) {
        memcpy(y + p * (Ca + Cb), a + p * Ca, sizeof(float) * (size_t)Ca);
        memcpy(y + p * (Ca + Cb) + Ca, b + p * Cb, sizeof(float) * (size_t)Cb);
    }
}

/* ---------------------------------------------------------------------------
 * yolo_decode for one scale.   reference: core/yolo_decode_layer.py:4-36
 * grid [B,gh,gw,3,5+nc]; anchors [3][2] (normalised w,h); writes rows
 * [off, off+gh*gw*3) of bboxes[B,N,4], conf[B,N,1], probs[B,N,nc].
 * ------------------------------------------------------------------------- */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

Y3O_API void y3o_decode_scale(const float *grid, int B, int gh, int gw, int nc, const float *anchors, int off, int N,
                              float *bboxes, float *conf, float *probs)
{
    const int F = 5 + nc;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int row = 0; row < gh; ++row)
            for (int col = 0; col < gw; ++col)
                for (int a = 0; a < 3; ++a) {
                    const float *t = grid + ((((size_t)b * gh + row) * gw + col) * 3 + a) * F;
                    const size_t n = (size_t)b * N + off + ((size_t)row * gw + col) * 3 + a;
                    /* grid = meshgrid(range(W), range(H)): (...,0)=col, (...,1)=row; divisor = cast([H,W]) */
                    const float x = (sigmoidf_(t[0]) + (float)col) / (float)gh;
                    const float y = (sigmoidf_(t[1]) + (float)row) / (float)gw;
                    const float w = expf(t[2]) * anchors[a * 2 + 0];
                    const float h = expf(t[3]) * anchors[a * 2 + 1];
                    bboxes[n * 4 + 0] = x - w / 2;
                    bboxes[n * 4 + 1] = y - h / 2;
                    bboxes[n * 4 + 2] = x + w / 2;
                    bboxes[n * 4 + 3] = y + h / 2;
                    conf[n] = sigmoidf_(t[4]);
                    for (int k = 0; k < nc; ++k) probs[n * nc + k] = sigmoidf_(t[5 + k]);
                }
}

/* class_indices = argmax(probs) (int64, first max), scores = conf * max(probs)
 * reference: core/yolo_nms.py:18-24 */
Y3O_API void y3o_scores(const float *conf, const float *probs, size_t n, int nc, int64_t *cls, float *scores)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float *p = probs + i * nc;
        int best = 0;
        float m = p[0];
        for (int k = 1; k < nc; ++k)
            if (p[k] > m) {
                m = p[k];
                best = k;
            }
        cls[i] = best;
        scores[i] = conf[i] * m;
    }
}

/* ---------------------------------------------------------------------------
 * tf.image.non_max_suppression_padded(boxes[B,N,4], scores[B,N], M, T, S,
 * pad_to_max_output_size=True) -- sequential statement of what the tiled "v2"
 * algorithm of TF 2.8 computes (literal tile-by-tile form in
 * oracle/nms_tiled_ref.py).  tile_size = 512.
 * reference call site: core/yolo_nms.py:26-33
 * ------------------------------------------------------------------------- */
typedef struct {
    float s;
    int32_t i;
} y3o_key;

static int key_cmp(const void *a, const void *b)
{
    const y3o_key *x = (const y3o_key *)a, *y = (const y3o_key *)b;
    if (x->s > y->s) return -1; /* descending score */
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i); /* top_k: lower index first among equals */
}

static inline float iou_(const float *a, const float *b)
{
    /* _bbox_overlap: boxes read as (y1,x1,y2,x2) = (c0,c1,c2,c3); fp32, this association order */
    const float i_xmin = fmaxf(a[1], b[1]), i_xmax = fminf(a[3], b[3]);
    const float i_ymin = fmaxf(a[0], b[0]), i_ymax = fminf(a[2], b[2]);
    const float i_area = fmaxf(i_xmax - i_xmin, 0.0f) * fmaxf(i_ymax - i_ymin, 0.0f);
    const float a_area = (a[2] - a[0]) * (a[3] - a[1]);
    const float b_area = (b[2] - b[0]) * (b[3] - b[1]);
    const float u_area = a_area + b_area - i_area + 1e-8f;
    return i_area / u_area;
}

Y3O_API void y3o_nms_padded(const float *boxes_in, const float *scores_in, int B, int N, int M, float T, float S,
                            int32_t *idx_out, int32_t *num_valid)
{
    /* step 1: score mask by multiplication */
    float *boxes = (float *)malloc(sizeof(float) * 4 * (size_t)B * N);
    float *scores = (float *)malloc(sizeof(float) * (size_t)B * N);
    for (size_t i = 0; i < (size_t)B * N; ++i) {
        const float m = (scores_in[i] > S) ? 1.0f : 0.0f;
        scores[i] = scores_in[i] * m;
        for (int c = 0; c < 4; ++c) boxes[i * 4 + c] = boxes_in[i * 4 + c] * m;
    }
    /* step 2: canonicalise coordinates; the decision looks at box [0,0] only and applies batch-wide */
    const int swap_y = !(boxes[0] <= boxes[2]);
    const int swap_x = !(boxes[1] <= boxes[3]);
    if (swap_y || swap_x)
        for (size_t i = 0; i < (size_t)B * N; ++i) {
            float *p = boxes + i * 4;
            if (swap_y) { float t = p[0]; p[0] = p[2]; p[2] = t; }
            if (swap_x) { float t = p[1]; p[1] = p[3]; p[3] = t; }
        }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *bx = boxes + (size_t)b * N * 4;
        y3o_key *keys = (y3o_key *)malloc(sizeof(y3o_key) * (size_t)N);
        for (int i = 0; i < N; ++i) {
            keys[i].s = scores[(size_t)b * N + i];
            keys[i].i = i;
        }
        /* step 3: descending sort, ties by lower original index (argsort DESCENDING == top_k(k=N)) */
        qsort(keys, (size_t)N, sizeof(y3o_key), key_cmp);
        /* step 5: greedy suppression in sorted order with `iou >= T`.  All-zero boxes have IoU 0 with
         * everything; for T > 0 they neither suppress nor get suppressed, so they are skipped. */
        int32_t *alive = (int32_t *)malloc(sizeof(int32_t) * (size_t)N); /* surviving sorted positions */
        int n_alive = 0, n_sel = 0;
        int32_t *out = idx_out + (size_t)b * M;
        for (int m = 0; m < M; ++m) out[m] = 0;
        if (!(T > 0.0f)) {
            /* T <= 0 quirk of the tiled algorithm: inside a tile `max(iou) < T` never holds, so no box "can
             * suppress others" and self-suppression removes nothing; across tiles `all(iou < T)` never holds,
             * so every tile after the first (512 sorted positions) is wiped.  Net effect: the first tile
             * survives untouched. */
            const int lim = N < 512 ? N : 512;
            for (int j = 0; j < lim && n_sel < M; ++j) {
                const float *bj = bx + (size_t)keys[j].i * 4;
                if (bj[0] > 0.0f || bj[1] > 0.0f || bj[2] > 0.0f || bj[3] > 0.0f) out[n_sel++] = keys[j].i;
            }
            num_valid[b] = n_sel;
            free(alive);
            free(keys);
            continue;
        }
        for (int j = 0; j < N && n_sel < M; ++j) {
            const float *bj = bx + (size_t)keys[j].i * 4;
            const int zero = (bj[0] == 0.0f && bj[1] == 0.0f && bj[2] == 0.0f && bj[3] == 0.0f);
            int suppressed = 0;
            if (!zero)
                for (int a = 0; a < n_alive && !suppressed; ++a)
                    if (iou_(bx + (size_t)keys[alive[a]].i * 4, bj) >= T) suppressed = 1;
            if (suppressed) continue;
            if (!zero) alive[n_alive++] = j;
            /* step 6: a position is selected iff any coordinate of its box is > 0 */
            if (bj[0] > 0.0f || bj[1] > 0.0f || bj[2] > 0.0f || bj[3] > 0.0f) out[n_sel++] = keys[j].i;
        }
        num_valid[b] = n_sel; /* step 7: min(output_size, M); entries >= num_valid stay 0 */
        free(alive);
        free(keys);
    }
    free(boxes);
    free(scores);
}

/* Inference.gather_valid_detections_results   reference: inference.py:21-28 */
Y3O_API void y3o_gather_valid(const float *bboxes, const int64_t *cls, const float *scores, const int32_t *sel,
                              int num_valid, float *out_boxes, int64_t *out_cls, float *out_scores)
{
    for (int i = 0; i < num_valid; ++i) {
        memcpy(out_boxes + i * 4, bboxes + (size_t)sel[i] * 4, sizeof(float) * 4);
        out_cls[i] = cls[sel[i]];
        out_scores[i] = scores[sel[i]];
    }
}

Y3O_API int y3o_version(void) { return 1; }
