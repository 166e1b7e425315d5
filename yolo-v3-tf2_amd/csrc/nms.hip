// tf.image.non_max_suppression_padded (tiled "v2" semantics) + detection packing on gfx950.
//
// Replaces the call in reference core/yolo_nms.py:26-33 and the per-image gathers of
// reference inference.py:21-28.  Semantics restated in SURVEY.md Appendix B.4 and oracle/y3_oracle.c:
//   1. boxes/scores with score <= S are zeroed (multiplication by the mask);
//   2. coordinate canonicalisation decided by box [0,0] of the batch only;
//   3. descending sort by score, ties by lower original index;
//   4. greedy suppression in sorted order, suppress when IoU >= T (fp32, TF's association order,
//      union + 1e-8); zeroed boxes have IoU 0 with everything;
//   5. a survivor is *selected* iff any of its coordinates is > 0; the first M selected positions
//      are returned as original indices (int32), the rest of the row is 0; num_valid = min(count, M).
// One workgroup (256 threads) per image: threshold+compact -> bitonic sort of 64-bit
// (score desc, index asc) keys in LDS (global scratch beyond 4096 candidates) -> chunks of 256
// candidates: every lane tests its candidate against the kept list (LDS; survivors beyond KEPT_CAP spill to the
// global workspace -- only boxes with no positive coordinate can pile up there), builds its 256-bit
// intra-chunk suppression row, and wave 0 resolves the chunk serially with scalar bit-ops.
// Integer/index work is exact; IoU arithmetic is fp32 without contraction.
// The kernel runs after the lanes of a step have joined, on B of the chip's 256 CUs with nothing beside it, so its latency chain is
// step time one to one.  Two chains were shortened in round 5 (same integer logic, same outputs): the score scan requests eight
// scores per thread before it looks at any (one global round trip per 2048 scores instead of one per 256: 6 instead of 42 at
// N = 10647), one LDS atomic per wave and pass; the serial resolve keeps the chunk's 256 suppression rows in wave 0's registers
// (four candidates per lane) and fetches row t with v_readlane -- no LDS read on the dependent chain, selected / kept positions
// are written to lists and turned into indices by all threads afterwards.  Two more (same round): the test against the kept list works on
// (candidate, kept box) pairs spread over all threads, and the suppression rows come from balanced pairs (128 steps per thread instead of
// 255 for thread 0).  93 -> 68 us per call at 64-128 images x 10647 boxes (profiles/r05_nms_chains.txt).
#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

static constexpr int NMS_THREADS = 256;
static constexpr int SORT_CAP = 4096;   // keys sorted in LDS
static constexpr int KEPT_CAP = 2048;   // surviving suppressor boxes held in LDS

__device__ __forceinline__ unsigned score_key(float s)
{
    s = s + 0.0f;  // -0 -> +0 so that equal floats give equal keys
    unsigned u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending float order
    return ~u;                                   // descending
}

__device__ __forceinline__ float iou_tf(const f32x4 a, const f32x4 b)
{
    const float i_xmin = fmaxf(a[1], b[1]), i_xmax = fminf(a[3], b[3]);
    const float i_ymin = fmaxf(a[0], b[0]), i_ymax = fminf(a[2], b[2]);
    const float i_area = fmaxf(i_xmax - i_xmin, 0.0f) * fmaxf(i_ymax - i_ymin, 0.0f);
    const float a_area = (a[2] - a[0]) * (a[3] - a[1]);
    const float b_area = (b[2] - b[0]) * (b[3] - b[1]);
    const float u_area = a_area + b_area - i_area + 1e-8f;
    return i_area / u_area;
}

__device__ void bitonic_sort(u64 *keys, int P, int tid)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += NMS_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // element with bit j clear
                const int l = i | j;
                const bool up = (i & k) == 0;
                const u64 a = keys[i], b = keys[l];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

struct NmsArgs {
    const float *boxes;   // [B,N,4]
    const float *scores;  // [B,N]
    int N, M;
    float T, S;
    int32_t *sel;         // [B,M]
    int32_t *num_valid;   // [B]
    u64 *ws;              // [B][P2] scratch keys, P2 = next pow2 >= N
    int P2;
    float *spill;         // [B][N][4] kept boxes beyond KEPT_CAP
};

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const NmsArgs p)
{
    __shared__ __attribute__((aligned(16))) u64 s_keys[SORT_CAP];
    __shared__ __attribute__((aligned(16))) float s_kbox[KEPT_CAP * 4];
    __shared__ __attribute__((aligned(16))) float s_cbox[NMS_THREADS * 4];
    __shared__ u64 s_mask[NMS_THREADS * 4];
    __shared__ u64 s_alive[4], s_posany[4];
    __shared__ int s_keptpos[NMS_THREADS], s_selpos[NMS_THREADS], s_ok[NMS_THREADS];
    __shared__ int s_cnt, s_nkept_chunk, s_nsel, s_nalive;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *boxes = p.boxes + (size_t)b * p.N * 4;
    const float *scores = p.scores + (size_t)b * p.N;
    int32_t *sel = p.sel + (size_t)b * p.M;
    u64 *gkeys = p.ws + (size_t)b * p.P2;
    float *gspill = p.spill + (size_t)b * p.N * 4;

    // canonicalisation flags from box [0,0] of the batch after masking (TF looks at that box only)
    const float m00 = (p.scores[0] > p.S) ? 1.0f : 0.0f;
    const bool swap_y = !(p.boxes[0] * m00 <= p.boxes[2] * m00);
    const bool swap_x = !(p.boxes[1] * m00 <= p.boxes[3] * m00);

    if (tid == 0) {
        s_cnt = 0;
        s_nsel = 0;
        s_nalive = 0;
    }
    for (int i = tid; i < p.M; i += NMS_THREADS) sel[i] = 0;
    __syncthreads();

    // ---- 1. threshold + compact (order irrelevant: the sort key carries the index) -----------------
    // eight scores per thread are requested before any is looked at; a wave reserves its slots with one LDS atomic per pass
    constexpr int SCAN_UNROLL = 8;
    const int lane = tid & 63;
    for (int i0 = 0; i0 < p.N; i0 += NMS_THREADS * SCAN_UNROLL) {
        float sv[SCAN_UNROLL];
#pragma unroll
        for (int k = 0; k < SCAN_UNROLL; ++k) {
            const int i = i0 + k * NMS_THREADS + tid;
            sv[k] = (i < p.N) ? scores[i] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < SCAN_UNROLL; ++k) {
            const int i = i0 + k * NMS_THREADS + tid;
            const bool hit = i < p.N && sv[k] > p.S;
            const u64 bal = __ballot(hit);
            if (bal == 0) continue;                       // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, __builtin_popcountll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) {
                const int slot = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                const u64 key = ((u64)score_key(sv[k]) << 32) | (unsigned)i;
                gkeys[slot] = key;
                if (slot < SORT_CAP) s_keys[slot] = key;
            }
        }
    }
    __syncthreads();
    // T <= 0 (tile quirk, see oracle/y3_oracle.c): nothing is suppressed inside the first 512 sorted positions and
    // everything behind them is wiped.  (The host rejects T <= 0 together with S < 0, where zeroed boxes could
    // sort in front of kept ones.)
    const bool degenerate = !(p.T > 0.0f);
    int nc = s_cnt;
    // ---- 2. sort ------------------------------------------------------------------------------------
    int P = 1;
    while (P < nc) P <<= 1;
    u64 *keys;
    if (P <= SORT_CAP) {
        for (int i = nc + tid; i < P; i += NMS_THREADS) s_keys[i] = ~0ull;   // the scan left the keys themselves here
        keys = s_keys;
    } else {
        for (int i = nc + tid; i < P; i += NMS_THREADS) gkeys[i] = ~0ull;
        keys = gkeys;
    }
    __syncthreads();
    bitonic_sort(keys, P, tid);
    if (degenerate) nc = min(nc, 512);

    // ---- 3. greedy suppression, 256 sorted candidates at a time ---------------------------------------
    for (int pos = 0; pos < nc; pos += NMS_THREADS) {
        const int cnt = min(NMS_THREADS, nc - pos);
        const int nalive = s_nalive;
        f32x4 mine = {0.f, 0.f, 0.f, 0.f};
        if (tid < cnt) {
            const int my_idx = (int)(unsigned)(keys[pos + tid] & 0xFFFFFFFFull);
            mine = *reinterpret_cast<const f32x4 *>(boxes + (size_t)my_idx * 4);
            if (swap_y) { const float t = mine[0]; mine[0] = mine[2]; mine[2] = t; }
            if (swap_x) { const float t = mine[1]; mine[1] = mine[3]; mine[3] = t; }
        }
        *reinterpret_cast<f32x4 *>(s_cbox + tid * 4) = mine;
        s_ok[tid] = tid < cnt ? 1 : 0;
        unsigned *m32 = reinterpret_cast<unsigned *>(s_mask);   // the chunk's suppression rows as 8 x 32 bits per candidate
#pragma unroll
        for (int w = 0; w < 8; ++w) m32[tid * 8 + w] = 0u;
        __syncthreads();
        // candidates against the kept list, the (candidate, kept box) pairs spread over all 256 threads: a chunk of cnt candidates takes
        // nalive * cp2 / 256 passes (cp2 = cnt rounded up to a power of two) -- a full chunk one kept box per pass and thread as before, the
        // short last chunk of an image (a handful of candidates against ~170 kept boxes) a few passes instead of one per kept box
        if (!degenerate && nalive > 0) {
            int sh = 0;
            while ((1 << sh) < cnt) ++sh;
            const int c = tid & ((1 << sh) - 1), a0 = tid >> sh, astep = NMS_THREADS >> sh;
            if (c < cnt) {
                const f32x4 cb = *reinterpret_cast<const f32x4 *>(s_cbox + c * 4);
                bool hit = false;
                for (int a = a0; a < nalive; a += astep) {
                    const f32x4 kb = a < KEPT_CAP ? *reinterpret_cast<const f32x4 *>(s_kbox + a * 4)
                                                  : *reinterpret_cast<const f32x4 *>(gspill + (size_t)(a - KEPT_CAP) * 4);   // spilled survivors (rare: see the header)
                    if (iou_tf(kb, cb) >= p.T) hit = true;
                }
                if (hit) s_ok[c] = 0;
            }
            __syncthreads();
        }
        const bool ok = s_ok[tid] != 0;
        const u64 bal = __ballot(ok);
        const u64 pbal = __ballot(mine[0] > 0.0f || mine[1] > 0.0f || mine[2] > 0.0f || mine[3] > 0.0f);   // "selected" iff any coordinate > 0
        if ((tid & 63) == 0) {
            s_alive[tid >> 6] = bal;
            s_posany[tid >> 6] = pbal;
        }
        // suppression rows: bit j of row i (i < j) = candidate i suppresses the later candidate j.  Rows of candidates that are not alive are
        // never applied, so every pair is simply evaluated once -- IoU(earlier, later), the argument order of the sequential definition.
        if (!degenerate) {
            if (cnt >= NMS_THREADS / 2) {
                // balanced: thread t takes the pairs (t, t + d mod 256), d = 1..128 (d = 128 from the lower half only): 128 steps for every
                // thread instead of 255 for thread 0; the bit goes to the earlier candidate's row with an LDS atomic
                for (int d = 1; d <= NMS_THREADS / 2; ++d) {
                    if (d == NMS_THREADS / 2 && tid >= NMS_THREADS / 2) break;
                    const int j = (tid + d) & (NMS_THREADS - 1);
                    const int lo = j > tid ? tid : j, hi = j > tid ? j : tid;
                    if (hi >= cnt) continue;
                    const f32x4 other = *reinterpret_cast<const f32x4 *>(s_cbox + j * 4);
                    const f32x4 a = j > tid ? mine : other, b = j > tid ? other : mine;
                    if (iou_tf(a, b) >= p.T) atomicOr(&m32[lo * 8 + (hi >> 5)], 1u << (hi & 31));
                }
            } else if (ok) {
                for (int j = tid + 1; j < cnt; ++j)
                    if (iou_tf(mine, *reinterpret_cast<const f32x4 *>(s_cbox + j * 4)) >= p.T) m32[tid * 8 + (j >> 5)] |= 1u << (j & 31);
            }
        }
        __syncthreads();
        const int nsel0 = s_nsel;
        if (tid < 64) {
            // wave 0, every lane redundantly (uniform control flow): walk the alive bits in order.  Lane L holds the rows of
            // candidates L, 64 + L, 128 + L, 192 + L; row t = lane (t & 63) of set (t >> 6), fetched with v_readlane.
            u64 r[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int w = 0; w < 4; ++w) r[c][w] = s_mask[(c * 64 + tid) * 4 + w];
            u64 al[4] = {s_alive[0], s_alive[1], s_alive[2], s_alive[3]};
            const u64 pa[4] = {s_posany[0], s_posany[1], s_posany[2], s_posany[3]};
            int nsel = nsel0, nk = 0;
            bool full = false;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                u64 rem = full ? 0ull : al[w];
                while (rem) {
                    const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
                    const int t = w * 64 + i;
                    if (tid == 0) s_keptpos[nk] = t;
                    ++nk;
                    if ((pa[w] >> i) & 1ull) {
                        if (tid == 0) s_selpos[nsel - nsel0] = t;
                        ++nsel;
                        if (nsel >= p.M) {
                            full = true;
                            break;
                        }
                    }
                    // a row only has bits behind its own candidate: words below w are untouched
#pragma unroll
                    for (int v = w; v < 4; ++v) {
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(r[w][v] & 0xFFFFFFFFull), i);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(r[w][v] >> 32), i);
                        al[v] &= ~(((u64)hi << 32) | lo);
                    }
                    rem = (i == 63) ? 0ull : (al[w] & ~((2ull << i) - 1ull));
                }
            }
            if (tid == 0) {
                s_nsel = nsel;
                s_nkept_chunk = nk;
            }
        }
        __syncthreads();
        // the chunk's selected positions -> original indices (every thread one)
        if (tid < s_nsel - nsel0) sel[nsel0 + tid] = (int32_t)(unsigned)(keys[pos + s_selpos[tid]] & 0xFFFFFFFFull);
        const int nk = s_nkept_chunk;
        if (tid < nk && !degenerate) {
            const int t = s_keptpos[tid];
            const int slot = nalive + tid;
            float *dst = slot < KEPT_CAP ? s_kbox + slot * 4 : gspill + (size_t)(slot - KEPT_CAP) * 4;
            *reinterpret_cast<f32x4 *>(dst) = *reinterpret_cast<const f32x4 *>(s_cbox + t * 4);
        }
        if (tid == 0) s_nalive = nalive + nk;
        __syncthreads();
        if (s_nsel >= p.M) break;
    }
    if (tid == 0) p.num_valid[b] = s_nsel;
}

static int next_pow2(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// sort keys [B][P2] u64, then the kept-list spill [B][N][4] f32
size_t nms_workspace_bytes(int B, int N)
{
    return (size_t)B * next_pow2(N > 1 ? N : 1) * sizeof(u64) + (size_t)B * (N > 1 ? N : 1) * 4 * sizeof(float);
}

hipError_t launch_nms(const float *boxes, const float *scores, int B, int N, int M, float T, float S, int32_t *sel,
                      int32_t *num_valid, void *ws, hipStream_t s)
{
    const int P2 = next_pow2(N > 1 ? N : 1);
    u64 *keys = static_cast<u64 *>(ws);
    NmsArgs a{boxes, scores, N, M, T, S, sel, num_valid, keys, P2, reinterpret_cast<float *>(keys + (size_t)B * P2)};
    dim3 grid(B), block(NMS_THREADS);
    hipLaunchKernelGGL(nms_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

// Batched Inference.gather_valid_detections_results (reference: inference.py:21-28), packed rows
// {xmin, ymin, xmax, ymax, score, class, index}; rows >= num_valid are zero.
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ cls,
                                                   const float *__restrict__ scores, const int32_t *__restrict__ sel,
                                                   const int32_t *__restrict__ nv, int B, int N, int M,
                                                   unsigned *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * M) return;
    const int b = i / M, r = i - b * M;
    unsigned *o = out + (size_t)i * 7;
    if (r < nv[b]) {
        const int idx = sel[i];
        const size_t g = (size_t)b * N + idx;
        const float *bx = boxes + g * 4;
        o[0] = __float_as_uint(bx[0]);
        o[1] = __float_as_uint(bx[1]);
        o[2] = __float_as_uint(bx[2]);
        o[3] = __float_as_uint(bx[3]);
        o[4] = __float_as_uint(scores[g]);
        o[5] = (unsigned)(int)cls[g];
        o[6] = (unsigned)idx;
    } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = 0;
    }
}

hipError_t launch_pack(const float *boxes, const int64_t *cls, const float *scores, const int32_t *sel,
                       const int32_t *nv, int B, int N, int M, void *packed, hipStream_t s)
{
    dim3 grid((B * M + 255) / 256), block(256);
    hipLaunchKernelGGL(pack_kernel, grid, block, 0, s, boxes, cls, scores, sel, nv, B, N, M,
                       static_cast<unsigned *>(packed));
    return hipGetLastError();
}

}  // namespace y3
