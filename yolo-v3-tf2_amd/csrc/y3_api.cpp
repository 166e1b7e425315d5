// C-ABI layer of liby3hip.so: network object (fused conv program), weight packing, activation arena,
// and the decode / NMS entry points.  See include/y3.h for the contract each function implements and
// the reference interface it replaces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <vector>

#include "../../include/y3.h"
#include "y3_kernels.h"

namespace {

// The last error of this thread: a fixed buffer, so that reporting a failure allocates nothing and cannot itself throw
// (include/y3.h: no entry point throws or aborts -- not even while it reports that the host ran out of memory).
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) noexcept
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

namespace y3 {
// for the other translation units of the library (comm.cpp)
int fail_msg(int code, const char *fmt, ...) noexcept
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// The exception barrier of the C ABI.  Every extern "C" entry point that can reach an allocation (std::vector, new, std::string)
// is a function-try-block ending in Y3_CATCH: a C++ exception becomes a status + message instead of crossing the boundary and
// terminating the host process (a ctypes / cgo / JNI caller has no handler for it).
int on_exception(const char *who) noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail_msg(Y3_ERR_OOM, "%s: out of host memory (std::bad_alloc)", who);
    } catch (const std::exception &e) {
        return fail_msg(Y3_ERR_INTERNAL, "%s: C++ exception: %s", who, e.what());
    } catch (...) {
        return fail_msg(Y3_ERR_INTERNAL, "%s: unknown C++ exception", who);
    }
}

// Test hook (tests/test_abi.py): Y3_TEST_FAIL_ALLOC=1 makes the object allocations of y3_net_create / y3_comm_init_rank fail the
// way operator new does; read on every call so that a test can switch it on and off inside one process.
bool test_fail_alloc() noexcept
{
    const char *e = getenv("Y3_TEST_FAIL_ALLOC");
    return e && e[0] == '1';
}
}  // namespace y3

#define Y3_CATCH(who) catch (...) { return y3::on_exception(who); }

namespace {

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? Y3_ERR_OOM : Y3_ERR_HIP, "%s: %s", #expr,     \
                        hipGetErrorString(e_));                                                   \
    } while (0)

// Enter the net's device for the duration of a call and give the caller its own current device back on every return path (a
// process driving several GPUs -- PyTorch with nets on different devices -- must not find its current device changed).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev); else if (err == hipSuccess) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define Y3_ENTER_DEVICE(net_)                                                                      \
    DeviceGuard dev_guard_((net_)->device);                                                        \
    if (dev_guard_.err != hipSuccess) return fail(Y3_ERR_HIP, "hipSetDevice(%d): %s", (net_)->device, hipGetErrorString(dev_guard_.err))

struct ConvSlot {
    y3_conv_desc d{};
    bool loaded = false;
    bool first_layer = false;  // Cin == 3 direct kernel
    int cout_pad = 0;
    int K = 0;
    int tile = -1;             // -1: choose by heuristic at plan time
    int tile_bf16 = -1;
    int tile_x3 = -1;
    int cout_pad64 = 0;        // Cout rounded up to 64 (the three-plane kernel has no 32-wide N tile)
    void *wx3_dev = nullptr;   // packed [CoutPad64][3 planes][K] bf16 (hi, mid, lo of the fp32 weights)
    int tile_x2 = -1;
    bool x2_ok = true;         // false: a BN-scaled weight is outside the fp16 range, the two-plane mode cannot be planned
    void *wx2_dev = nullptr;   // packed [CoutPad64][2 planes][K] fp16 (h, l' = (w - h) * 2^11 of the BN-scaled weights)
    float *w_dev = nullptr;    // packed [CoutPad][K] (or HWIO for the first layer)
    float *w0stem_dev = nullptr;   // first layer only: [28][Cout] = HWIO rows x BN scale, row 27 zero (fused stem kernel, fp32)
    float *w0raw_dev = nullptr;    // first layer only: the same without the scale (fused stem kernel, bf16 mode)
    void *wbf_dev = nullptr;   // same, bf16 (not for the first layer)
    float *scale_dev = nullptr;
    float *shift_dev = nullptr;
};

struct Op {
    int kind;  // 0 conv, 1 aux
    int index;
};

constexpr int Y3_MAX_LANES = 4;
constexpr int Y3_MAX_OUTPUT_BOXES = 1024;   // upper bound of max_output_size (y3_nms_padded) the detect scratch is sized for

}  // namespace

struct y3_net {
    int device = 0;
    int n_cus = 0;                 // compute units of `device`, read once by y3_net_plan (grids of the persistent kernels)
    std::vector<y3_tensor_desc> tensors;
    std::vector<Op> ops;
    std::vector<ConvSlot> convs;
    std::vector<y3_aux_desc> aux;
    int input_tensor = 0;
    int outputs[3] = {0, 0, 0};
    int nclasses = 0;
    // plan
    int max_batch = 0, image_size = 0, dtype = Y3_DTYPE_F32;
    int keep_all = 0;              // 1: no buffer reuse, every intermediate stays readable after a forward
    int lanes = 1;                 // sub-batches run concurrently on forked streams (y3_net_set_lanes)
    int early_convs = 0;           // y3_net_set_early_chunk: the first early_convs convs run early_chunk images at a time
    int early_chunk = 0;
    int early_ops = 0;             // (at plan time) number of leading ops that form the chunked segment
    std::vector<char> dense;       // tensor written by the chunked segment: own block, image i at i * image_bytes
    // (non-fp32 modes) output tensors that another op reads, or that a residual / first-layer conv writes: produced in
    // the arena in the mode's own format and converted into the caller's fp32 buffer at the end of the forward
    std::vector<char> staged;
    // y3_net_detect scratch (grids, decoded boxes / classes / scores, selected indices, NMS workspace): allocated by
    // y3_net_plan for max_batch images and Y3_MAX_OUTPUT_BOXES rows, so y3_net_detect itself only enqueues work
    void *det_buf = nullptr;
    size_t det_bytes = 0;
    // y3_net_forward_decode: while set, the three head convs decode their own tiles into these buffers (per scale: first box
    // index, grid size, anchors) instead of writing grids.  Null outside that call.
    const y3::DecodeHead *fuse = nullptr;   // [3], in output order
    int stem_mode = 1;             // y3_net_set_stem_fusion: 1 = conv0 + conv1 (+ the 1x1 after them) as one kernel when the graph allows it; 2 = conv0 + conv1 only
    bool stem_mode_set = false;    // y3_net_set_stem_fusion was called (the Y3_STEM_MODE tool override then stays out)
    bool stem_fused = false;       // (at plan time) the first two convs run as the fused stem kernel
    bool stem_conv2 = false;       // ... and the 1x1 conv that follows them (64 -> 32) runs inside it as well (fp32 and bf16 plans)
    unsigned long long *clk_stamps = nullptr;   // y3_net_measure_sclk: device buffer one conv launch stamps into (else null)
    int clk_conv = -1;                          // ... and which conv (-2: every conv, 8 words each at clk_stamps + 8 conv)
    int xcd_mode = 1;              // y3_net_set_xcd_mode: 0 contiguous tile runs per XCD, 1 XCD-blocked order chosen per conv
    int k_chunk = -1;              // y3_net_set_k_chunk: fp32 3x3 convs walk K chunk-major, this many input channels per chunk; 0 tap-major; -1 per-conv default
    int cur_batch = 1;             // batch of the forward being enqueued
    hipEvent_t fork_ev = nullptr;
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t join_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> tdev;      // arena pointer per tensor (nullptr: not materialised / external)
    std::vector<size_t> tbytes;    // bytes at max_batch
    std::vector<size_t> tblock;    // size of the arena block the tensor lives in
    std::vector<void *> blocks;    // distinct hipMalloc'ed blocks
};

namespace {

int spatial(const y3_net *n, int t) { return n->image_size / n->tensors[t].div; }

void free_plan(y3_net *n)
{
    if (n->det_buf) (void)hipFree(n->det_buf);
    n->det_buf = nullptr;
    n->det_bytes = 0;
    for (void *p : n->blocks) (void)hipFree(p);
    n->blocks.clear();
    n->tdev.assign(n->tensors.size(), nullptr);
}

unsigned short f32_to_bf16_rne(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

float bf16_to_f32(unsigned short h)
{
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fp32 -> fp16 bits, round to nearest even, subnormals kept, >= 65520 -> inf
unsigned short f32_to_f16_rne(float f)
{
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned short sign = (unsigned short)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);
    if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);
    if (x < 0x38800000u) {   // below 2^-14: a multiple of 2^-24
        float a;
        memcpy(&a, &x, 4);
        return (unsigned short)(sign | (unsigned short)lrintf(a * 16777216.0f));
    }
    unsigned h = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const unsigned rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (unsigned short)(sign | h);
}

float f16_to_f32(unsigned short h)
{
    const int e = (h >> 10) & 31, m = h & 0x3ff;
    float v;
    if (e == 0)
        v = ldexpf((float)m, -24);
    else if (e == 31)
        v = m ? NAN : INFINITY;
    else
        v = ldexpf((float)(1024 + m), e - 25);
    return (h & 0x8000) ? -v : v;
}

int choose_tile_x2(const ConvSlot &c, long long M)
{
    // widest tile that still gives every CU at least two workgroups
    const int cand128[] = {4, 8, 0, 3, 2}, cand64[] = {1, 2};
    const int *cand = c.cout_pad64 % 128 == 0 ? cand128 : cand64;
    const int n = c.cout_pad64 % 128 == 0 ? 5 : 2;
    int best = cand[n - 1];
    for (int k = 0; k < n; ++k) {
        y3::TileInfo s = y3::conv_x3_tile_info(cand[k]);
        if (c.cout_pad64 % s.bn || c.d.cin % s.stages || (c.d.src1 >= 0 && c.d.c0 % s.stages)) continue;
        long long blocks = ((M + s.bm - 1) / s.bm) * (c.cout_pad64 / s.bn);
        if (blocks >= 512) {
            best = cand[k];
            break;
        }
    }
    return best;
}

int choose_tile_x3(const ConvSlot &c, long long M)
{
    std::vector<int> cand;
    if (c.cout_pad64 % 128 == 0)
        cand = {0, 3, 2};
    else
        cand = {1, 2};
    int best = cand.back();
    for (int t : cand) {
        y3::TileInfo s = y3::conv_x3_tile_info(t);
        if (c.cout_pad64 % s.bn) continue;
        long long blocks = ((M + s.bm - 1) / s.bm) * (c.cout_pad64 / s.bn);
        if (blocks >= 512) {
            best = t;
            break;
        }
    }
    return best;
}

// M = rows of this call (per lane); M_plan = rows of the planned batch.  The MFMA SHAPE (16x16x32 vs 32x32x16: two K groupings,
// results differ in the last bits) is decided from plan-time quantities only, so that an image's result does not depend on the
// batch or lane it runs in (y3_net_set_lanes: "results are unchanged"); the tile SIZE within one shape follows the call.
int choose_tile_bf16(const ConvSlot &c, long long M, long long M_plan, bool bf16_out)
{
    auto blocks = [&](int t) {
        y3::TileInfo s = y3::conv_bf16_tile_info(t);
        return ((M + s.bm - 1) / s.bm) * (c.cout_pad / s.bn);
    };
    // large 3x3 convs: the 16x16x32 form once the PLANNED batch fills the chip with 256x256 tiles of 16 waves (tile 24 wins every
    // such signature of the 64- and 128-image tables, tuning/bf16_b*_s416.json); smaller calls of the same plan take the 128x128 /
    // 64x128 tiles of the same MFMA shape (27, 29)
    if (c.d.size == 3 && c.d.src1 < 0 && c.d.cin % 64 == 0 && c.cout_pad % 256 == 0 && ((M_plan + 255) / 256) * (c.cout_pad / 256) >= 256) {
        if (blocks(24) >= 256) return 24;
        return blocks(27) >= 512 ? 27 : 29;
    }
    // early 3x3 / stride-1 convs with Cin = 32 / 64 (K = 288 / 576): weights resident in LDS, input patch by LDS-DMA (tile id 32,
    // conv_res_bf16.hip) -- from the conv's shape alone, so batch- and lane-independent
    if (bf16_out && c.d.size == 3 && c.d.stride == 1 && c.d.src1 < 0 && (c.d.cin == 32 || c.d.cin == 64) && c.d.cout % 64 == 0) return 32;
    std::vector<int> cand;
    if (c.d.cin % 64)
        cand = {5, 6};                       // BK = 32 (Cin = 32 layers, Cout = 64)
    else if (c.cout_pad % 128 == 0)
        cand = {8, 12, 11};                  // LDS-DMA variants: 128x128, 64x128, 64x64
    else if (c.cout_pad % 64 == 0)
        cand = {10, 11};
    else
        cand = {4};
    int best = cand.back();
    for (int t : cand) {
        if (c.cout_pad % y3::conv_bf16_tile_info(t).bn) continue;
        if (blocks(t) >= 512) {
            best = t;
            break;
        }
    }
    return best;
}

// channels per K chunk of a 3x3 fp32 conv when the caller has not chosen (y3_net_set_k_chunk(-1)); Y3_K_CHUNK overrides (tools)
int default_k_chunk(const ConvSlot &c)
{
    static const int env = [] { const char *e = getenv("Y3_K_CHUNK"); return e ? atoi(e) : -1; }();
    if (env >= 0) return env;
    // 128 channels per chunk: traffic beyond L2 of the whole conv stack 42.1 -> 29.6 GB per 64-image step (1.96 x -> 1.42 x the
    // algorithmic bytes) at the same images/s (-0.1 %, inside the run-to-run spread); 64 per chunk: 27.2 GB but -0.4 %
    // (profiles/r03_k_chunk_sweep.txt, r03_traffic_per_layer_*.txt)
    return (c.d.size == 3 && c.d.cin >= 256) ? 128 : 0;
}

// How the 8 XCDs (each with its own 4 MB L2) divide the tile matrix of one fp32 conv launch: as a (8/gn) x gn grid of
// blocks.  An XCD then streams 1/gn of the weights and 1/gm of the activations; the L2-miss traffic of the launch is
// about gn * (activation bytes) + gm * (weight bytes), provided an XCD's weight slice stays L2-resident (<= 2.5 MB) while
// its workgroups walk the K loop.  0 = not applicable (tile count too small / not divisible).
int choose_xcd_gn(const ConvSlot &c, const y3::ConvArgs &a, const y3::TileInfo &t)
{
    const int tilesN = a.CoutPad / t.bn;
    const long long tilesM = (a.M + t.bm - 1) / t.bm;
    if (tilesM * tilesN < 64) return 0;
    const double w_bytes = (double)a.CoutPad * c.K * 4.0, a_bytes = (double)a.src0_bytes + a.src1_bytes;
    int best = 0;
    double best_cost = 0;
    for (int gn = 1; gn <= 8; gn *= 2) {
        const int gm = 8 / gn;
        if (tilesN % gn || tilesM < gm) continue;
        double cost = gn * a_bytes + gm * w_bytes;
        if (w_bytes / gn > 2.5e6) cost += 8.0 * (a_bytes + w_bytes);   // weight slice does not stay in L2: last resort
        if (!best || cost < best_cost) {
            best = gn;
            best_cost = cost;
        }
    }
    return best > 1 ? best : 0;   // gn = 1 is the contiguous order (8 pixel-tile runs), which needs no padding workgroups
}

int choose_tile(const ConvSlot &c, long long M)
{
    // the first residual block's 3x3 (32 -> 64 @208): weights resident in registers, input patch by LDS-DMA (tile id 33, conv_res_f32.hip).
    // From the conv's shape alone (never the rows of the call); bit-identical to the generic tiles anyway.
    if (c.d.size == 3 && c.d.stride == 1 && c.d.src1 < 0 && c.d.cin == 32 && c.d.cout % 64 == 0 && c.cout_pad == c.d.cout) return 33;
    // measured on MI355X (tools/tune_tiles.py): many co-resident waves beat big wave tiles for the 64-cycle
    // fp32 MFMA; prefer the largest block tile that still yields >= 2 workgroups per CU
    std::vector<int> cand;
    if (c.cout_pad % 128 == 0)
        cand = {10, 11};
    else if (c.cout_pad % 64 == 0)
        cand = {11};
    else
        cand = {8};
    int best = cand.back();
    for (int t : cand) {
        y3::TileInfo s = y3::conv_tile_info(t);
        long long blocks = ((M + s.bm - 1) / s.bm) * (c.cout_pad / s.bn);
        if (blocks >= 1024) {
            best = t;
            break;
        }
    }
    return best;
}

}  // namespace

// The first two ops are conv0 (3x3/1, 3 -> 32) and conv1 (3x3/2, 32 -> 64, no shortcut, single source) reading it, nobody
// else reads conv0's output, and the plan is fp32 with every intermediate reusable: the pair runs as csrc/conv_stem.hip.
static bool stem_applicable(const y3_net *net)
{
    if ((net->dtype != Y3_DTYPE_F32 && net->dtype != Y3_DTYPE_BF16) || net->keep_all || net->image_size % 32 || net->early_ops > 0)
        return false;
    if (net->ops.size() < 2 || net->ops[0].kind != 0 || net->ops[1].kind != 0) return false;
    const ConvSlot &c0 = net->convs[net->ops[0].index], &c1 = net->convs[net->ops[1].index];
    const y3_conv_desc &a = c0.d, &b = c1.d;
    if (!c0.first_layer || a.size != 3 || a.stride != 1 || a.cout != 32 || a.residual >= 0 || a.src1 >= 0) return false;
    if (b.size != 3 || b.stride != 2 || b.cin != 32 || b.cout != 64 || b.residual >= 0 || b.src1 >= 0 || b.src0 != a.dst) return false;
    if (a.src0 != net->input_tensor) return false;
    for (int k = 0; k < 3; ++k)
        if (net->outputs[k] == a.dst || net->outputs[k] == b.dst) return false;
    for (size_t i = 2; i < net->ops.size(); ++i) {
        if (net->ops[i].kind == 0) {
            const y3_conv_desc &d = net->convs[net->ops[i].index].d;
            if (d.src0 == a.dst || d.src1 == a.dst || d.residual == a.dst) return false;
        } else {
            const y3_aux_desc &x = net->aux[net->ops[i].index];
            if (x.src0 == a.dst || x.src1 == a.dst) return false;
        }
    }
    return true;
}

// third op = 1x1 conv 64 -> 32 reading conv1's output (backbone.yaml layer 3): computed by the stem kernel from the tile it
// still holds on chip (fp32 and bf16 plans)
static bool stem_conv2_applicable(const y3_net *net)
{
    if ((net->dtype != Y3_DTYPE_F32 && net->dtype != Y3_DTYPE_BF16) || net->ops.size() < 3 || net->ops[2].kind != 0) return false;
    const y3_conv_desc &b = net->convs[net->ops[1].index].d, &c = net->convs[net->ops[2].index].d;
    if (c.size != 1 || c.stride != 1 || c.cin != 64 || c.cout != 32 || c.residual >= 0 || c.src1 >= 0 || c.src0 != b.dst) return false;
    for (int k = 0; k < 3; ++k)
        if (net->outputs[k] == c.dst) return false;
    return true;
}

// Does this conv's launch write an fp32 net output directly (bf16 / plane-split plans)?  Mirrors the `staged` rule of y3_net_plan: an
// output that another op reads, or that a shortcut / first-layer conv writes, stays in the arena in the mode's format instead.
static bool writes_f32_output(const y3_net *net, const ConvSlot &c)
{
    const int t = c.d.dst;
    if (t != net->outputs[0] && t != net->outputs[1] && t != net->outputs[2]) return false;
    if (c.d.residual >= 0 || c.first_layer) return false;
    for (const ConvSlot &o : net->convs)
        if (o.d.src0 == t || o.d.src1 == t || o.d.residual == t) return false;
    return true;
}

extern "C" {

int y3_version(void) { return 100; }

const char *y3_last_error(void) { return g_err; }

int y3_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int y3_tile_built(int dtype, int tile)
{
    switch (dtype) {
        case Y3_DTYPE_F32: return y3::conv_tile_built(tile) ? 1 : 0;
        case Y3_DTYPE_BF16: return y3::conv_bf16_tile_built(tile) ? 1 : 0;
        case Y3_DTYPE_F32X3: return (tile >= 0 && tile < y3::X3_TILE_COUNT && y3::conv_x3_tile_built(tile)) ? 1 : 0;
        case Y3_DTYPE_F32X2: return (tile >= 0 && tile < y3::X3_TILE_COUNT && y3::conv_x2_tile_built(tile)) ? 1 : 0;
        default: return 0;
    }
}

y3_status y3_net_create(const y3_tensor_desc *tensors, int n_tensors, const int32_t *op_kinds, int n_ops,
                        const y3_conv_desc *convs, int n_convs, const y3_aux_desc *aux, int n_aux, int input_tensor,
                        const int32_t outputs[3], int nclasses, y3_net **out)
try {
    if (!tensors || !op_kinds || !convs || !outputs || !out || n_tensors <= 0 || n_ops <= 0 || n_convs < 0 || n_aux < 0)
        return fail(Y3_ERR_INVALID, "y3_net_create: null, empty or negative-count argument");
    if (y3::test_fail_alloc()) throw std::bad_alloc();   // tests only: the allocation below failing
    std::unique_ptr<y3_net> net(new y3_net());           // released to the caller on success only: no path leaks it, thrown ones included
    if (hipGetDevice(&net->device) != hipSuccess) return fail(Y3_ERR_NODEVICE, "y3_net_create: no HIP device");
    net->tensors.assign(tensors, tensors + n_tensors);
    net->aux.assign(aux, aux + (aux ? n_aux : 0));
    net->convs.resize(n_convs);
    auto bad_t = [&](int t) { return t < 0 || t >= n_tensors; };
    for (int i = 0; i < n_convs; ++i) {
        ConvSlot &c = net->convs[i];
        c.d = convs[i];
        const y3_conv_desc &d = c.d;
        int err = 0;
        if (bad_t(d.src0) || bad_t(d.dst) || (d.src1 >= 0 && bad_t(d.src1)) || (d.residual >= 0 && bad_t(d.residual))) err = 1;
        if (!(d.size == 1 || d.size == 3) || !(d.stride == 1 || (d.stride == 2 && d.size == 3))) err = 2;
        if (d.src1 >= 0 && (d.size != 1 || d.c0 % 32 || (d.cin - d.c0) % 32 || d.c0 <= 0 || d.c0 >= d.cin)) err = 3;
        if (d.src1 < 0 && (d.c0 != d.cin || d.src0_upsample)) err = 4;
        c.first_layer = (d.cin == 3);
        if (c.first_layer && !(d.size == 3 && d.stride == 1 && d.cout == 32 && d.residual < 0 && d.src1 < 0)) err = 5;
        if (!c.first_layer && d.cin % 32) err = 6;
        if (d.cout <= 0 || d.out_div != d.in_div * d.stride) err = err ? err : 7;
        if (err) {
            return fail(Y3_ERR_INVALID, "y3_net_create: conv %d unsupported or inconsistent (check %d)", i, err);
        }
        c.cout_pad = (d.cout + 31) / 32 * 32;
        c.cout_pad64 = (d.cout + 63) / 64 * 64;
        c.K = d.size * d.size * d.cin;
    }
    int ci = 0, ai = 0;
    for (int i = 0; i < n_ops; ++i) {
        if (op_kinds[i] == 0) {
            if (ci >= n_convs) { return fail(Y3_ERR_INVALID, "y3_net_create: more conv ops than descriptors"); }
            net->ops.push_back({0, ci++});
        } else {
            if (ai >= n_aux) { return fail(Y3_ERR_INVALID, "y3_net_create: more aux ops than descriptors"); }
            net->ops.push_back({1, ai++});
        }
    }
    if (bad_t(input_tensor)) { return fail(Y3_ERR_INVALID, "y3_net_create: bad input tensor"); }
    net->input_tensor = input_tensor;
    for (int i = 0; i < 3; ++i) {
        // nclasses == 0: raw feature outputs (layer tests); otherwise the yolo head layout is enforced
        if (bad_t(outputs[i]) || (nclasses > 0 && tensors[outputs[i]].channels != 3 * (5 + nclasses))) {
            return fail(Y3_ERR_INVALID, "y3_net_create: output %d must have 3*(5+nclasses) channels", i);
        }
        net->outputs[i] = outputs[i];
    }
    net->nclasses = nclasses;
    net->tdev.assign(n_tensors, nullptr);
    net->tbytes.assign(n_tensors, 0);
    net->tblock.assign(n_tensors, 0);
    *out = net.release();
    return Y3_OK;
}
Y3_CATCH("y3_net_create")

void y3_net_destroy(y3_net *net)
{
    if (!net) return;
    free_plan(net);
    if (net->fork_ev) {
        (void)hipEventDestroy(net->fork_ev);
        for (int i = 0; i < Y3_MAX_LANES; ++i) {
            (void)hipStreamDestroy(net->lane_stream[i]);
            (void)hipEventDestroy(net->join_ev[i]);
        }
    }
    for (ConvSlot &c : net->convs) {
        if (c.w_dev) (void)hipFree(c.w_dev);
        if (c.w0stem_dev) (void)hipFree(c.w0stem_dev);
        if (c.w0raw_dev) (void)hipFree(c.w0raw_dev);
        if (c.wbf_dev) (void)hipFree(c.wbf_dev);
        if (c.wx3_dev) (void)hipFree(c.wx3_dev);
        if (c.wx2_dev) (void)hipFree(c.wx2_dev);
        if (c.scale_dev) (void)hipFree(c.scale_dev);
        if (c.shift_dev) (void)hipFree(c.shift_dev);
    }
    delete net;
}

y3_status y3_net_set_conv_weights(y3_net *net, int slot, const float *w, const float *gamma, const float *beta,
                                  const float *mean, const float *var, const float *bias, float eps)
try {
    if (!net || slot < 0 || slot >= (int)net->convs.size() || !w)
        return fail(Y3_ERR_INVALID, "y3_net_set_conv_weights: bad slot or null weights");
    ConvSlot &c = net->convs[slot];
    const y3_conv_desc &d = c.d;
    if (d.bn ? !(gamma && beta && mean && var) : !bias)
        return fail(Y3_ERR_INVALID, "y3_net_set_conv_weights: conv %d needs %s", slot, d.bn ? "gamma/beta/mean/var" : "bias");
    const int K = c.K, CP = c.cout_pad, CP64 = c.cout_pad64;
    std::vector<float> scale(CP64, 1.0f), shift(CP64, 0.0f);
    for (int n = 0; n < d.cout; ++n) {
        if (d.bn) {
            // BatchNormalization inference: y = x*scale + (beta - mean*scale), scale = gamma*rsqrt(var+eps)
            const float inv = 1.0f / sqrtf(var[n] + eps);
            scale[n] = inv * gamma[n];
            shift[n] = beta[n] - mean[n] * scale[n];
        } else {
            shift[n] = bias[n];
        }
    }
    std::vector<float> pk;
    if (c.first_layer) {
        pk.assign(w, w + (size_t)K * d.cout);  // HWIO as is
    } else {
        pk.assign((size_t)CP * K, 0.0f);       // [CoutPad][K], k = tap*Cin + c  (HWIO is [K][Cout])
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < d.cout; ++n) pk[(size_t)n * K + k] = w[(size_t)k * d.cout + n];
    }
    // fp32 path: the BN scale is folded into the packed weights (one VALU multiply less per output element; VALU
    // time is matrix-pipe time for the fp32 MFMA).  The bf16 copy keeps the unscaled weights + scale in the epilogue.
    std::vector<float> pk_scaled;
    if (!c.first_layer) {
        pk_scaled = pk;
        for (int n = 0; n < d.cout; ++n)
            for (int k = 0; k < K; ++k) pk_scaled[(size_t)n * K + k] *= scale[n];
    }
    Y3_ENTER_DEVICE(net);
    if (!c.w_dev) HIP_TRY(hipMalloc(&c.w_dev, pk.size() * sizeof(float)));
    if (!c.scale_dev) HIP_TRY(hipMalloc(&c.scale_dev, CP64 * sizeof(float)));
    if (!c.shift_dev) HIP_TRY(hipMalloc(&c.shift_dev, CP64 * sizeof(float)));
    HIP_TRY(hipMemcpy(c.w_dev, (c.first_layer ? pk : pk_scaled).data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c.first_layer) {
        // fused stem kernel: conv0 on the matrix cores wants K = 27 padded to 28 and the BN scale folded in
        std::vector<float> w28((size_t)28 * d.cout, 0.0f);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < d.cout; ++n) w28[(size_t)k * d.cout + n] = w[(size_t)k * d.cout + n] * scale[n];
        if (!c.w0stem_dev) HIP_TRY(hipMalloc(&c.w0stem_dev, w28.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c.w0stem_dev, w28.data(), w28.size() * sizeof(float), hipMemcpyHostToDevice));
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < d.cout; ++n) w28[(size_t)k * d.cout + n] = w[(size_t)k * d.cout + n];
        if (!c.w0raw_dev) HIP_TRY(hipMalloc(&c.w0raw_dev, w28.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c.w0raw_dev, w28.data(), w28.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!c.first_layer) {
        std::vector<unsigned short> pb(pk.size());
        for (size_t i = 0; i < pk.size(); ++i) pb[i] = f32_to_bf16_rne(pk[i]);
        if (!c.wbf_dev) HIP_TRY(hipMalloc(&c.wbf_dev, pb.size() * sizeof(unsigned short)));
        HIP_TRY(hipMemcpy(c.wbf_dev, pb.data(), pb.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(c.scale_dev, scale.data(), CP64 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.shift_dev, shift.data(), CP64 * sizeof(float), hipMemcpyHostToDevice));
    if (!c.first_layer) {
        // three-plane split of the (unscaled) weights: x = hi + mid + lo exactly
        std::vector<unsigned short> px((size_t)CP64 * 3 * K, 0);
        for (int n = 0; n < d.cout; ++n)
            for (int k = 0; k < K; ++k) {
                const float x = pk[(size_t)n * K + k];
                const unsigned short h = f32_to_bf16_rne(x);
                const float r1 = x - bf16_to_f32(h);
                const unsigned short m = f32_to_bf16_rne(r1);
                const float r2 = r1 - bf16_to_f32(m);
                px[((size_t)n * 3 + 0) * K + k] = h;
                px[((size_t)n * 3 + 1) * K + k] = m;
                px[((size_t)n * 3 + 2) * K + k] = f32_to_bf16_rne(r2);
            }
        if (!c.wx3_dev) HIP_TRY(hipMalloc(&c.wx3_dev, px.size() * sizeof(unsigned short)));
        HIP_TRY(hipMemcpy(c.wx3_dev, px.data(), px.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        // two fp16 planes of the BN-scaled weights: w = h + l' * 2^-11 (up to 2^-22 |w|); |w| must stay below 65504
        std::vector<unsigned short> p2((size_t)CP64 * 2 * K, 0);
        c.x2_ok = true;
        for (int n = 0; n < d.cout; ++n)
            for (int k = 0; k < K; ++k) {
                const float x = pk_scaled[(size_t)n * K + k];
                if (!(fabsf(x) < 65504.0f)) c.x2_ok = false;   // reported by y3_net_plan(Y3_DTYPE_F32X2); other modes are unaffected
                const unsigned short h = f32_to_f16_rne(x);
                p2[((size_t)n * 2 + 0) * K + k] = h;
                p2[((size_t)n * 2 + 1) * K + k] = f32_to_f16_rne((x - f16_to_f32(h)) * 2048.0f);
            }
        if (!c.wx2_dev) HIP_TRY(hipMalloc(&c.wx2_dev, p2.size() * sizeof(unsigned short)));
        HIP_TRY(hipMemcpy(c.wx2_dev, p2.data(), p2.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    c.loaded = true;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_conv_weights")

y3_status y3_net_set_tile(y3_net *net, int slot, int tile)
try {
    if (!net || slot < 0 || slot >= (int)net->convs.size() || tile >= y3::TILE_COUNT)
        return fail(Y3_ERR_INVALID, "y3_net_set_tile: bad argument");
    ConvSlot &c = net->convs[slot];
    if (tile >= 0) {
        if (!y3::conv_tile_built(tile))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile: tile id %d is retired (the timing ablations of rounds 1-2; y3_tile_built)", tile);
        y3::TileInfo s = y3::conv_tile_info(tile);
        if (c.first_layer || c.cout_pad % s.bn) return fail(Y3_ERR_INVALID, "y3_net_set_tile: tile does not divide Cout");
        if (tile == 33 && !(c.d.size == 3 && c.d.stride == 1 && c.d.src1 < 0 && c.d.cin == 32 && c.d.cout % 64 == 0))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile: tile 33 (weight-resident) needs a 3x3 / stride-1 conv with 32 input channels and Cout %% 64 == 0");
    }
    c.tile = tile;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_tile")

y3_status y3_net_set_tile_bf16(y3_net *net, int slot, int tile)
try {
    if (!net || slot < 0 || slot >= (int)net->convs.size() || tile >= y3::BF16_TILE_COUNT)
        return fail(Y3_ERR_INVALID, "y3_net_set_tile_bf16: bad argument");
    ConvSlot &c = net->convs[slot];
    if (tile >= 0) {
        if (!y3::conv_bf16_tile_built(tile))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_bf16: tile id %d is retired (20: the pipelined tile of round 2; 33..36: tap-row reuse and the four-wave tile of round 4; y3_tile_built)", tile);
        y3::TileInfo s = y3::conv_bf16_tile_info(tile);
        if (c.first_layer || c.cout_pad % s.bn || c.d.cin % s.stages || (c.d.src1 >= 0 && c.d.c0 % s.stages))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_bf16: tile does not fit this conv");
        if (tile == 32 && !(c.d.size == 3 && c.d.stride == 1 && c.d.src1 < 0 && (c.d.cin == 32 || c.d.cin == 64) && c.d.cout % 64 == 0))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_bf16: tile 32 (weight-resident) needs a 3x3 / stride-1 conv with 32 or 64 input channels and Cout %% 64 == 0");
        // tile 32 stores bf16 only: a conv whose destination is a net output that the forward hands over as fp32 straight from the launch
        // (not read again inside the net, no shortcut: y3_net_plan does not stage it) cannot take it -- refused here, by name, instead of a
        // launch error in the forward
        if (tile == 32 && writes_f32_output(net, c))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_bf16: tile 32 (weight-resident) stores bf16 only; conv %d writes an fp32 net output", slot);
    }
    c.tile_bf16 = tile;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_tile_bf16")

y3_status y3_net_set_tile_x3(y3_net *net, int slot, int tile)
try {
    if (!net || slot < 0 || slot >= (int)net->convs.size() || tile >= y3::X3_TILE_COUNT || (tile >= 0 && !y3::conv_x3_tile_built(tile)))
        return fail(Y3_ERR_INVALID, "y3_net_set_tile_x3: bad argument");
    ConvSlot &c = net->convs[slot];
    if (tile >= 0) {
        y3::TileInfo s = y3::conv_x3_tile_info(tile);
        if (c.first_layer || c.cout_pad64 % s.bn || c.d.cin % s.stages || (c.d.src1 >= 0 && c.d.c0 % s.stages))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_x3: tile does not fit this conv");
    }
    c.tile_x3 = tile;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_tile_x3")

y3_status y3_net_set_tile_x2(y3_net *net, int slot, int tile)
try {
    if (!net || slot < 0 || slot >= (int)net->convs.size() || tile >= y3::X3_TILE_COUNT)
        return fail(Y3_ERR_INVALID, "y3_net_set_tile_x2: bad argument");
    ConvSlot &c = net->convs[slot];
    if (tile >= 0) {
        y3::TileInfo s = y3::conv_x3_tile_info(tile);
        if (!y3::conv_x2_tile_built(tile) || c.first_layer || c.cout_pad64 % s.bn || c.d.cin % s.stages ||
            (c.d.src1 >= 0 && c.d.c0 % s.stages))
            return fail(Y3_ERR_INVALID, "y3_net_set_tile_x2: tile does not fit this conv");
    }
    c.tile_x2 = tile;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_tile_x2")

y3_status y3_net_set_lanes(y3_net *net, int lanes)
try {
    if (!net || lanes < 1 || lanes > Y3_MAX_LANES) return fail(Y3_ERR_INVALID, "y3_net_set_lanes: lanes must be in [1,%d]", Y3_MAX_LANES);
    net->lanes = lanes;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_lanes")

y3_status y3_net_set_stem_fusion(y3_net *net, int on)
try {
    if (!net || on < 0 || on > 2) return fail(Y3_ERR_INVALID, "y3_net_set_stem_fusion: argument must be 0, 1 or 2");
    net->stem_mode = on;
    net->stem_mode_set = true;
    // takes effect at once on a planned net when the graph qualifies (decided again by the next y3_net_plan)
    if (net->image_size) {
        net->stem_fused = on && stem_applicable(net);
        net->stem_conv2 = net->stem_fused && on == 1 && stem_conv2_applicable(net);
    }
    return Y3_OK;
}
Y3_CATCH("y3_net_set_stem_fusion")

y3_status y3_net_set_k_chunk(y3_net *net, int channels)
try {
    if (!net || channels < -1 || (channels > 0 && channels % 32)) return fail(Y3_ERR_INVALID, "y3_net_set_k_chunk: -1, 0 or a multiple of 32 channels");
    net->k_chunk = channels;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_k_chunk")

y3_status y3_net_set_xcd_mode(y3_net *net, int mode)
try {
    if (!net || mode < 0 || mode > 1) return fail(Y3_ERR_INVALID, "y3_net_set_xcd_mode: mode must be 0 or 1");
    net->xcd_mode = mode;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_xcd_mode")

y3_status y3_net_set_early_chunk(y3_net *net, int n_convs, int chunk_images)
try {
    if (!net || n_convs < 0 || chunk_images < 0) return fail(Y3_ERR_INVALID, "y3_net_set_early_chunk: bad argument");
    if (n_convs >= (int)net->convs.size()) return fail(Y3_ERR_INVALID, "y3_net_set_early_chunk: n_convs must leave at least one conv for the full batch");
    net->early_convs = (chunk_images > 0) ? n_convs : 0;
    net->early_chunk = (n_convs > 0) ? chunk_images : 0;
    return Y3_OK;
}
Y3_CATCH("y3_net_set_early_chunk")

y3_status y3_net_keep_activations(y3_net *net, int keep)
try {
    if (!net) return fail(Y3_ERR_INVALID, "y3_net_keep_activations: null net");
    net->keep_all = keep ? 1 : 0;
    return Y3_OK;
}
Y3_CATCH("y3_net_keep_activations")

static void detect_layout(const y3_net *net, int batch, size_t off[9], size_t *n_boxes, int32_t gs[3], size_t gelems[3]);

// forked streams / events of the concurrent sub-batches: created at plan time so that a forward enqueues work only
static y3_status ensure_lanes(y3_net *net)
{
    if (net->fork_ev) return Y3_OK;
    HIP_TRY(hipEventCreateWithFlags(&net->fork_ev, hipEventDisableTiming));
    for (int i = 0; i < Y3_MAX_LANES; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&net->lane_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&net->join_ev[i], hipEventDisableTiming));
    }
    return Y3_OK;
}

y3_status y3_net_plan(y3_net *net, int max_batch, int image_size, int dtype)
try {
    if (!net || max_batch <= 0 || image_size <= 0) return fail(Y3_ERR_INVALID, "y3_net_plan: bad argument");
    if (dtype != Y3_DTYPE_F32 && dtype != Y3_DTYPE_BF16 && dtype != Y3_DTYPE_F32X3 && dtype != Y3_DTYPE_F32X2)
        return fail(Y3_ERR_INVALID, "y3_net_plan: unknown dtype %d", dtype);
    for (const y3_tensor_desc &t : net->tensors)
        if (t.div <= 0 || image_size % t.div) return fail(Y3_ERR_INVALID, "y3_net_plan: image_size %d not divisible by %d", image_size, t.div);
    if (dtype == Y3_DTYPE_F32X2)
        for (size_t i = 0; i < net->convs.size(); ++i)
            if (net->convs[i].loaded && !net->convs[i].x2_ok)
                return fail(Y3_ERR_INVALID, "y3_net_plan: conv %zu has a BN-scaled weight outside the fp16 range (|w| >= 65504); "
                                            "the two-plane mode cannot represent it, use Y3_DTYPE_F32 or Y3_DTYPE_F32X3", i);
    Y3_ENTER_DEVICE(net);
    if (net->n_cus <= 0) {   // once per net: the launch path itself makes no device query
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, net->device));
        net->n_cus = cus > 0 ? cus : 256;
    }
    free_plan(net);
    net->max_batch = max_batch;
    net->image_size = image_size;
    net->dtype = dtype;
    const int nt = (int)net->tensors.size();
    // liveness over the op list; tensors with equal lifetime class share blocks (first-fit free list)
    std::vector<int> first(nt, -1), last(nt, -1);
    auto touch = [&](int t, int i) {
        if (t < 0) return;
        if (first[t] < 0) first[t] = i;
        last[t] = i;
    };
    for (int i = 0; i < (int)net->ops.size(); ++i) {
        const Op &o = net->ops[i];
        if (o.kind == 0) {
            const y3_conv_desc &d = net->convs[o.index].d;
            touch(d.src0, i); touch(d.src1, i); touch(d.residual, i); touch(d.dst, i);
        } else {
            const y3_aux_desc &a = net->aux[o.index];
            touch(a.src0, i); touch(a.src1, i); touch(a.dst, i);
        }
    }
    net->staged.assign(nt, 0);
    if (dtype != Y3_DTYPE_F32) {
        for (int i = 0; i < (int)net->ops.size(); ++i) {
            const Op &o = net->ops[i];
            if (o.kind != 0) continue;
            const ConvSlot &c = net->convs[o.index];
            const y3_conv_desc &d = c.d;
            for (int k = 0; k < 3; ++k) {
                const int out = net->outputs[k];
                if (d.src0 == out || d.src1 == out || d.residual == out) net->staged[out] = 1;     // read again inside the net
                if (d.dst == out && (d.residual >= 0 || c.first_layer)) net->staged[out] = 1;       // no fp32-output form of that launch
            }
        }
        for (int k = 0; k < 3; ++k)
            if (net->staged[net->outputs[k]]) last[net->outputs[k]] = (int)net->ops.size();        // alive until the final conversion
    }
    // chunked leading segment: every op before the (early_convs)-th conv; tensors it writes get blocks of their own,
    // laid out densely by image, because they are rewritten chunk after chunk while earlier chunks' results are still live
    net->early_ops = 0;
    if (net->early_convs > 0 && net->early_chunk > 0) {
        int seen = 0;
        for (int i = 0; i < (int)net->ops.size(); ++i) {
            if (net->ops[i].kind == 0 && seen++ == net->early_convs) break;
            net->early_ops = i + 1;
        }
        if (net->early_ops >= (int)net->ops.size()) net->early_ops = 0;
    }
    net->dense.assign(nt, 0);
    for (int t = 0; t < nt; ++t) net->dense[t] = (first[t] >= 0 && first[t] < net->early_ops) ? 1 : 0;
    for (int t = 0; t < nt; ++t) {
        const int s = image_size / net->tensors[t].div;
        net->tbytes[t] = (size_t)max_batch * s * s * net->tensors[t].channels * (dtype == Y3_DTYPE_BF16 ? 2 : dtype == Y3_DTYPE_F32X3 ? 6 : 4);   // two fp16 planes: 4 bytes as well
        if (net->tbytes[t] >= 0xFFFFFFF0ull && first[t] >= 0)
            return fail(Y3_ERR_INVALID, "y3_net_plan: tensor %d is %zu bytes; 32-bit buffer offsets need < 4 GiB, lower max_batch", t, net->tbytes[t]);
    }
    struct Blk { void *p; size_t bytes; int free_at; };
    std::vector<Blk> pool;
    // allocate in order of first definition; the image batch and the head grids are caller-owned
    std::vector<int> order;
    for (int t = 0; t < nt; ++t) {
        const bool external = (t == net->input_tensor ||
                               ((t == net->outputs[0] || t == net->outputs[1] || t == net->outputs[2]) && !net->staged[t]));
        if (first[t] >= 0 && !external) order.push_back(t);
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
    for (int t : order) {
        const int until = (net->keep_all || net->dense[t]) ? (int)net->ops.size() + 1 : last[t];
        int pick = -1;
        for (int k = 0; k < (int)pool.size() && !net->dense[t]; ++k)
            if (pool[k].free_at < first[t] && pool[k].bytes >= net->tbytes[t] &&
                (pick < 0 || pool[k].bytes < pool[pick].bytes))
                pick = k;
        if (pick < 0) {
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, net->tbytes[t] + 4096);
            if (e != hipSuccess) {
                free_plan(net);
                return fail(Y3_ERR_OOM, "y3_net_plan: hipMalloc(%zu) failed: %s", net->tbytes[t], hipGetErrorString(e));
            }
            net->blocks.push_back(p);
            pool.push_back({p, net->tbytes[t], until});
            pick = (int)pool.size() - 1;
        }
        pool[pick].free_at = until;
        net->tdev[t] = pool[pick].p;
        net->tblock[t] = pool[pick].bytes;
    }
    if (y3_status st = ensure_lanes(net); st != Y3_OK) return st;
    {   // Y3_STEM_MODE (tools: same-process-tree A/B of the stem forms) overrides the default, not an explicit setter call
        static const int env = [] { const char *e = getenv("Y3_STEM_MODE"); return e ? atoi(e) : -1; }();
        if (env >= 0 && env <= 2 && !net->stem_mode_set) net->stem_mode = env;
    }
    net->stem_fused = net->stem_mode && stem_applicable(net);
    net->stem_conv2 = net->stem_fused && net->stem_mode == 1 && stem_conv2_applicable(net);
    if (net->nclasses > 0) {   // scratch of y3_net_detect: no allocation inside the stream-ordered call
        size_t off[9], n_boxes, gelems[3];
        int32_t gs[3];
        detect_layout(net, max_batch, off, &n_boxes, gs, gelems);
        hipError_t e = hipMalloc(&net->det_buf, off[8]);
        if (e != hipSuccess) {
            free_plan(net);
            return fail(Y3_ERR_OOM, "y3_net_plan: hipMalloc(%zu) for the detect scratch failed: %s", off[8], hipGetErrorString(e));
        }
        net->det_bytes = off[8];
    }
    return Y3_OK;
}
Y3_CATCH("y3_net_plan")

// byte offsets of the y3_net_detect scratch for `batch` images (and the total in [8])
static void detect_layout(const y3_net *net, int batch, size_t off[9], size_t *n_boxes, int32_t gs[3], size_t gelems[3])
{
    const size_t per = (size_t)3 * (5 + net->nclasses);
    size_t n = 0;
    for (int i = 0; i < 3; ++i) {
        gs[i] = spatial(net, net->outputs[i]);
        gelems[i] = (size_t)batch * gs[i] * gs[i] * per;
        n += (size_t)3 * gs[i] * gs[i];
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    off[0] = 0;                                              // grid 0
    off[1] = off[0] + up(gelems[0] * 4);                     // grid 1
    off[2] = off[1] + up(gelems[1] * 4);                     // grid 2
    off[3] = off[2] + up(gelems[2] * 4);                     // boxes
    off[4] = off[3] + up((size_t)batch * n * 16);            // class indices (i64)
    off[5] = off[4] + up((size_t)batch * n * 8);             // scores
    off[6] = off[5] + up((size_t)batch * n * 4);             // selected indices
    off[7] = off[6] + up((size_t)batch * Y3_MAX_OUTPUT_BOXES * 4);   // NMS workspace
    off[8] = off[7] + y3::nms_workspace_bytes(batch, (int)n);
    *n_boxes = n;
}

double y3_net_flops_per_image(const y3_net *net)
{
    if (!net || !net->image_size) return 0.0;
    double tot = 0;
    for (const ConvSlot &c : net->convs) {
        const double ho = net->image_size / c.d.out_div;
        tot += 2.0 * c.d.size * c.d.size * c.d.cin * c.d.cout * ho * ho;
    }
    return tot;
}

// Enqueue the whole op list for images [b0, b0+nb) of the batch on stream s (tensor pointers offset by b0 images).
// op_begin/op_end select a segment of the op list (-1: to the end).
static y3_status run_slice(y3_net *net, const float *images, float *const grids[3], int b0, int nb, hipStream_t s,
                           float *ms_out, int n_ms, int lane = 0, int /*lanes*/ = 1, int op_begin = 0, int op_end = -1)
{
    if (op_end < 0) op_end = (int)net->ops.size();
    auto img_elems = [&](int t) -> size_t {
        const int sp = spatial(net, t);
        return (size_t)sp * sp * net->tensors[t].channels;
    };
    const bool bf = net->dtype == Y3_DTYPE_BF16;
    const bool x3 = net->dtype == Y3_DTYPE_F32X3;
    const bool x2 = net->dtype == Y3_DTYPE_F32X2;
    const size_t asz = bf ? 2 : x3 ? 6 : 4;   // bytes per element of an arena tensor (fp32, or 2 x fp16)
    auto is_out = [&](int t) { return (t == net->outputs[0] || t == net->outputs[1] || t == net->outputs[2]) && !net->staged[t]; };
    // element size: head grids are always fp32; the image batch is fp32 when the Cin = 3 first-layer kernel reads it
    // (a model whose input feeds an MFMA conv directly hands bf16 in bf16 mode); everything else follows the plan
    auto esz = [&](int t) -> size_t {
        if (is_out(t)) return 4;
        if (t == net->input_tensor) return net->tensors[t].channels != 3 ? asz : 4;
        return asz;
    };
    auto ptr = [&](int t) -> void * {
        if (t < 0) return nullptr;
        char *base = nullptr;
        if (t == net->input_tensor) base = reinterpret_cast<char *>(const_cast<float *>(images));
        for (int i = 0; i < 3 && !base; ++i)
            if (t == net->outputs[i] && !net->staged[t]) base = reinterpret_cast<char *>(grids[i]);
        if (base) return base + (size_t)b0 * img_elems(t) * esz(t);
        // arena tensors share blocks with other (dead) tensors of different per-image size: give every lane its
        // own 1/lanes region of the block so that concurrent sub-batches never alias
        char *blk = static_cast<char *>(net->tdev[t]);
        if (!blk) return nullptr;
        if (net->dense[t]) return blk + (size_t)b0 * img_elems(t) * esz(t);
        // lane regions start at the lane's first image (scaled to the block size), 256-B aligned; blocks carry 4 KiB of slack
        const size_t off = ((size_t)((double)net->tblock[t] * b0 / net->cur_batch) + 255) & ~(size_t)255;
        return blk + (lane ? off : 0);
    };
    auto bytes = [&](int t) -> size_t { return (size_t)nb * img_elems(t) * esz(t); };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ms_out) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
    }
    for (int oi = op_begin; oi < op_end; ++oi) {
        const Op &o = net->ops[oi];
        if (o.kind == 0) {
            ConvSlot &c = net->convs[o.index];
            const y3_conv_desc &d = c.d;
            y3::ConvArgs a{};
            a.src0 = ptr(d.src0);
            a.src1 = ptr(d.src1);
            a.wpk = c.w_dev;
            a.scale = c.scale_dev;
            a.shift = c.shift_dev;
            a.residual = ptr(d.residual);
            a.dst = ptr(d.dst);
            a.B = nb;
            a.H = a.W = net->image_size / d.in_div;
            a.Ho = a.Wo = net->image_size / d.out_div;
            a.Cin = d.cin;
            a.C0 = d.c0;
            a.Cout = d.cout;
            a.CoutPad = c.cout_pad;
            a.ksize = d.size;
            a.stride = d.stride;
            a.pad = (d.size == 3) ? 1 : 0;
            a.up0 = d.src0_upsample;
            a.leaky = d.leaky;
            a.M = nb * a.Ho * a.Wo;
            a.K = c.K;
            a.src0_bytes = (unsigned)bytes(d.src0);
            a.src1_bytes = d.src1 >= 0 ? (unsigned)bytes(d.src1) : 0;
            a.w_bytes = (unsigned)((size_t)c.cout_pad * c.K * sizeof(float));
            a.dst_bytes = (unsigned)bytes(d.dst);
            a.xcd_gn = 0;
            a.k_chunk = 0;
            a.n_cus = net->n_cus;
            a.device = net->device;
            a.clk_stamps = (net->clk_conv == o.index) ? net->clk_stamps : (net->clk_conv == -2 && net->clk_stamps) ? net->clk_stamps + 8 * o.index : nullptr;   // fp32 MFMA kernel and stem only
            int head = -1;   // fused decode: which output this conv produces (its grid is then not written)
            if (net->fuse)
                for (int k = 0; k < 3; ++k)
                    if (d.dst == net->outputs[k] && !net->staged[d.dst]) head = k;
            if (head >= 0) {
                a.dec = net->fuse[head];
                a.dec.boxes += (size_t)b0 * a.dec.N * 4;
                a.dec.cls += (size_t)b0 * a.dec.N;
                a.dec.scores += (size_t)b0 * a.dec.N;
                a.dst = nullptr;
                a.dst_bytes = 0;
            }
            if (!a.src0 || (!a.dst && head < 0)) return fail(Y3_ERR_STATE, "conv %d: tensor not planned", o.index);
            if ((net->stem_fused && oi == 0) || (net->stem_conv2 && oi == 2)) {   // runs inside conv1's launch (fused stem)
                if (ms_out && o.index < n_ms) ms_out[o.index] = 0.0f;
                continue;
            }
            if (ms_out) HIP_TRY(hipEventRecord(ev0, s));
            hipError_t e;
            if (net->stem_fused && oi == 1) {
                const ConvSlot &c0 = net->convs[net->ops[0].index];
                y3::StemArgs sa{};
                sa.img = static_cast<const float *>(ptr(c0.d.src0));
                sa.w0 = bf ? c0.w0raw_dev : c0.w0stem_dev;
                sa.scale0 = c0.scale_dev;
                sa.shift0 = c0.shift_dev;
                sa.w1 = bf ? c.wbf_dev : static_cast<const void *>(c.w_dev);
                sa.scale1 = c.scale_dev;
                sa.shift1 = c.shift_dev;
                sa.dst = a.dst;
                sa.B = nb;
                sa.S = net->image_size;
                sa.leaky0 = c0.d.leaky;
                sa.leaky1 = d.leaky;
                sa.img_bytes = (unsigned)bytes(c0.d.src0);
                sa.dst_bytes = a.dst_bytes;
                sa.device = net->device;
                sa.n_cus = net->n_cus;
                sa.clk_stamps = (net->clk_conv == o.index) ? net->clk_stamps : (net->clk_conv == -2 && net->clk_stamps) ? net->clk_stamps + 8 * o.index : nullptr;
                if (net->stem_conv2) {
                    const ConvSlot &c2 = net->convs[net->ops[2].index];
                    sa.w2 = bf ? c2.wbf_dev : static_cast<const void *>(c2.w_dev);
                    sa.scale2 = c2.scale_dev;
                    sa.shift2 = c2.shift_dev;
                    sa.dst2 = ptr(c2.d.dst);
                    sa.leaky2 = c2.d.leaky;
                    sa.dst2_bytes = (unsigned)bytes(c2.d.dst);
                    if (!sa.dst2) return fail(Y3_ERR_STATE, "conv %d: tensor not planned", net->ops[2].index);
                }
                e = bf ? y3::launch_conv_stem_bf16(sa, s) : y3::launch_conv_stem_f32(sa, s);
            } else if ((bf || x3 || x2) && c.first_layer) {
                if (is_out(d.dst)) return fail(Y3_ERR_INVALID, "conv %d: first layer cannot be a head in this mode", o.index);
                e = bf ? y3::launch_conv_first_bf16(a, c.w_dev, s)
                       : x3 ? y3::launch_conv_first_f32x3(a, c.w_dev, s) : y3::launch_conv_first_f32x2(a, c.w_dev, s);
            } else if (x2) {
                a.wpk = c.wx2_dev;
                a.CoutPad = c.cout_pad64;
                a.w_bytes = (unsigned)((size_t)c.cout_pad64 * 2 * c.K * 2);
                const bool out_f32 = is_out(d.dst);
                if (d.residual >= 0 && out_f32) return fail(Y3_ERR_INVALID, "conv %d: residual on a head output is not supported in this mode", o.index);
                const int tile = c.tile_x2 >= 0 ? c.tile_x2 : choose_tile_x2(c, a.M);
                e = y3::launch_conv_f32x2(a, tile, out_f32, s);
            } else if (x3) {
                a.wpk = c.wx3_dev;
                a.CoutPad = c.cout_pad64;
                a.w_bytes = (unsigned)((size_t)c.cout_pad64 * 3 * c.K * 2);
                const bool out_f32 = is_out(d.dst);
                if (d.residual >= 0 && out_f32) return fail(Y3_ERR_INVALID, "conv %d: residual on a head output is not supported in this mode", o.index);
                const int tile = c.tile_x3 >= 0 ? c.tile_x3 : choose_tile_x3(c, a.M);
                e = y3::launch_conv_f32x3(a, tile, out_f32, s);
            } else if (bf) {
                a.wpk = c.wbf_dev;
                a.w_bytes = (unsigned)((size_t)c.cout_pad * c.K * 2);
                const bool out_f32 = is_out(d.dst);
                if (d.residual >= 0 && out_f32) return fail(Y3_ERR_INVALID, "conv %d: residual on a head output is not supported in bf16 mode", o.index);
                int tile = c.tile_bf16 >= 0 ? c.tile_bf16 : choose_tile_bf16(c, a.M, (long long)net->max_batch * a.Ho * a.Wo, !out_f32);
                if (head >= 0 && y3::conv_bf16_tile_info(tile).bn != 256) {   // a box's logits must meet in one workgroup: all 256 channels in the tile
                    const bool m16 = tile >= 24 && tile <= 29;                // keep the MFMA shape of the plan's tile: same K grouping, same bits
                    const bool big = (a.M + 255) / 256 >= 256;                // 256x256 once it fills the chip, else 128x256 (16 waves both)
                    tile = m16 ? (big ? 24 : 26) : (big ? 17 : 19);
                }
                e = y3::launch_conv_bf16(a, tile, out_f32, s);
            } else if (c.first_layer) {
                e = y3::launch_conv_first_f32(a, c.w_dev, s);
            } else if (head >= 0) {
                e = y3::launch_conv_head_decode_f32(a, s);
            } else {
                const int tile = c.tile >= 0 ? c.tile : choose_tile(c, a.M);
                if (net->xcd_mode) a.xcd_gn = choose_xcd_gn(c, a, y3::conv_tile_info(tile));
                {   // K order of the 3x3 convs (conv_f32.hip): chunk-major when the conv has more input channels than one chunk
                    const int ck = net->k_chunk >= 0 ? net->k_chunk : default_k_chunk(c);
                    if (d.size == 3 && d.src1 < 0 && ck > 0 && d.cin > ck && d.cin % ck == 0 && ck % 32 == 0) a.k_chunk = ck;
                }
                e = y3::launch_conv_f32(a, tile, s);
            }
            if (e != hipSuccess) return fail(Y3_ERR_HIP, "conv %d launch: %s", o.index, hipGetErrorString(e));
            if (ms_out) {
                HIP_TRY(hipEventRecord(ev1, s));
                HIP_TRY(hipEventSynchronize(ev1));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
                if (o.index < n_ms) ms_out[o.index] = ms;
            }
        } else {
            if (bf || x3 || x2) return fail(Y3_ERR_INVALID, "stand-alone add/upsample/concat ops are fp32 only");
            const y3_aux_desc &x = net->aux[o.index];
            const int sp = spatial(net, x.dst);
            const int C = net->tensors[x.dst].channels;
            hipError_t e = hipSuccess;
            if (x.kind == Y3_AUX_ADD)
                e = y3::launch_add((const float *)ptr(x.src0), (const float *)ptr(x.src1), (float *)ptr(x.dst),
                                   (size_t)nb * sp * sp * C, s);
            else if (x.kind == Y3_AUX_UPSAMPLE2X)
                e = y3::launch_upsample2x((const float *)ptr(x.src0), nb, sp / 2, sp / 2, C, (float *)ptr(x.dst), s);
            else if (x.kind == Y3_AUX_CONCAT)
                e = y3::launch_concat((const float *)ptr(x.src0), net->tensors[x.src0].channels,
                                      (const float *)ptr(x.src1), net->tensors[x.src1].channels,
                                      (size_t)nb * sp * sp, (float *)ptr(x.dst), s);
            else
                return fail(Y3_ERR_INVALID, "unknown aux op kind %d", x.kind);
            if (e != hipSuccess) return fail(Y3_ERR_HIP, "aux op %d launch: %s", o.index, hipGetErrorString(e));
        }
    }
    if (op_end == (int)net->ops.size()) {
        for (int k = 0; k < 3; ++k) {
            const int t = net->outputs[k];
            if (!net->staged[t]) continue;
            float *dst = grids[k] + (size_t)b0 * img_elems(t);
            const int sp = spatial(net, t), C = net->tensors[t].channels;
            const size_t npix = (size_t)nb * sp * sp;
            hipError_t e = x2 ? y3::launch_x2_to_f32(ptr(t), dst, npix, C, s)
                         : x3 ? y3::launch_x3_to_f32(ptr(t), dst, npix, C, s)
                              : y3::launch_bf16_to_f32(ptr(t), dst, npix * C, s);
            if (e != hipSuccess) return fail(Y3_ERR_HIP, "output %d conversion: %s", k, hipGetErrorString(e));
        }
    }
    if (ms_out) {
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
    return Y3_OK;
}

static y3_status run(y3_net *net, const float *images, int batch, float *const grids[3], hipStream_t s,
                     float *ms_out, int n_ms)
{
    if (!net || !images || !grids || batch <= 0) return fail(Y3_ERR_INVALID, "y3_net_forward: bad argument");
    if (!net->image_size) return fail(Y3_ERR_STATE, "y3_net_forward: call y3_net_plan first");
    if (batch > net->max_batch) return fail(Y3_ERR_INVALID, "y3_net_forward: batch %d > planned %d", batch, net->max_batch);
    for (size_t i = 0; i < net->convs.size(); ++i)
        if (!net->convs[i].loaded) return fail(Y3_ERR_STATE, "y3_net_forward: conv %zu has no weights", i);
    if (net->dtype == Y3_DTYPE_F32X2)
        for (size_t i = 0; i < net->convs.size(); ++i)
            if (!net->convs[i].x2_ok) return fail(Y3_ERR_INVALID, "y3_net_forward: conv %zu has a weight outside the fp16 range of the two-plane mode", i);
    for (int i = 0; i < 3; ++i)
        if (!grids[i] || ((uintptr_t)grids[i] & 15)) return fail(Y3_ERR_INVALID, "y3_net_forward: grid %d null or not 16-byte aligned", i);
    if ((uintptr_t)images & 3) return fail(Y3_ERR_INVALID, "y3_net_forward: images not 4-byte aligned");
    Y3_ENTER_DEVICE(net);   // launches go to the net's device whatever the caller's current one is; restored on return
    // Images are independent, so the batch can run as `lanes` sub-batches on forked streams: while one sub-batch's
    // conv kernel drains (its last workgroups leave CUs under-occupied), the other sub-batch's kernel fills them.
    net->cur_batch = batch;
    int lanes = (ms_out || net->lanes < 2) ? 1 : net->lanes;
    while (lanes > 1 && batch / lanes < 1) --lanes;
    // leading segment in chunks small enough for their activations to stay in the 256 MB Infinity Cache between the
    // conv that writes them and the one that reads them, then the rest of the op list on the whole (sub-)batch
    const int k_early = ms_out ? 0 : net->early_ops;
    auto run_lane = [&](int b0, int nb, hipStream_t st, int lane, int nl) -> y3_status {
        if (k_early > 0) {
            for (int c0 = 0; c0 < nb; c0 += net->early_chunk) {
                const int cn = nb - c0 < net->early_chunk ? nb - c0 : net->early_chunk;
                y3_status r = run_slice(net, images, grids, b0 + c0, cn, st, nullptr, 0, lane, nl, 0, k_early);
                if (r != Y3_OK) return r;
            }
        }
        return run_slice(net, images, grids, b0, nb, st, ms_out, n_ms, lane, nl, k_early, -1);
    };
    if (lanes == 1) return run_lane(0, batch, s, 0, 1);
    if (y3_status st = ensure_lanes(net); st != Y3_OK) return st;
    HIP_TRY(hipEventRecord(net->fork_ev, s));
    // equal sub-batches (measured with tools/lanes_sweep.py: weighted 2:3 / 3:4:5 splits were 2-3 % slower)
    int start[Y3_MAX_LANES + 1];
    for (int l = 0; l <= lanes; ++l) start[l] = (int)((long long)batch * l / lanes);
    // lane 0 runs on the caller's stream itself: its first kernel needs no cross-queue signal to start, and the join waits for the other lanes only
    // (fp32 eager step 31.324 -> 31.218 ms, +0.3 %, every round of three; bf16 graph replay unchanged: profiles/r05_ab_lane0_on_caller_stream.txt)
    hipStream_t ls[Y3_MAX_LANES];
    for (int l = 0; l < Y3_MAX_LANES; ++l) ls[l] = l == 0 ? s : net->lane_stream[l];
    for (int l = 0; l < lanes; ++l)
        if (start[l + 1] > start[l] && ls[l] != s) HIP_TRY(hipStreamWaitEvent(ls[l], net->fork_ev, 0));
    if (k_early > 0) {
        for (int l = 0; l < lanes; ++l) {
            const int nb = start[l + 1] - start[l];
            if (nb <= 0) continue;
            y3_status st = run_lane(start[l], nb, ls[l], l, lanes);
            if (st != Y3_OK) return st;
        }
    } else {
        // op-major enqueue: op k of every lane before op k+1 of any.  Enqueued lane by lane, an eager forward gives lane 0 a
        // head start of one whole forward's worth of host launch time (0.3 ms; 0.9 ms under a profiler: the per-queue timeline of
        // tools/timeline_dump.py shows lane 0 four kernels ahead), and a start offset between the lanes only costs
        // (profiles/r03_ab_lane_stagger.txt).  A captured forward replays with both branches released at once either way.
        const int n_ops = (int)net->ops.size();
        // (A start offset between the lanes was measured again in round 5 for the fp32 plan -- lane 1 released 1 / 2 / 4 ops behind lane 0: at most
        // +0.28 %, inside the process-to-process spread, profiles/r05_ab_f32_lane_stagger.txt; not kept.)
        for (int oi = 0; oi < n_ops; ++oi)
            for (int l = 0; l < lanes; ++l) {
                const int nb = start[l + 1] - start[l];
                if (nb <= 0) continue;
                y3_status st = run_slice(net, images, grids, start[l], nb, ls[l], nullptr, 0, l, lanes, oi, oi + 1);
                if (st != Y3_OK) return st;
            }
    }
    for (int l = 0; l < lanes; ++l) {
        if (start[l + 1] <= start[l]) continue;
        if (ls[l] == s) continue;
        HIP_TRY(hipEventRecord(net->join_ev[l], ls[l]));
        HIP_TRY(hipStreamWaitEvent(s, net->join_ev[l], 0));
    }
    return Y3_OK;
}

y3_status y3_net_forward(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], void *stream)
try {
    return run(net, images_dev, batch, grids_dev, (hipStream_t)stream, nullptr, 0);
}
Y3_CATCH("y3_net_forward")

y3_status y3_net_profile_convs(y3_net *net, const float *images_dev, int batch, float *ms_out, int n, void *stream)
try {
    if (!net || !ms_out) return fail(Y3_ERR_INVALID, "y3_net_profile_convs: bad argument");
    // head grids go to scratch owned by this call
    if (!net->image_size || batch <= 0) return fail(Y3_ERR_STATE, "y3_net_profile_convs: call y3_net_plan first (and batch > 0)");
    Y3_ENTER_DEVICE(net);
    float *g[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        const int sp = spatial(net, net->outputs[i]);
        hipError_t e = hipMalloc(&g[i], (size_t)batch * sp * sp * net->tensors[net->outputs[i]].channels * sizeof(float));
        if (e != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipFree(g[k]);
            return fail(Y3_ERR_OOM, "y3_net_profile_convs: hipMalloc: %s", hipGetErrorString(e));
        }
    }
    y3_status st = run(net, images_dev, batch, g, (hipStream_t)stream, ms_out, n);
    (void)hipStreamSynchronize((hipStream_t)stream);
    for (int i = 0; i < 3; ++i) (void)hipFree(g[i]);
    return st;
}
Y3_CATCH("y3_net_profile_convs")

namespace {
// can conv slot i of this plan carry the clock stamps?  The fp32 MFMA kernel (fp32 plans) and the fused stem kernel do.
bool conv_carries_stamps(const y3_net *net, size_t i)
{
    const ConvSlot &c = net->convs[i];
    const bool stem = net->stem_fused && net->ops.size() > 1 && net->ops[1].kind == 0 && net->ops[1].index == (int)i;
    if (stem) return true;
    if (net->stem_fused && (i == (size_t)net->ops[0].index || (net->stem_conv2 && net->ops.size() > 2 && i == (size_t)net->ops[2].index)))
        return false;    // runs inside the stem launch
    return net->dtype == Y3_DTYPE_F32 && !c.first_layer;
}
y3_status measure_sclk_impl(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                            int pick, float *mhz_out, void *stream);
}  // namespace

y3_status y3_net_measure_sclk_conv(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                                   int conv, float *mhz_out, void *stream)
try {
    if (!net || !mhz_out || forwards < 1 || conv < 0 || conv >= (int)net->convs.size())
        return fail(Y3_ERR_INVALID, "y3_net_measure_sclk_conv: bad argument");
    if (!conv_carries_stamps(net, (size_t)conv))
        return fail(Y3_ERR_STATE, "y3_net_measure_sclk_conv: the launch of conv %d carries no clock stamps in this plan", conv);
    return measure_sclk_impl(net, images_dev, batch, grids_dev, forwards, conv, mhz_out, stream);
}
Y3_CATCH("y3_net_measure_sclk_conv")

y3_status y3_net_measure_sclk(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                              float *mhz_out, void *stream)
try {
    if (!net || !mhz_out || forwards < 1) return fail(Y3_ERR_INVALID, "y3_net_measure_sclk: bad argument");
    // the launch that carries the stamps: the conv with the most FLOPs among those whose kernel has them -- the fp32 MFMA
    // kernel (fp32 plans; a steady-state workgroup of a ~0.8 ms launch) or the fused stem kernel (fp32 and bf16 plans)
    int pick = -1;
    double best = 0;
    for (size_t i = 0; i < net->convs.size(); ++i) {
        const ConvSlot &c = net->convs[i];
        if (!conv_carries_stamps(net, i)) continue;
        const double ho = net->image_size ? net->image_size / c.d.out_div : 0;
        const double fl = 2.0 * c.d.size * c.d.size * c.d.cin * c.d.cout * ho * ho;
        if (fl > best) { best = fl; pick = (int)i; }
    }
    if (pick < 0) return fail(Y3_ERR_STATE, "y3_net_measure_sclk: no launch of this plan carries clock stamps (fp32 plan or fused stem needed)");
    return measure_sclk_impl(net, images_dev, batch, grids_dev, forwards, pick, mhz_out, stream);
}
Y3_CATCH("y3_net_measure_sclk")

namespace {
// pick >= 0: that conv, *mhz_out one value; pick == -2: every conv that carries stamps, mhz_out / start_us / end_us arrays of
// convs.size() entries (0 where a conv left no stamps; times relative to the earliest stamp, from s_memrealtime)
y3_status measure_sclk_arrays(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                              int pick, float *mhz_out, double *start_us, double *end_us, void *stream)
{
    Y3_ENTER_DEVICE(net);
    const size_t nconv = net->convs.size();
    const size_t words = pick == -2 ? 8 * nconv : 8;   // per conv: memtime, realtime at entry; the same after the epilogue; realtime at the entry of workgroup 0
    std::vector<unsigned long long> host(words, 0ull);
    unsigned long long *buf = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), words * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(buf, 0, words * sizeof(unsigned long long), (hipStream_t)stream);
    // the chip's clock follows the load of the last milliseconds: stamp the LAST of `forwards` back-to-back forwards
    y3_status st = Y3_OK;
    const int lanes_saved = net->lanes;
    net->lanes = 1;             // one launch of a stamped conv (concurrent sub-batches would each stamp the same words)
    for (int i = 0; i < forwards && st == Y3_OK && e == hipSuccess; ++i) {
        net->clk_stamps = (i == forwards - 1) ? buf : nullptr;
        net->clk_conv = (i == forwards - 1) ? pick : -1;
        st = run(net, images_dev, batch, grids_dev, (hipStream_t)stream, nullptr, 0);
    }
    net->clk_stamps = nullptr;
    net->clk_conv = -1;
    net->lanes = lanes_saved;
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpy(host.data(), buf, words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (st != Y3_OK) return st;
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_net_measure_sclk: %s", hipGetErrorString(e));
    if (pick >= 0) {
        const double ticks = (double)(host[2] - host[0]), real = (double)(host[3] - host[1]);
        if (!(real > 0.0) || !(ticks > 0.0)) return fail(Y3_ERR_STATE, "y3_net_measure_sclk: conv %d left no stamps", pick);
        *mhz_out = (float)(ticks / real * 100.0);   // s_memrealtime counts at 100 MHz
        return Y3_OK;
    }
    // start = entry of the launch's FIRST workgroup (word 4; the stem kernel stamps in workgroup 0 throughout: word 1),
    // end = after the epilogue of the clock-stamped workgroup (a middle one; the stem: workgroup 0, resident to the end)
    auto first = [&](size_t c) { return host[8 * c + 4] ? host[8 * c + 4] : host[8 * c + 1]; };
    unsigned long long t0 = ~0ull;
    for (size_t c = 0; c < nconv; ++c)
        if (host[8 * c + 3] > host[8 * c + 1] && first(c) < t0) t0 = first(c);
    int stamped = 0;
    for (size_t c = 0; c < nconv; ++c) {
        const double ticks = (double)(host[8 * c + 2] - host[8 * c]), real = (double)(host[8 * c + 3] - host[8 * c + 1]);
        const bool ok = host[8 * c + 3] > host[8 * c + 1] && host[8 * c + 2] > host[8 * c];
        mhz_out[c] = ok ? (float)(ticks / real * 100.0) : 0.0f;
        if (start_us) start_us[c] = ok ? (double)(first(c) - t0) / 100.0 : 0.0;
        if (end_us) end_us[c] = ok ? (double)(host[8 * c + 3] - t0) / 100.0 : 0.0;
        stamped += ok;
    }
    if (!stamped) return fail(Y3_ERR_STATE, "y3_net_measure_sclk_all: no launch of this plan left clock stamps (fp32 plan or fused stem needed)");
    return Y3_OK;
}
y3_status measure_sclk_impl(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                            int pick, float *mhz_out, void *stream)
{
    return measure_sclk_arrays(net, images_dev, batch, grids_dev, forwards, pick, mhz_out, nullptr, nullptr, stream);
}
}  // namespace

y3_status y3_net_measure_sclk_all(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                                  float *mhz_out, double *start_us, double *end_us, void *stream)
try {
    if (!net || !mhz_out || forwards < 1) return fail(Y3_ERR_INVALID, "y3_net_measure_sclk_all: bad argument");
    return measure_sclk_arrays(net, images_dev, batch, grids_dev, forwards, -2, mhz_out, start_us, end_us, stream);
}
Y3_CATCH("y3_net_measure_sclk_all")

y3_status y3_net_read_tensor(y3_net *net, int t, int batch, float *dst_dev, size_t *n_elems, void *stream)
try {
    if (!net || t < 0 || t >= (int)net->tensors.size() || !net->image_size)
        return fail(Y3_ERR_INVALID, "y3_net_read_tensor: bad argument");
    const int sp = spatial(net, t);
    const size_t n = (size_t)batch * sp * sp * net->tensors[t].channels;
    if (n_elems) *n_elems = n;
    if (!dst_dev) return Y3_OK;
    if (!net->tdev[t]) return fail(Y3_ERR_STATE, "y3_net_read_tensor: tensor %d is not held in the arena", t);
    Y3_ENTER_DEVICE(net);   // the conversion kernels / the copy below read the net's arena: enqueue them on its device
    if (net->dtype == Y3_DTYPE_F32X2) {
        hipError_t e = y3::launch_x2_to_f32(net->tdev[t], dst_dev, (size_t)batch * sp * sp, net->tensors[t].channels, (hipStream_t)stream);
        if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_net_read_tensor: %s", hipGetErrorString(e));
        return Y3_OK;
    }
    if (net->dtype == Y3_DTYPE_F32X3) {
        hipError_t e = y3::launch_x3_to_f32(net->tdev[t], dst_dev, (size_t)batch * sp * sp, net->tensors[t].channels, (hipStream_t)stream);
        if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_net_read_tensor: %s", hipGetErrorString(e));
        return Y3_OK;
    }
    if (net->dtype == Y3_DTYPE_BF16) {
        hipError_t e = y3::launch_bf16_to_f32(net->tdev[t], dst_dev, n, (hipStream_t)stream);
        if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_net_read_tensor: %s", hipGetErrorString(e));
        return Y3_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst_dev, net->tdev[t], n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return Y3_OK;
}
Y3_CATCH("y3_net_read_tensor")

// ------------------------------------------------------------------------------------------ image input
y3_status y3_preprocess_image(const void *image_dev, int is_uint8, int height, int width, int channels,
                              float *batch_dev, int slot, int image_size, void *stream)
try {
    if (!image_dev || !batch_dev || height <= 0 || width <= 0 || channels < 3 || channels > 4 || slot < 0 ||
        image_size <= 0 || is_uint8 < 0 || is_uint8 > 2 || (is_uint8 == 0 && ((uintptr_t)image_dev & 3)))
        return fail(Y3_ERR_INVALID, "y3_preprocess_image: bad argument (channels must be 3 or 4)");
    float *dst = batch_dev + (size_t)slot * image_size * image_size * 3;
    hipError_t e = y3::launch_resize(image_dev, is_uint8, height, width, channels, dst, image_size, (hipStream_t)stream);
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_preprocess_image launch: %s", hipGetErrorString(e));
    return Y3_OK;
}
Y3_CATCH("y3_preprocess_image")

// ------------------------------------------------------------------------------------------ TFRecord checksum
uint32_t y3_crc32c(const void *data_host, size_t nbytes)
{
    // slicing-by-8 over the reflected Castagnoli polynomial
    static uint32_t T[8][256];
    static bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        return true;
    }();
    (void)ready;
    const unsigned char *p = static_cast<const unsigned char *>(data_host);
    uint32_t c = 0xFFFFFFFFu;
    while (nbytes >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^ T[3][hi & 0xFF] ^
            T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8;
        nbytes -= 8;
    }
    while (nbytes--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF];
    return c ^ 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------ decode
static y3_status decode_common(const float *const grids[3], const int32_t gs[3], int batch, int nc,
                               const float *anchors, float *bboxes, float *conf, float *probs, int64_t *cls,
                               float *scores, void *stream, const char *who)
{
    if (!grids || !gs || !anchors || !bboxes || batch <= 0 || nc <= 0) return fail(Y3_ERR_INVALID, "%s: bad argument", who);
    y3::DecodeArgs a{};
    int off = 0;
    for (int s = 0; s < 3; ++s) {
        if (!grids[s] || gs[s] <= 0 || ((uintptr_t)grids[s] & 15))
            return fail(Y3_ERR_INVALID, "%s: grid %d null, empty or not 16-byte aligned", who, s);
        a.grid[s] = grids[s];
        a.g[s] = gs[s];
        a.off[s] = off;
        off += gs[s] * gs[s] * 3;
        for (int k = 0; k < 3; ++k) {
            a.anchors[s][k][0] = anchors[(s * 3 + k) * 2 + 0];
            a.anchors[s][k][1] = anchors[(s * 3 + k) * 2 + 1];
        }
    }
    if ((uintptr_t)bboxes & 15) return fail(Y3_ERR_INVALID, "%s: bboxes not 16-byte aligned", who);
    a.B = batch;
    a.N = off;
    a.nc = nc;
    hipError_t e = y3::launch_decode(a, bboxes, conf, probs, cls, scores, (hipStream_t)stream);
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "%s launch: %s", who, hipGetErrorString(e));
    return Y3_OK;
}

y3_status y3_yolo_decode(const float *const grids_dev[3], const int32_t grid_sizes[3], int batch, int nclasses,
                         const float *anchors_host, float *bboxes_dev, float *conf_dev, float *probs_dev, void *stream)
try {
    if (!conf_dev || !probs_dev) return fail(Y3_ERR_INVALID, "y3_yolo_decode: null output");
    return decode_common(grids_dev, grid_sizes, batch, nclasses, anchors_host, bboxes_dev, conf_dev, probs_dev, nullptr,
                         nullptr, stream, "y3_yolo_decode");
}
Y3_CATCH("y3_yolo_decode")

y3_status y3_yolo_decode_scores(const float *const grids_dev[3], const int32_t grid_sizes[3], int batch, int nclasses,
                                const float *anchors_host, float *bboxes_dev, int64_t *class_idx_dev,
                                float *scores_dev, void *stream)
try {
    if (!class_idx_dev || !scores_dev) return fail(Y3_ERR_INVALID, "y3_yolo_decode_scores: null output");
    return decode_common(grids_dev, grid_sizes, batch, nclasses, anchors_host, bboxes_dev, nullptr, nullptr,
                         class_idx_dev, scores_dev, stream, "y3_yolo_decode_scores");
}
Y3_CATCH("y3_yolo_decode_scores")

y3_status y3_class_scores(const float *conf_dev, const float *probs_dev, int batch, int n, int nclasses,
                          int64_t *class_idx_dev, float *scores_dev, void *stream)
try {
    if (!conf_dev || !probs_dev || !class_idx_dev || !scores_dev || batch <= 0 || n <= 0 || nclasses <= 0)
        return fail(Y3_ERR_INVALID, "y3_class_scores: bad argument");
    hipError_t e = y3::launch_class_scores(conf_dev, probs_dev, (size_t)batch * n, nclasses, class_idx_dev, scores_dev,
                                           (hipStream_t)stream);
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_class_scores launch: %s", hipGetErrorString(e));
    return Y3_OK;
}
Y3_CATCH("y3_class_scores")

// ------------------------------------------------------------------------------------------ forward + decode
namespace {
// Can the three heads decode in place?  Each output must come from a 1x1 / stride-1 single-source conv without shortcut whose
// 3 * (5 + nc) channels fit one 256-wide tile, written straight to the caller-visible grid (not staged), in an fp32 or bf16
// plan.  Y3_FUSE_DECODE=0 (tools: A/B against the composed route) switches the fusion off.
bool heads_can_decode(const y3_net *net)
{
    static const bool off = [] { const char *e = getenv("Y3_FUSE_DECODE"); return e && e[0] == '0'; }();
    if (off || net->nclasses <= 0 || (net->dtype != Y3_DTYPE_F32 && net->dtype != Y3_DTYPE_BF16)) return false;
    if (net->keep_all || net->early_ops > 0 || 3 * (5 + net->nclasses) > 256) return false;
    for (int k = 0; k < 3; ++k) {
        const int t = net->outputs[k];
        if (net->staged[t]) return false;
        int producers = 0;
        for (const ConvSlot &c : net->convs) {
            if (c.d.dst != t) continue;
            ++producers;
            if (c.first_layer || c.d.size != 1 || c.d.stride != 1 || c.d.src1 >= 0 || c.d.residual >= 0 || c.d.cin % 64 ||
                c.d.cout != 3 * (5 + net->nclasses) || c.cout_pad != 256)
                return false;
        }
        if (producers != 1) return false;
        // the fused route does not write the grid: nobody inside the net may read it (fp32 plans never stage an output, so a
        // consumer would read the caller's buffer -- on this route uninitialised scratch)
        for (const ConvSlot &c : net->convs)
            if (c.d.src0 == t || c.d.src1 == t || c.d.residual == t) return false;
        for (const y3_aux_desc &x : net->aux)
            if (x.dst == t || x.src0 == t || x.src1 == t) return false;
    }
    return true;
}
}  // namespace

y3_status y3_net_forward_decode(y3_net *net, const float *images_dev, int batch, const float *anchors_host, float *bboxes_dev,
                                int64_t *class_idx_dev, float *scores_dev, void *stream)
try {
    if (!net || !images_dev || !anchors_host || !bboxes_dev || !class_idx_dev || !scores_dev || batch <= 0)
        return fail(Y3_ERR_INVALID, "y3_net_forward_decode: bad argument");
    if (net->nclasses <= 0) return fail(Y3_ERR_STATE, "y3_net_forward_decode: the net was created without detection heads (nclasses = 0)");
    if (!net->image_size) return fail(Y3_ERR_STATE, "y3_net_forward_decode: call y3_net_plan first");
    if (batch > net->max_batch) return fail(Y3_ERR_INVALID, "y3_net_forward_decode: batch %d > planned %d", batch, net->max_batch);
    if ((uintptr_t)bboxes_dev & 15) return fail(Y3_ERR_INVALID, "y3_net_forward_decode: bboxes not 16-byte aligned");
    Y3_ENTER_DEVICE(net);
    int32_t gs[3];
    size_t gelems[3], n = 0, off[9];
    detect_layout(net, batch, off, &n, gs, gelems);
    if (!net->det_buf || net->det_bytes < off[8])
        return fail(Y3_ERR_STATE, "y3_net_forward_decode: detect scratch not planned (y3_net_plan allocates it)");
    char *b = static_cast<char *>(net->det_buf);
    float *grids[3] = {reinterpret_cast<float *>(b + off[0]), reinterpret_cast<float *>(b + off[1]), reinterpret_cast<float *>(b + off[2])};
    if (!heads_can_decode(net)) {   // composed route: grids into the scratch, then the stand-alone decode
        y3_status st = y3_net_forward(net, images_dev, batch, grids, stream);
        if (st != Y3_OK) return st;
        return y3_yolo_decode_scores(grids, gs, batch, net->nclasses, anchors_host, bboxes_dev, class_idx_dev, scores_dev, stream);
    }
    y3::DecodeHead heads[3];
    int first = 0;
    for (int k = 0; k < 3; ++k) {
        heads[k].boxes = bboxes_dev;
        heads[k].cls = class_idx_dev;
        heads[k].scores = scores_dev;
        heads[k].g = gs[k];
        heads[k].off = first;
        heads[k].N = (int)n;
        heads[k].nc = net->nclasses;
        for (int a = 0; a < 3; ++a) {
            heads[k].anchors[a][0] = anchors_host[(k * 3 + a) * 2 + 0];
            heads[k].anchors[a][1] = anchors_host[(k * 3 + a) * 2 + 1];
        }
        first += gs[k] * gs[k] * 3;
    }
    net->fuse = heads;
    y3_status st = run(net, images_dev, batch, grids, (hipStream_t)stream, nullptr, 0);
    net->fuse = nullptr;
    return st;
}
Y3_CATCH("y3_net_forward_decode")

// ------------------------------------------------------------------------------------------ nms
// ------------------------------------------------------------------------------------------ whole pipeline
y3_status y3_net_detect(y3_net *net, const float *images_dev, int batch, const float *anchors_host, int max_boxes,
                        float iou_threshold, float score_threshold, void *packed_dev, int32_t *num_valid_dev,
                        void *stream)
try {
    if (!net || !images_dev || !anchors_host || !packed_dev || !num_valid_dev || batch <= 0)
        return fail(Y3_ERR_INVALID, "y3_net_detect: bad argument");
    if (net->nclasses <= 0) return fail(Y3_ERR_STATE, "y3_net_detect: the net was created without detection heads (nclasses = 0)");
    if (!net->image_size) return fail(Y3_ERR_STATE, "y3_net_detect: call y3_net_plan first");
    if (batch > net->max_batch) return fail(Y3_ERR_INVALID, "y3_net_detect: batch %d > planned %d", batch, net->max_batch);
    if (max_boxes <= 0 || max_boxes > Y3_MAX_OUTPUT_BOXES)
        return fail(Y3_ERR_INVALID, "y3_net_detect: max_boxes must be in [1,%d]", Y3_MAX_OUTPUT_BOXES);
    Y3_ENTER_DEVICE(net);   // the decode / NMS / pack launches below go to the net's device; the caller's current device is restored on return
    int32_t gs[3];
    size_t gelems[3], n = 0, off[9];
    detect_layout(net, batch, off, &n, gs, gelems);
    if (!net->det_buf || net->det_bytes < off[8])
        return fail(Y3_ERR_STATE, "y3_net_detect: detect scratch not planned (y3_net_plan allocates it)");
    const size_t o_box = off[3], o_cls = off[4], o_score = off[5];
    const size_t o_sel = off[6], o_ws = off[7];
    const size_t ws_bytes = y3::nms_workspace_bytes(batch, (int)n);
    char *b = static_cast<char *>(net->det_buf);
    float *boxes = reinterpret_cast<float *>(b + o_box), *scores = reinterpret_cast<float *>(b + o_score);
    int64_t *cls = reinterpret_cast<int64_t *>(b + o_cls);
    int32_t *sel = reinterpret_cast<int32_t *>(b + o_sel);
    // conv program with the head convs decoding their own tiles (grids neither written nor read back) where the graph allows it.
    // (Round 5 ran NMS + pack per lane, on each lane's stream behind its last conv: bit-identical and 0.2 % slower under graph replay -- the NMS
    // workgroups take CUs from the other lane's last convs; profiles/r05_ab_bf16_lane_nms.txt.  Batch-wide launches behind the join again.)
    y3_status st = y3_net_forward_decode(net, images_dev, batch, anchors_host, boxes, cls, scores, stream);
    if (st != Y3_OK) return st;
    st = y3_nms_padded(boxes, scores, batch, (int)n, max_boxes, iou_threshold, score_threshold, sel, num_valid_dev,
                       b + o_ws, ws_bytes, stream);
    if (st != Y3_OK) return st;
    return y3_pack_detections(boxes, cls, scores, sel, num_valid_dev, batch, (int)n, max_boxes, packed_dev, stream);
}
Y3_CATCH("y3_net_detect")

size_t y3_nms_workspace_bytes(int batch, int n) { return (batch > 0 && n > 0) ? y3::nms_workspace_bytes(batch, n) : 0; }

y3_status y3_nms_padded(const float *bboxes_dev, const float *scores_dev, int batch, int n, int max_output_size,
                        float iou_threshold, float score_threshold, int32_t *selected_idx_dev,
                        int32_t *num_valid_dev, void *workspace_dev, size_t workspace_bytes, void *stream)
try {
    if (!bboxes_dev || !scores_dev || !selected_idx_dev || !num_valid_dev || batch <= 0 || n <= 0)
        return fail(Y3_ERR_INVALID, "y3_nms_padded: bad argument");
    if (max_output_size <= 0 || max_output_size > 1024)
        return fail(Y3_ERR_INVALID, "y3_nms_padded: max_output_size must be in [1,1024]");
    if ((uintptr_t)bboxes_dev & 15) return fail(Y3_ERR_INVALID, "y3_nms_padded: bboxes not 16-byte aligned");
    if (!(iou_threshold > 0.0f) && score_threshold < 0.0f)
        return fail(Y3_ERR_INVALID, "y3_nms_padded: iou_threshold <= 0 together with score_threshold < 0 is not supported");
    if (!workspace_dev || workspace_bytes < y3::nms_workspace_bytes(batch, n))
        return fail(Y3_ERR_INVALID, "y3_nms_padded: workspace too small (need %zu bytes)", y3::nms_workspace_bytes(batch, n));
    hipError_t e = y3::launch_nms(bboxes_dev, scores_dev, batch, n, max_output_size, iou_threshold, score_threshold,
                                  selected_idx_dev, num_valid_dev, workspace_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_nms_padded launch: %s", hipGetErrorString(e));
    return Y3_OK;
}
Y3_CATCH("y3_nms_padded")

y3_status y3_pack_detections(const float *bboxes_dev, const int64_t *class_idx_dev, const float *scores_dev,
                             const int32_t *selected_idx_dev, const int32_t *num_valid_dev, int batch, int n,
                             int max_out, void *packed_dev, void *stream)
try {
    if (!bboxes_dev || !class_idx_dev || !scores_dev || !selected_idx_dev || !num_valid_dev || !packed_dev ||
        batch <= 0 || n <= 0 || max_out <= 0)
        return fail(Y3_ERR_INVALID, "y3_pack_detections: bad argument");
    hipError_t e = y3::launch_pack(bboxes_dev, class_idx_dev, scores_dev, selected_idx_dev, num_valid_dev, batch, n,
                                   max_out, packed_dev, (hipStream_t)stream);
    if (e != hipSuccess) return fail(Y3_ERR_HIP, "y3_pack_detections launch: %s", hipGetErrorString(e));
    return Y3_OK;
}
Y3_CATCH("y3_pack_detections")

}  // extern "C"
