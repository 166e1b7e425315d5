// Multi-GPU exchange of the detect path behind the C ABI: one RCCL communicator per process (one process per GPU) and
// ONE collective per batch -- the all-gather of the packed detections [batch, max_boxes, 7] and of num_valid [batch],
// issued as a single RCCL group on the caller's stream, so that y3_net_detect + y3_allgather_results can be captured
// into one HIP graph.  The reference has no multi-device code; the message is what north_star calls
// "the final box list" (SURVEY.md 8e): 2804 B per image, 179 KB per rank at 64 images -- latency-bound, xGMI direct.
//
// librccl is bound at run time (dlopen): the process usually already holds one (PyTorch-ROCm ships its own copy and
// loads it first), and a library that is used without a communicator must not need RCCL to load.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>

#include "../../include/y3.h"

namespace y3 {
int fail_msg(int code, const char *fmt, ...) noexcept;   // y3_api.cpp: fills the thread-local error buffer, returns code
int on_exception(const char *who) noexcept;              // y3_api.cpp: the exception barrier of the C ABI (rethrows and classifies)
bool test_fail_alloc() noexcept;                         // y3_api.cpp: Y3_TEST_FAIL_ALLOC=1 (tests)
}
#define Y3_CATCH(who) catch (...) { return y3::on_exception(who); }

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an instance already mapped into the process first (RTLD_NOLOAD), then the system one
        // Y3_RCCL_LIB names the library instead (tests use it to force the not-found path)
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        const char *forced = getenv("Y3_RCCL_LIB");
        if (forced && forced[0]) {
            r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *n : names)
                if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            for (const char *n : names)
                if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            const char *m = dlerror();   // one call: dlerror() clears the message it returns
            r.error = std::string("librccl not found: ") + (m ? m : "?");
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

#define RCCL_TRY(r, expr)                                                                                      \
    do {                                                                                                       \
        ncclResult_t e_ = (expr);                                                                              \
        if (e_ != ncclSuccess) return y3::fail_msg(Y3_ERR_COMM, "%s: %s", #expr, (r).GetErrorString(e_));     \
    } while (0)

}  // namespace

struct y3_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" {

y3_status y3_comm_get_unique_id(void *id_out)
try {
    static_assert(Y3_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return y3::fail_msg(Y3_ERR_INVALID, "y3_comm_get_unique_id: null argument");
    Rccl &r = rccl();
    if (!r.error.empty()) return y3::fail_msg(Y3_ERR_COMM, "%s", r.error.c_str());
    ncclUniqueId id;
    RCCL_TRY(r, r.GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return Y3_OK;
}
Y3_CATCH("y3_comm_get_unique_id")

y3_status y3_comm_init_rank(const void *id, int world_size, int rank, y3_comm **out)
try {
    if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size)
        return y3::fail_msg(Y3_ERR_INVALID, "y3_comm_init_rank: bad argument");
    if (y3::test_fail_alloc()) throw std::bad_alloc();   // tests only: the allocation below failing
    Rccl &r = rccl();
    if (!r.error.empty()) return y3::fail_msg(Y3_ERR_COMM, "%s", r.error.c_str());
    std::unique_ptr<y3_comm> c(new y3_comm());
    if (hipGetDevice(&c->device) != hipSuccess) return y3::fail_msg(Y3_ERR_NODEVICE, "y3_comm_init_rank: no HIP device");
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t e = r.CommInitRank(&c->comm, world_size, uid, rank);
    if (e != ncclSuccess)
        return y3::fail_msg(Y3_ERR_COMM, "ncclCommInitRank(world %d, rank %d): %s", world_size, rank, r.GetErrorString(e));
    c->world = world_size;
    c->rank = rank;
    *out = c.release();
    return Y3_OK;
}
Y3_CATCH("y3_comm_init_rank")

void y3_comm_destroy(y3_comm *comm)
{
    if (!comm) return;
    Rccl &r = rccl();
    if (comm->comm && r.CommDestroy) (void)r.CommDestroy(comm->comm);
    delete comm;
}

y3_status y3_comm_info(const y3_comm *comm, int32_t *world_size, int32_t *rank)
{
    if (!comm) return y3::fail_msg(Y3_ERR_INVALID, "y3_comm_info: null communicator");
    if (world_size) *world_size = comm->world;
    if (rank) *rank = comm->rank;
    return Y3_OK;
}

y3_status y3_allgather_results(y3_comm *comm, const void *packed_dev, const int32_t *num_valid_dev, int batch,
                               int max_boxes, void *packed_all_dev, int32_t *num_valid_all_dev, void *stream)
try {
    if (!comm || !packed_dev || !num_valid_dev || !packed_all_dev || !num_valid_all_dev || batch <= 0 || max_boxes <= 0)
        return y3::fail_msg(Y3_ERR_INVALID, "y3_allgather_results: bad argument");
    Rccl &r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream);
    // both gathers travel as one RCCL group = one launch on the caller's stream
    RCCL_TRY(r, r.GroupStart());
    ncclResult_t e1 = r.AllGather(packed_dev, packed_all_dev, (size_t)batch * max_boxes * 7, ncclInt32, comm->comm, s);
    ncclResult_t e2 = r.AllGather(num_valid_dev, num_valid_all_dev, (size_t)batch, ncclInt32, comm->comm, s);
    ncclResult_t e3 = r.GroupEnd();
    if (e1 != ncclSuccess || e2 != ncclSuccess || e3 != ncclSuccess)
        return y3::fail_msg(Y3_ERR_COMM, "y3_allgather_results: %s",
                            r.GetErrorString(e1 != ncclSuccess ? e1 : e2 != ncclSuccess ? e2 : e3));
    return Y3_OK;
}
Y3_CATCH("y3_allgather_results")

}  // extern "C"
