// libzkt_comm_rccl.so: zkt_comm_vtable over RCCL (include/zkt_comm_rccl.h).  Host code only; links librccl + libamdhip64.
#include "../../include/zkt_comm_rccl.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

static_assert(sizeof(ncclUniqueId) == ZKT_COMM_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

struct zkt_comm_rccl {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t own_stream = nullptr;   // host-buffer exchanges
    void* stage = nullptr;              // device staging: send slot + world receive slots
    size_t stage_bytes = 0;
    bool dead = false;                  // a collective failed locally: the communicator was aborted, every later call fails
    std::string err;
};

namespace {

int fail(zkt_comm_rccl* c, int code, const char* what, const char* detail) {
    if (c) c->err = std::string(what) + ": " + (detail ? detail : "?");
    return code;
}
#define CK_HIP(c, call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) return fail((c), ZKT_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)
#define CK_NCCL(c, call)                                                                 \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) return fail((c), ZKT_ERR_COMM, #call, ncclGetErrorString(r_)); \
    } while (0)

// A rank that fails inside an exchange must not leave its peers blocked in theirs: the communicator is aborted (the
// peers' pending and later collectives then fail instead of hanging) and marked dead.  A communicator is used by ONE host
// thread at a time (like the context it serves): the staging buffer and the error text are not guarded.
int poison(zkt_comm_rccl* c, int rc) {
    if (rc && c && c->comm && !c->dead) {
        c->dead = true;
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
    }
    return rc;
}

int ensure_stage(zkt_comm_rccl* c, size_t bytes) {
    if (c->stage_bytes >= bytes) return ZKT_OK;
    if (c->stage) CK_HIP(c, hipFree(c->stage));
    c->stage = nullptr;
    c->stage_bytes = 0;
    CK_HIP(c, hipMalloc(&c->stage, bytes));
    c->stage_bytes = bytes;
    return ZKT_OK;
}

int all_gather_impl(zkt_comm_rccl* c, const void* send, void* recv, size_t bytes, int on_device, void* hip_stream, bool wait) {
    if (bytes == 0) return 0;
    const size_t world = (size_t)c->world;
    CK_HIP(c, hipSetDevice(c->device));
    if (on_device) {
        hipStream_t st = static_cast<hipStream_t>(hip_stream);
        const char* s = static_cast<const char*>(send);
        char* r = static_cast<char*>(recv);
        const bool in_place = s == r + (size_t)c->rank * bytes;
        const bool overlap = s < r + world * bytes && r < s + bytes;
        if (overlap && !in_place) {   // not RCCL's in-place form: give the send data a home of its own
            if (int rc = ensure_stage(c, bytes)) return rc;
            CK_HIP(c, hipMemcpyAsync(c->stage, send, bytes, hipMemcpyDeviceToDevice, st));
            send = c->stage;
        }
        CK_NCCL(c, ncclAllGather(send, recv, bytes, ncclChar, c->comm, st));
        if (wait) CK_HIP(c, hipStreamSynchronize(st));   // all_gather's contract: complete when the callback returns
        return 0;
    }
    if (int rc = ensure_stage(c, bytes * (world + 1))) return rc;
    char* ds = static_cast<char*>(c->stage);
    char* dr = ds + bytes;
    CK_HIP(c, hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->own_stream));
    CK_NCCL(c, ncclAllGather(ds, dr, bytes, ncclChar, c->comm, c->own_stream));
    CK_HIP(c, hipMemcpyAsync(recv, dr, bytes * world, hipMemcpyDeviceToHost, c->own_stream));
    CK_HIP(c, hipStreamSynchronize(c->own_stream));
    return 0;
}

int all_gather(void* user, const void* send, void* recv, size_t bytes, int on_device, void* hip_stream) {
    zkt_comm_rccl* c = static_cast<zkt_comm_rccl*>(user);
    if (!c || !c->comm || c->dead) return 1;
    return poison(c, all_gather_impl(c, send, recv, bytes, on_device, hip_stream, true));
}
// zkt_comm_vtable::all_gather_async: the same collective, enqueued on the caller's stream and left there
int all_gather_async(void* user, const void* d_send, void* d_recv, size_t bytes, void* hip_stream) {
    zkt_comm_rccl* c = static_cast<zkt_comm_rccl*>(user);
    if (!c || !c->comm || c->dead) return 1;
    return poison(c, all_gather_impl(c, d_send, d_recv, bytes, 1, hip_stream, false));
}

}  // namespace

extern "C" {

int zkt_comm_rccl_unique_id(uint8_t out[ZKT_COMM_RCCL_UNIQUE_ID_BYTES]) {
    if (!out) return ZKT_ERR_INVALID_ARGUMENT;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ZKT_ERR_COMM;
    memcpy(out, &id, sizeof(id));
    return ZKT_OK;
}

int zkt_comm_rccl_create(const uint8_t id_bytes[ZKT_COMM_RCCL_UNIQUE_ID_BYTES], int rank, int world, int device,
                         zkt_comm_rccl** out) {
    if (!out) return ZKT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!id_bytes || rank < 0 || rank >= world || !(world == 1 || world == 2 || world == 4 || world == 8))
        return ZKT_ERR_INVALID_ARGUMENT;
    zkt_comm_rccl* c = new zkt_comm_rccl();
    c->rank = rank;
    c->world = world;
    c->device = device;
    auto init = [&]() -> int {
        CK_HIP(c, hipSetDevice(device));
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        CK_NCCL(c, ncclCommInitRank(&c->comm, world, id, rank));
        CK_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        return ZKT_OK;
    };
    const int rc = init();
    if (rc) {
        zkt_comm_rccl_destroy(c);
        return rc;
    }
    *out = c;
    return ZKT_OK;
}

int zkt_comm_rccl_vtable(zkt_comm_rccl* c, zkt_comm_vtable* out) {
    if (!c || !out) return ZKT_ERR_INVALID_ARGUMENT;
    out->user = c;
    out->rank = c->rank;
    out->world = c->world;
    out->device_buffers = 1;
    out->all_gather = all_gather;
    out->all_gather_async = all_gather_async;
    return ZKT_OK;
}

void zkt_comm_rccl_destroy(zkt_comm_rccl* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) {
        (void)hipStreamSynchronize(c->own_stream);
        (void)hipStreamDestroy(c->own_stream);
    }
    if (c->stage) (void)hipFree(c->stage);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

const char* zkt_comm_rccl_last_error(const zkt_comm_rccl* c) { return c ? c->err.c_str() : ""; }

}  // extern "C"
