// Multi-GPU exchange for the sharded NORA sweep: one process per GPU, RCCL over xGMI.
// Replaces the mpi4py gathers of per-rank pools (gpry/gp_acquisition.py:1148-1191,
// gpry/mpi.py:118-131) with one fixed-size all-gather of shortlist records.
#include "common.h"
#include <rccl/rccl.h>

struct gpry_comm {
    gpry_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    void *dsend = nullptr, *drecv = nullptr;
    int64_t cap_send = 0, cap_recv = 0;
    // The exchanges move host buffers only, so they run on a stream of their own: nothing on the
    // compute stream ever queues behind a collective (a peer that never shows up must not wedge it).
    hipStream_t stream = nullptr;
};

#define NCCL_TRY(ctx, expr)                                                              \
    do {                                                                                 \
        ncclResult_t _r = (expr);                                                        \
        if (_r != ncclSuccess)                                                           \
            return gpry_fail(ctx, -5, "%s failed: %s", #expr, ncclGetErrorString(_r));   \
    } while (0)

static int comm_buffers(gpry_comm* c, int64_t send_bytes, int64_t recv_bytes) {
    gpry_ctx* ctx = c->ctx;
    if (send_bytes > c->cap_send) {
        if (c->dsend) HIP_TRY(ctx, hipFree(c->dsend));
        HIP_TRY(ctx, hipMalloc(&c->dsend, (size_t)send_bytes));
        c->cap_send = send_bytes;
    }
    if (recv_bytes > c->cap_recv) {
        if (c->drecv) HIP_TRY(ctx, hipFree(c->drecv));
        HIP_TRY(ctx, hipMalloc(&c->drecv, (size_t)recv_bytes));
        c->cap_recv = recv_bytes;
    }
    return 0;
}

extern "C" {

int gpry_comm_unique_id(uint8_t id[128]) {
    ncclUniqueId uid;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclResult_t r = ncclGetUniqueId(&uid);
    if (r != ncclSuccess) return gpry_fail(nullptr, -5, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id, &uid, 128);
    return 0;
}

int gpry_comm_init(gpry_ctx* ctx, int world, int rank, const uint8_t id[128], gpry_comm** out) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_comm_init: ctx is NULL");
    if (!out || !id) return gpry_fail(ctx, -1, "comm_init: id and out must not be NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return gpry_fail(ctx, -1, "comm_init: rank %d outside world of %d", rank, world);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gpry_comm* c = new gpry_comm();
    c->ctx = ctx; c->world = world; c->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return gpry_fail(ctx, -2, "comm_init: hipStreamCreate failed");
    }
    ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return gpry_fail(ctx, -5, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    *out = c;
    return 0;
}

int gpry_comm_info(gpry_comm* c, int* world, int* rank, int* device) {
    if (!c) return gpry_fail(nullptr, -1, "gpry_comm_info: communicator is NULL");
    int n = 0, r = 0, dv = 0;
    NCCL_TRY(c->ctx, ncclCommCount(c->comm, &n));
    NCCL_TRY(c->ctx, ncclCommUserRank(c->comm, &r));
    NCCL_TRY(c->ctx, ncclCommCuDevice(c->comm, &dv));
    if (world) *world = n;
    if (rank) *rank = r;
    if (device) *device = dv;
    return 0;
}

int gpry_comm_destroy(gpry_comm* c) {
    if (!c) return 0;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int gpry_comm_allgather(gpry_comm* c, const void* send, int64_t bytes, void* recv) {
    if (!c) return gpry_fail(nullptr, -1, "gpry_comm_allgather: communicator is NULL");
    gpry_ctx* ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GPRY_TRY(comm_buffers(c, bytes, bytes * c->world));
    hipStream_t st = c->stream;
    HIP_TRY(ctx, hipMemcpyAsync(c->dsend, send, (size_t)bytes, hipMemcpyHostToDevice, st));
    NCCL_TRY(ctx, ncclAllGather(c->dsend, c->drecv, (size_t)bytes, ncclChar, c->comm, st));
    HIP_TRY(ctx, hipMemcpyAsync(recv, c->drecv, (size_t)(bytes * c->world), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int gpry_comm_allreduce_max(gpry_comm* c, double* inout, int64_t n) {
    if (!c) return gpry_fail(nullptr, -1, "gpry_comm_allreduce_max: communicator is NULL");
    gpry_ctx* ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GPRY_TRY(comm_buffers(c, n * 8, n * 8));
    hipStream_t st = c->stream;
    HIP_TRY(ctx, hipMemcpyAsync(c->dsend, inout, (size_t)n * 8, hipMemcpyHostToDevice, st));
    NCCL_TRY(ctx, ncclAllReduce(c->dsend, c->drecv, (size_t)n, ncclDouble, ncclMax, c->comm, st));
    HIP_TRY(ctx, hipMemcpyAsync(inout, c->drecv, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int gpry_comm_barrier(gpry_comm* c) {
    if (!c) return gpry_fail(nullptr, -1, "gpry_comm_barrier: communicator is NULL");
    double x = 0.0;
    return gpry_comm_allreduce_max(c, &x, 1);
}

}  // extern "C"
