// Batch scatter / gather over RCCL for hosts that do not have torch.distributed (pb_comm_*).
//
// The reference has no communication at all (its only batching is a sequential loop over patch groups,
// deblurring.py:310-336); images are independent, so the engine's one exchange pattern is: the batch lives on a root
// rank, every rank deblurs a contiguous shard (first B mod n ranks get one extra image -- the same rule as
// polyblur_amd/distributed.py:shard_bounds), the results return to the root.  One process per GPU; the data path is
// grouped ncclSend / ncclRecv (shards are uneven, so not ncclScatter), one group per call, over xGMI inside a node.
//
// RCCL is loaded with dlopen when the first communicator is made: the engine itself does not depend on librccl.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace {

// the slice of rccl.h this file needs (ROCm 7.2: /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,459-465,700,722,923)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[PB_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1, ncclFloat32 = 7, ncclFloat16 = 6 };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (r.handle) {
#define PB_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
            PB_SYM(GetUniqueId, "ncclGetUniqueId"); PB_SYM(CommInitRank, "ncclCommInitRank"); PB_SYM(CommDestroy, "ncclCommDestroy");
            PB_SYM(Send, "ncclSend"); PB_SYM(Recv, "ncclRecv"); PB_SYM(GroupStart, "ncclGroupStart"); PB_SYM(GroupEnd, "ncclGroupEnd");
            PB_SYM(GetErrorString, "ncclGetErrorString");
#undef PB_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
                dlclose(r.handle);
                r.handle = nullptr;
            }
        }
    }
    return r.handle ? &r : nullptr;
}

size_t elem_size(int dtype) { return dtype == PB_F16 ? 2 : (dtype == PB_U8 ? 1 : 4); }
int nccl_type(int dtype) { return dtype == PB_F16 ? ncclFloat16 : (dtype == PB_U8 ? ncclUint8 : ncclFloat32); }

}  // namespace

struct pb_comm {
    pb_ctx *ctx;
    ncclComm_t comm;     // nullptr for a world of one (plain device copies)
    int rank, world;
};

extern "C" {

// first image and number of images of `rank`'s shard of a batch of B (contiguous; the first B % world ranks get one more)
int pb_comm_shard(int B, int world, int rank, int *first, int *count) {
    if (B < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) return PB_ERR_BADARG;
    const int base = B / world, extra = B % world;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + (rank < extra ? rank : extra);
    return PB_OK;
}

int pb_comm_unique_id(unsigned char *id) {
    if (!id) return PB_ERR_BADARG;
    Rccl *r = rccl();
    if (!r) return PB_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != 0) return PB_ERR_HIP;
    memcpy(id, u.internal, PB_COMM_ID_BYTES);
    return PB_OK;
}

int pb_comm_init(pb_comm **out, pb_ctx *ctx, int rank, int world, const unsigned char *id) {
    if (!out || !ctx || world < 1 || rank < 0 || rank >= world) return PB_ERR_BADARG;
    *out = nullptr;
    pb_comm *c = new pb_comm{ctx, nullptr, rank, world};
    if (world > 1) {
        if (!id) { delete c; return pb_fail(ctx, PB_ERR_BADARG, "pb_comm_init: a world of %d needs the unique id of pb_comm_unique_id", world); }
        Rccl *r = rccl();
        if (!r) { delete c; return pb_fail(ctx, PB_ERR_UNSUPPORTED, "pb_comm_init: librccl.so could not be loaded"); }
        PB_HIP(hipSetDevice(ctx->device));
        ncclUniqueId u;
        memcpy(u.internal, id, PB_COMM_ID_BYTES);
        const ncclResult_t rc = r->CommInitRank(&c->comm, world, u, rank);
        if (rc != 0) {
            delete c;
            return pb_fail(ctx, PB_ERR_HIP, "ncclCommInitRank: %s", r->GetErrorString ? r->GetErrorString(rc) : "failed");
        }
    }
    *out = c;
    return PB_OK;
}

int pb_comm_destroy(pb_comm *c) {
    if (!c) return PB_OK;
    if (c->comm) {
        (void)hipStreamSynchronize(c->ctx->stream);
        if (Rccl *r = rccl()) (void)r->CommDestroy(c->comm);
    }
    delete c;
    return PB_OK;
}

// scatter == true: root's batch -> every rank's shard; false: every rank's shard -> root's batch.  On the context's stream.
static int exchange(pb_comm *c, const void *root_batch_in, void *root_batch_out, const void *shard_in, void *shard_out, int dtype, int B,
                    int C, int H, int W, int root, bool scatter) {
    if (!c || B < 1 || C < 1 || H < 1 || W < 1 || root < 0 || root >= c->world) return PB_ERR_BADARG;
    pb_ctx *ctx = c->ctx;
    PB_HIP(hipSetDevice(ctx->device));
    const size_t img = (size_t)C * H * W, es = elem_size(dtype);
    int first = 0, count = 0;
    pb_comm_shard(B, c->world, c->rank, &first, &count);
    if (c->rank == root) {
        if ((scatter && !root_batch_in) || (!scatter && !root_batch_out)) return pb_fail(ctx, PB_ERR_BADARG, "the root passes the whole batch");
        // the root's own shard: a device copy
        if (count > 0) {
            if (scatter) { if (!shard_out) return PB_ERR_BADARG; PB_HIP(hipMemcpyAsync(shard_out, static_cast<const char *>(root_batch_in) + first * img * es, count * img * es, hipMemcpyDeviceToDevice, ctx->stream)); }
            else { if (!shard_in) return PB_ERR_BADARG; PB_HIP(hipMemcpyAsync(static_cast<char *>(root_batch_out) + first * img * es, shard_in, count * img * es, hipMemcpyDeviceToDevice, ctx->stream)); }
        }
    }
    if (c->world == 1) return PB_OK;
    Rccl *r = rccl();
    if (!r) return PB_ERR_UNSUPPORTED;
    ncclResult_t rc = r->GroupStart();
    if (c->rank == root) {
        for (int p = 0; p < c->world && rc == 0; ++p) {
            if (p == root) continue;
            int pf = 0, pc = 0;
            pb_comm_shard(B, c->world, p, &pf, &pc);
            if (pc == 0) continue;
            if (scatter) rc = r->Send(static_cast<const char *>(root_batch_in) + pf * img * es, pc * img, nccl_type(dtype), p, c->comm, ctx->stream);
            else rc = r->Recv(static_cast<char *>(root_batch_out) + pf * img * es, pc * img, nccl_type(dtype), p, c->comm, ctx->stream);
        }
    } else if (count > 0) {
        if (scatter) { if (!shard_out) rc = -1; else rc = r->Recv(shard_out, count * img, nccl_type(dtype), root, c->comm, ctx->stream); }
        else { if (!shard_in) rc = -1; else rc = r->Send(shard_in, count * img, nccl_type(dtype), root, c->comm, ctx->stream); }
    }
    const ncclResult_t rc2 = r->GroupEnd();
    if (rc != 0 || rc2 != 0) return pb_fail(ctx, PB_ERR_HIP, "RCCL exchange failed: %s", r->GetErrorString ? r->GetErrorString(rc ? rc : rc2) : "");
    return PB_OK;
}

int pb_comm_scatter(pb_comm *c, const void *root_batch, void *shard, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, root_batch, nullptr, nullptr, shard, dtype, B, C, H, W, root, true);
}

int pb_comm_gather(pb_comm *c, const void *shard, void *root_batch, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, nullptr, root_batch, shard, nullptr, dtype, B, C, H, W, root, false);
}

}  // extern "C"
