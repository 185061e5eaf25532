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
    // pb_comm_deblur_from_root: the exchange runs on a stream of its own beside the context's (compute) stream
    hipStream_t xs = nullptr;
    hipEvent_t ev_ready = nullptr, ev_all = nullptr, ev_arrived[3] = {nullptr, nullptr, nullptr}, ev_done[3] = {nullptr, nullptr, nullptr};
};

namespace {
bool known_dtype(int dtype) { return dtype == PB_F32 || dtype == PB_F16 || dtype == PB_U8; }
int make_exchange_stream(pb_comm *c) {
    pb_ctx *ctx = c->ctx;
    if (c->xs) return PB_OK;
    PB_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    PB_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    PB_HIP(hipEventCreateWithFlags(&c->ev_all, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) {
        PB_HIP(hipEventCreateWithFlags(&c->ev_arrived[i], hipEventDisableTiming));
        PB_HIP(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
    }
    return PB_OK;
}
}  // namespace

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
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->comm) {
        (void)hipStreamSynchronize(c->ctx->stream);
        if (Rccl *r = rccl()) (void)r->CommDestroy(c->comm);
    }
    if (c->xs) {
        (void)hipStreamDestroy(c->xs);
        (void)hipEventDestroy(c->ev_ready); (void)hipEventDestroy(c->ev_all);
        for (int i = 0; i < 3; ++i) { (void)hipEventDestroy(c->ev_arrived[i]); (void)hipEventDestroy(c->ev_done[i]); }
    }
    delete c;
    return PB_OK;
}

// scatter == true: root's batch -> every rank's shard; false: every rank's shard -> root's batch.  On the context's stream.
static int exchange(pb_comm *c, const void *root_batch_in, void *root_batch_out, const void *shard_in, void *shard_out, int dtype, int B,
                    int C, int H, int W, int root, bool scatter) {
    if (!c || B < 1 || C < 1 || H < 1 || W < 1 || root < 0 || root >= c->world) return PB_ERR_BADARG;
    pb_ctx *ctx = c->ctx;
    if (!known_dtype(dtype)) return pb_fail(ctx, PB_ERR_BADARG, "pb_comm: dtype must be PB_F32, PB_F16 or PB_U8");
    PB_HIP(hipSetDevice(ctx->device));
    const size_t img = (size_t)C * H * W, es = elem_size(dtype);
    int first = 0, count = 0;
    pb_comm_shard(B, c->world, c->rank, &first, &count);
    // (every argument is checked BEFORE anything is posted: a rank that bailed out between ncclGroupStart and its
    // send / recv would leave its peers' grouped operations hanging on their streams)
    if (c->rank != root && count > 0 && ((scatter && !shard_out) || (!scatter && !shard_in)))
        return pb_fail(ctx, PB_ERR_BADARG, "pb_comm: this rank's shard has %d image(s) but its buffer is NULL", count);
    if (c->rank == root && count > 0 && ((scatter && !shard_out) || (!scatter && !shard_in)))
        return pb_fail(ctx, PB_ERR_BADARG, "pb_comm: the root's own shard has %d image(s) but its buffer is NULL", count);
    if (c->rank == root) {
        if ((scatter && !root_batch_in) || (!scatter && !root_batch_out)) return pb_fail(ctx, PB_ERR_BADARG, "the root passes the whole batch");
        // the root's own shard: a device copy
        if (count > 0) {
            if (scatter) { PB_HIP(hipMemcpyAsync(shard_out, static_cast<const char *>(root_batch_in) + first * img * es, count * img * es, hipMemcpyDeviceToDevice, ctx->stream)); }
            else { PB_HIP(hipMemcpyAsync(static_cast<char *>(root_batch_out) + first * img * es, shard_in, count * img * es, hipMemcpyDeviceToDevice, ctx->stream)); }
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
        if (scatter) rc = r->Recv(shard_out, count * img, nccl_type(dtype), root, c->comm, ctx->stream);
        else rc = r->Send(shard_in, count * img, nccl_type(dtype), root, c->comm, ctx->stream);
    }
    const ncclResult_t rc2 = r->GroupEnd();
    if (rc != 0 || rc2 != 0) return pb_fail(ctx, PB_ERR_HIP, "RCCL exchange failed: %s", r->GetErrorString ? r->GetErrorString(rc ? rc : rc2) : "");
    return PB_OK;
}

// ---- the overlapped pattern: image by image, transfers beside the compute --------------------------------------------
// Exchange step t is ONE grouped operation in which the root sends image t of every peer's shard and receives result
// t - 2 from every peer: its xGMI links carry traffic concurrently, and every peer has image t arriving and result t - 2
// leaving while it deblurs image t - 1.  Both sides enumerate a step's operations in the same order (no tags: RCCL
// ignores them) -- pb_comm_plan is that order, the one polyblur_amd/distributed.py:exchange_plan states in Python (the two
// are compared for every (B, world, root, rank, step) on the CPU).
int pb_comm_plan_steps(int B, int world, int root) {
    if (B < 0 || world < 1 || root < 0 || root >= world) return PB_ERR_BADARG;
    int mx = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root) continue;
        int f = 0, n = 0;
        pb_comm_shard(B, world, r, &f, &n);
        if (n > mx) mx = n;
    }
    return mx > 0 ? mx + 2 : 0;
}

// ops[3 i + 0..2] = { 1 send / 0 recv, peer, image index }; at most 2 (world - 1) operations on the root, 2 elsewhere
int pb_comm_plan(int B, int world, int root, int rank, int step, int *ops, int *n_ops) {
    if (B < 0 || world < 1 || root < 0 || root >= world || rank < 0 || rank >= world || step < 0 || !ops || !n_ops) return PB_ERR_BADARG;
    int n = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root || (rank != root && r != rank)) continue;
        int lo = 0, cnt = 0;
        pb_comm_shard(B, world, r, &lo, &cnt);
        const int hi = lo + cnt, peer = rank == root ? r : root;
        if (lo + step < hi) { ops[3 * n] = rank == root ? 1 : 0; ops[3 * n + 1] = peer; ops[3 * n + 2] = lo + step; ++n; }
        if (step >= 2 && lo + step - 2 < hi) { ops[3 * n] = rank == root ? 0 : 1; ops[3 * n + 1] = peer; ops[3 * n + 2] = lo + step - 2; ++n; }
    }
    *n_ops = n;
    return PB_OK;
}

int pb_comm_deblur_from_root(pb_comm *c, const void *root_batch, void *root_out, int dtype, int B, int C, int H, int W,
                             const pb_options *opt, int root) {
    if (!c || !opt || B < 1 || C < 1 || H < 1 || W < 1 || root < 0 || root >= c->world) return PB_ERR_BADARG;
    pb_ctx *ctx = c->ctx;
    if (!known_dtype(dtype)) return pb_fail(ctx, PB_ERR_BADARG, "pb_comm: dtype must be PB_F32, PB_F16 or PB_U8");
    if (c->rank == root && (!root_batch || !root_out || root_batch == root_out))
        return pb_fail(ctx, PB_ERR_BADARG, "pb_comm_deblur_from_root: the root passes the whole batch and a distinct output batch");
    PB_HIP(hipSetDevice(ctx->device));
    const size_t img = (size_t)C * H * W, ib = img * elem_size(dtype);
    int lo = 0, cnt = 0;
    pb_comm_shard(B, c->world, c->rank, &lo, &cnt);
    if (c->world == 1)
        return pb_polyblur_batch(ctx, root_batch, root_out, dtype, B, C, H, W, opt, nullptr);
    Rccl *r = rccl();
    if (!r) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "pb_comm: librccl.so could not be loaded");
    int rc = make_exchange_stream(c);
    if (rc) return rc;
    const int nsteps = pb_comm_plan_steps(B, c->world, root);
    std::vector<int> ops(6 * (size_t)c->world);
    // whatever produced the batch (or last used the ring buffers) on the compute stream comes first
    PB_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    PB_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
    auto post = [&](int step, char *const *recv_ring, char *const *send_ring) -> int {
        int n = 0;
        pb_comm_plan(B, c->world, root, c->rank, step, ops.data(), &n);
        if (!n) return PB_OK;
        ncclResult_t e = r->GroupStart();
        for (int i = 0; i < n && e == 0; ++i) {
            const int send = ops[3 * i], peer = ops[3 * i + 1], k = ops[3 * i + 2];
            if (c->rank == root) {
                if (send) e = r->Send(static_cast<const char *>(root_batch) + (size_t)k * ib, img, nccl_type(dtype), peer, c->comm, c->xs);
                else e = r->Recv(static_cast<char *>(root_out) + (size_t)k * ib, img, nccl_type(dtype), peer, c->comm, c->xs);
            } else {
                if (send) e = r->Send(send_ring[(k - lo) % 3], img, nccl_type(dtype), peer, c->comm, c->xs);
                else e = r->Recv(recv_ring[(k - lo) % 3], img, nccl_type(dtype), peer, c->comm, c->xs);
            }
        }
        const ncclResult_t e2 = r->GroupEnd();
        if (e != 0 || e2 != 0) return pb_fail(ctx, PB_ERR_HIP, "RCCL exchange step %d failed: %s", step, r->GetErrorString ? r->GetErrorString(e ? e : e2) : "");
        return PB_OK;
    };
    if (c->rank == root) {
        // the root's own shard is spread over the steps; its transfers read root_batch and write other images of root_out
        const int per_step = nsteps ? (cnt + nsteps - 1) / nsteps : cnt;
        int done = 0;
        for (int t = 0; t < nsteps; ++t) {
            rc = post(t, nullptr, nullptr);
            if (rc) return rc;
            for (int j = 0; j < per_step && done < cnt; ++j, ++done) {
                rc = pb_polyblur_batch(ctx, static_cast<const char *>(root_batch) + (size_t)(lo + done) * ib,
                                       static_cast<char *>(root_out) + (size_t)(lo + done) * ib, dtype, 1, C, H, W, opt, nullptr);
                if (rc) return rc;
            }
        }
        for (; done < cnt; ++done) {
            rc = pb_polyblur_batch(ctx, static_cast<const char *>(root_batch) + (size_t)(lo + done) * ib,
                                   static_cast<char *>(root_out) + (size_t)(lo + done) * ib, dtype, 1, C, H, W, opt, nullptr);
            if (rc) return rc;
        }
    } else {
        // ring of three: arriving / in work / leaving
        char *in_ring[3], *out_ring[3];
        for (int i = 0; i < 3; ++i) {
            const std::string a = "comm.in" + std::to_string(i), b = "comm.out" + std::to_string(i);
            in_ring[i] = static_cast<char *>(pb_scratch(ctx, a.c_str(), ib));
            out_ring[i] = static_cast<char *>(pb_scratch(ctx, b.c_str(), ib));
            if (!in_ring[i] || !out_ring[i]) return PB_ERR_NOMEM;
        }
        for (int t = 0; t < nsteps; ++t) {
            // step t sends the result of image t - 2 (deblurred in step t - 1) and receives image t into the buffer image
            // t - 3 was deblurred from: both computes are behind ev_done of the younger one
            if (t >= 2 && t - 2 < cnt) PB_HIP(hipStreamWaitEvent(c->xs, c->ev_done[(t - 2) % 3], 0));
            rc = post(t, in_ring, out_ring);
            if (rc) return rc;
            PB_HIP(hipEventRecord(c->ev_arrived[t % 3], c->xs));
            const int i = t - 1;                                    // image i arrived in step t - 1: deblur it now
            if (i >= 0 && i < cnt) {
                // (out_ring[i % 3] last held result i - 3, sent in step i - 1: complete before step i's event)
                PB_HIP(hipStreamWaitEvent(ctx->stream, c->ev_arrived[i % 3], 0));
                rc = pb_polyblur_batch(ctx, in_ring[i % 3], out_ring[i % 3], dtype, 1, C, H, W, opt, nullptr);
                if (rc) return rc;
                PB_HIP(hipEventRecord(c->ev_done[i % 3], ctx->stream));
            }
        }
    }
    // the call returns with the compute stream behind every transfer (the root's output batch is complete there)
    PB_HIP(hipEventRecord(c->ev_all, c->xs));
    PB_HIP(hipStreamWaitEvent(ctx->stream, c->ev_all, 0));
    return PB_OK;
}

int pb_comm_scatter(pb_comm *c, const void *root_batch, void *shard, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, root_batch, nullptr, nullptr, shard, dtype, B, C, H, W, root, true);
}

int pb_comm_gather(pb_comm *c, const void *shard, void *root_batch, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, nullptr, root_batch, shard, nullptr, dtype, B, C, H, W, root, false);
}

}  // extern "C"
