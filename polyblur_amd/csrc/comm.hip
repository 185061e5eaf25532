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

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

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
    int chunk = 0;       // images per exchange step of pb_comm_deblur_from_root (pb_comm_set_chunk); 0 = image by image, PB_COMM_CHUNK_AUTO = pb_comm_default_chunk
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

// ---- the overlapped pattern: chunk by chunk, transfers beside the compute ---------------------------------------------
// Exchange step t is ONE grouped operation in which the root sends chunk t (k consecutive images) of every peer's shard
// and receives result chunk t - 2 from every peer: its xGMI links carry traffic concurrently, and every peer has chunk t
// arriving and result t - 2 leaving while it deblurs chunk t - 1 as one batch of k images.  Both sides enumerate a step's
// operations in the same order (no tags: RCCL ignores them) -- pb_comm_plan_chunked is that order, the one
// polyblur_amd/distributed.py:exchange_plan states in Python (the two are compared for every (B, world, root, rank, step, k) on
// the CPU).  k = 1 is the image-by-image exchange of rounds 2 - 4; a lone 1080p call costs twice what an image costs inside
// a batch, hence pb_comm_default_chunk ~ sqrt(shard / 2) -- opt-in (pb_comm_set_chunk(c, PB_COMM_CHUNK_AUTO) or k > 1): the default
// stays k = 1 until the chunked exchange has run on two GPUs.  UNMEASURED on more than one GPU (one-GPU lease).
int pb_comm_default_chunk(int B, int world, int root) {
    if (B < 0 || world < 1 || root < 0 || root >= world) return PB_ERR_BADARG;
    int mx = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root) continue;
        int f = 0, n = 0;
        pb_comm_shard(B, world, r, &f, &n);
        if (n > mx) mx = n;
    }
    int k = 1;
    while (2 * (k + 1) * (k + 1) <= mx) ++k;
    return k;
}

int pb_comm_plan_steps_chunked(int B, int world, int root, int chunk) {
    if (B < 0 || world < 1 || root < 0 || root >= world || chunk < 1) return PB_ERR_BADARG;
    int mx = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root) continue;
        int f = 0, n = 0;
        pb_comm_shard(B, world, r, &f, &n);
        if (n > mx) mx = n;
    }
    return mx > 0 ? (mx + chunk - 1) / chunk + 2 : 0;
}
int pb_comm_plan_steps(int B, int world, int root) { return pb_comm_plan_steps_chunked(B, world, root, 1); }

// ops[4 i + 0..3] = { 1 send / 0 recv, peer, first image, images }; at most 2 (world - 1) operations on the root, 2 elsewhere
int pb_comm_plan_chunked(int B, int world, int root, int rank, int step, int chunk, int *ops, int *n_ops) {
    if (B < 0 || world < 1 || root < 0 || root >= world || rank < 0 || rank >= world || step < 0 || chunk < 1 || !ops || !n_ops) return PB_ERR_BADARG;
    int n = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root || (rank != root && r != rank)) continue;
        int lo = 0, cnt = 0;
        pb_comm_shard(B, world, r, &lo, &cnt);
        const int hi = lo + cnt, peer = rank == root ? r : root;
        const long a = (long)lo + (long)step * chunk, b = (long)lo + (long)(step - 2) * chunk;
        if (a < hi) { ops[4 * n] = rank == root ? 1 : 0; ops[4 * n + 1] = peer; ops[4 * n + 2] = (int)a; ops[4 * n + 3] = (int)std::min<long>(chunk, hi - a); ++n; }
        if (step >= 2 && b < hi) { ops[4 * n] = rank == root ? 0 : 1; ops[4 * n + 1] = peer; ops[4 * n + 2] = (int)b; ops[4 * n + 3] = (int)std::min<long>(chunk, hi - b); ++n; }
    }
    *n_ops = n;
    return PB_OK;
}
// (the image-by-image plan: ops[3 i + 0..2] = { 1 send / 0 recv, peer, image index })
int pb_comm_plan(int B, int world, int root, int rank, int step, int *ops, int *n_ops) {
    if (world < 1 || !ops || !n_ops) return PB_ERR_BADARG;
    std::vector<int> o4(8 * (size_t)world);
    const int rc = pb_comm_plan_chunked(B, world, root, rank, step, 1, o4.data(), n_ops);
    if (rc) return rc;
    for (int i = 0; i < *n_ops; ++i) { ops[3 * i] = o4[4 * i]; ops[3 * i + 1] = o4[4 * i + 1]; ops[3 * i + 2] = o4[4 * i + 2]; }
    return PB_OK;
}

int pb_comm_set_chunk(pb_comm *c, int chunk) {
    if (!c || chunk < PB_COMM_CHUNK_AUTO) return PB_ERR_BADARG;
    c->chunk = chunk;
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
    // image by image unless the caller opted in: the chunked exchange has not run on two GPUs yet
    const int k = c->chunk > 0 ? c->chunk : (c->chunk == PB_COMM_CHUNK_AUTO ? pb_comm_default_chunk(B, c->world, root) : 1);
    const int nsteps = pb_comm_plan_steps_chunked(B, c->world, root, k);
    std::vector<int> ops(8 * (size_t)c->world);
    // every buffer the steps will name exists BEFORE anything is posted: a rank that bailed out between two steps would leave
    // its peers' grouped operations hanging on their streams
    char *in_ring[3] = {nullptr, nullptr, nullptr}, *out_ring[3] = {nullptr, nullptr, nullptr};
    if (c->rank != root && cnt > 0) {
        for (int i = 0; i < 3; ++i) {
            const std::string a = "comm.in" + std::to_string(i), b = "comm.out" + std::to_string(i);
            in_ring[i] = static_cast<char *>(pb_scratch(ctx, a.c_str(), ib * k));
            out_ring[i] = static_cast<char *>(pb_scratch(ctx, b.c_str(), ib * k));
            if (!in_ring[i] || !out_ring[i]) return PB_ERR_NOMEM;
        }
    }
    // whatever produced the batch (or last used the ring buffers) on the compute stream comes first
    PB_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    PB_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
    // From here on every step IS posted whatever fails in between -- a compute error, a HIP error: the first one is kept and
    // returned at the end, the remaining steps still move their (then meaningless) buffers, so that no peer is left waiting
    // for a matching send / recv; and the compute stream is joined behind the exchange stream on every way out.
    int first_err = PB_OK;
    std::string first_msg;
    auto keep = [&](int e) { if (e && !first_err) { first_err = e; first_msg = ctx->err; } };
    auto post = [&](int step) -> int {
        int n = 0;
        pb_comm_plan_chunked(B, c->world, root, c->rank, step, k, ops.data(), &n);
        if (!n) return PB_OK;
        ncclResult_t e = r->GroupStart();
        for (int i = 0; i < n && e == 0; ++i) {
            const int send = ops[4 * i], peer = ops[4 * i + 1], a = ops[4 * i + 2];
            const size_t count = (size_t)ops[4 * i + 3] * img;
            if (c->rank == root) {
                if (send) e = r->Send(static_cast<const char *>(root_batch) + (size_t)a * ib, count, nccl_type(dtype), peer, c->comm, c->xs);
                else e = r->Recv(static_cast<char *>(root_out) + (size_t)a * ib, count, nccl_type(dtype), peer, c->comm, c->xs);
            } else {
                const int slot = ((a - lo) / k) % 3;
                if (send) e = r->Send(out_ring[slot], count, nccl_type(dtype), peer, c->comm, c->xs);
                else e = r->Recv(in_ring[slot], count, nccl_type(dtype), peer, c->comm, c->xs);
            }
        }
        const ncclResult_t e2 = r->GroupEnd();
        if (e != 0 || e2 != 0) return pb_fail(ctx, PB_ERR_HIP, "RCCL exchange step %d failed: %s", step, r->GetErrorString ? r->GetErrorString(e ? e : e2) : "");
        return PB_OK;
    };
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) keep(pb_fail(ctx, PB_ERR_HIP, "%s failed: %s (pb_comm_deblur_from_root)", what, hipGetErrorString(e)));
    };
    auto compute = [&](const char *src, char *dst, int n) {
        if (first_err) return;                                   // (after a failure: move buffers, compute nothing)
        keep(pb_polyblur_batch(ctx, src, dst, dtype, n, C, H, W, opt, nullptr));
    };
    if (c->rank == root) {
        // the root's own shard is spread over the steps, one batch per step; its transfers read root_batch and write other
        // images of root_out
        const int per_step = nsteps ? (cnt + nsteps - 1) / nsteps : cnt;
        int done = 0;
        for (int t = 0; t < nsteps; ++t) {
            keep(post(t));
            const int n = std::min(per_step, cnt - done);
            if (n > 0) {
                compute(static_cast<const char *>(root_batch) + (size_t)(lo + done) * ib, static_cast<char *>(root_out) + (size_t)(lo + done) * ib, n);
                done += n;
            }
        }
        if (done < cnt)
            compute(static_cast<const char *>(root_batch) + (size_t)(lo + done) * ib, static_cast<char *>(root_out) + (size_t)(lo + done) * ib, cnt - done);
    } else {
        const int nchunks = (cnt + k - 1) / k;
        for (int t = 0; t < nsteps; ++t) {
            // step t sends the results of chunk t - 2 (deblurred in step t - 1) and receives chunk t into the buffer chunk
            // t - 3 was deblurred from: both computes are behind ev_done of the younger one
            if (t >= 2 && t - 2 < nchunks) hip_ok(hipStreamWaitEvent(c->xs, c->ev_done[(t - 2) % 3], 0), "hipStreamWaitEvent");
            keep(post(t));
            hip_ok(hipEventRecord(c->ev_arrived[t % 3], c->xs), "hipEventRecord");
            const int i = t - 1;                                    // chunk i arrived in step t - 1: deblur it now
            if (i >= 0 && i < nchunks) {
                // (out_ring[i % 3] last held result chunk i - 3, sent in step i - 1: complete before step i's event)
                hip_ok(hipStreamWaitEvent(ctx->stream, c->ev_arrived[i % 3], 0), "hipStreamWaitEvent");
                compute(in_ring[i % 3], out_ring[i % 3], std::min(k, cnt - i * k));
                hip_ok(hipEventRecord(c->ev_done[i % 3], ctx->stream), "hipEventRecord");
            }
        }
    }
    // the call returns with the compute stream behind every transfer (the root's output batch is complete there)
    hip_ok(hipEventRecord(c->ev_all, c->xs), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(ctx->stream, c->ev_all, 0), "hipStreamWaitEvent");
    if (first_err) ctx->err = first_msg;
    return first_err;
}

int pb_comm_scatter(pb_comm *c, const void *root_batch, void *shard, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, root_batch, nullptr, nullptr, shard, dtype, B, C, H, W, root, true);
}

int pb_comm_gather(pb_comm *c, const void *shard, void *root_batch, int dtype, int B, int C, int H, int W, int root) {
    return exchange(c, nullptr, root_batch, shard, nullptr, dtype, B, C, H, W, root, false);
}

}  // extern "C"
