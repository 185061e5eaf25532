"""Batch sharding across the GPUs of one node.

Images are independent (every reduction in the pipeline is per image), so the batch dimension is
split into contiguous shards, one per rank (= one process per GPU): the first ``B mod n`` ranks get
one extra image.  There is no data-path collective inside the hot path; RCCL over xGMI (the
``nccl`` backend of torch.distributed on ROCm) is used only to move shards from / to the root
rank.

Two usage patterns:
  * shards already resident on every GPU (what ``bench.py`` times by default, "weak" scaling):
        out_local = polyblur_deblurring(x_local, ...)
  * a whole batch on the root rank (``bench.py --mode from_root``):
        out = deblur_from_root(x_or_None, shape, dtype, ...)      # returns the full batch on root

``deblur_from_root`` moves one CHUNK of k consecutive images per peer per step, all peers at once:
step t is ONE grouped point-to-point exchange (``batch_isend_irecv`` = ncclGroupStart/End on RCCL) in
which the root sends chunk t of every peer's shard and receives result chunk t-2 from every peer, so
the root's seven xGMI links carry traffic concurrently and each peer has chunk t arriving and result
t-2 leaving while it deblurs chunk t-1 -- as ONE batch of k images -- on its compute stream.  k = 1
is the image-by-image exchange of rounds 2-4 and **the default** (``chunk=None``): the chunked exchange
has never run on more than one GPU (one-GPU lease: gloo runs and one-rank nccl runs only), so it is
opt-in until a two-GPU run of tests/test_gpu_rccl.py has passed -- ``chunk=k`` (the same k on every
rank), ``chunk="auto"`` (``default_chunk``: k ~ sqrt(shard / 2), the rule for "a lone call costs twice
what an image costs inside a batch": 4 for the 32 images per GPU of BASELINE config 4, 1 for config 5's
single 8K image) or ``chunk="measure"`` (``measured_chunk``: the same minimisation with the per-image
times of THIS engine on THIS image size, taken from two timed calls on the root and broadcast).
Nothing relies on message tags (RCCL ignores them): both sides enumerate the exchanges of a step in
the same order.  Results land directly in the output batch (no staging buffers on the root).

Errors: a ``compute`` that raises on one rank must not leave the others waiting for a matching send /
recv.  Every rank therefore posts EVERY remaining step after its first failure (buffers whose content
no longer matters), then all ranks agree on a status word (one 4-byte all-reduce, control plane) and
every rank raises -- the failing one its own exception, the others ``RuntimeError`` naming the rank.
pb_comm_deblur_from_root drains the same way in C.

The compute function is a parameter so that the sharding logic can be exercised on CPU (gloo)
without a GPU; in production it is polyblur_amd.polyblur_deblurring.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous shard; the first (batch % world) ranks hold one extra image."""
    if world < 1 or not (0 <= rank < world) or batch < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(batch: int, world: int):
    return [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]


def default_chunk(batch: int, world: int, root: int = 0) -> int:
    """Images per exchange step: ~sqrt(largest peer shard / 2), at least 1 (pb_comm_default_chunk states the same rule in C)."""
    sizes = [n for r, n in enumerate(shard_sizes(batch, world)) if r != root]
    n = max(sizes) if sizes else 0
    k = 1
    while 2 * (k + 1) * (k + 1) <= n:
        k += 1
    return k


def measured_chunk(shard: int, t_alone: float, t_in_batch: float) -> int:
    """Images per exchange step from MEASURED call times: a step of k images costs ~ t_alone + (k - 1) t_in_batch and a shard
    of n images takes n / k + 2 steps, so the total is least at k = sqrt(n (t_alone - t_in_batch) / (2 t_in_batch)) (which is
    default_chunk's sqrt(n / 2) when a lone call costs twice an image inside a batch).  At least 1, at most the shard."""
    if shard <= 1 or not (t_alone > 0.0 and t_in_batch > 0.0) or t_alone <= t_in_batch:
        return 1
    k = int(round((shard * (t_alone - t_in_batch) / (2.0 * t_in_batch)) ** 0.5))
    return max(1, min(int(shard), k))


def exchange_plan(batch: int, world: int, root: int, rank: int, step: int, chunk: int = 1):
    """The point-to-point operations of `rank` in exchange step `step`, as (kind, peer, first image, count) -- for chunk == 1
    (kind, peer, image index) -- in the order both sides enumerate them: the root's list for a step, filtered to one peer, is
    that peer's list with send and recv swapped.  Chunk t of a shard travels root -> peer in step t, its results come back
    in step t + 2."""
    if chunk < 1:
        raise ValueError("chunk must be >= 1")
    ops = []
    peers = [r for r in range(world) if r != root] if rank == root else [rank]
    for r in peers:
        lo, hi = shard_bounds(batch, world, r)
        peer = r if rank == root else root
        a = lo + step * chunk
        if a < hi:
            op = ("send" if rank == root else "recv", peer, a) + ((min(chunk, hi - a),) if chunk > 1 else ())
            ops.append(op)
        b = lo + (step - 2) * chunk
        if step >= 2 and b < hi:
            op = ("recv" if rank == root else "send", peer, b) + ((min(chunk, hi - b),) if chunk > 1 else ())
            ops.append(op)
    return ops


def exchange_steps(batch: int, world: int, root: int, chunk: int = 1) -> int:
    """Number of exchange steps: the chunks of the largest peer shard plus the two-step return lag (0 without peers)."""
    sizes = [n for r, n in enumerate(shard_sizes(batch, world)) if r != root]
    return (-(-max(sizes) // chunk) + 2) if sizes and max(sizes) > 0 else 0


def deblur_from_root(images, shape: Sequence[int], dtype, compute: Optional[Callable] = None, device=None,
                     root: int = 0, group=None, chunk=None, check: bool = True, **kwargs):
    """Scatter a (B,C,H,W) batch that lives on `root`, deblur every shard where it lands, gather
    the result on `root` (other ranks return None).  `images` is ignored on non-root ranks;
    `shape` / `dtype` must be given on all ranks; `chunk` = images per exchange step, the same value on every rank:
    None or 1 = image by image (the default), an int k >= 1, "auto" (default_chunk) or "measure" (measured_chunk from two
    timed calls on the root).  `check`: agree on a status word afterwards so that a failure on one rank raises on all."""
    import time
    import torch
    import torch.distributed as dist
    if compute is None:
        from .deblurring import polyblur_deblurring as compute
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = int(shape[0])
    lo, hi = shard_bounds(B, world, rank)
    if isinstance(chunk, bool) or not (chunk is None or chunk in ("auto", "measure") or isinstance(chunk, int)):
        raise ValueError("chunk must be None, an int >= 1, 'auto' or 'measure'")
    if isinstance(chunk, int) and chunk < 1:
        raise ValueError("chunk must be >= 1 (got %d)" % chunk)           # (the same on every rank: nobody has posted anything yet)
    if device is None:
        device = images.device if (rank == root and images is not None) else torch.device("cpu")
    first_exc = None                                                       # this rank's first failure; every step is still posted
    root_ok = rank != root or (images is not None and tuple(images.shape) == tuple(shape))
    if not root_ok:
        first_exc = ValueError("root must pass the full batch with the announced shape")
    peer_max = max([n for r, n in enumerate(shard_sizes(B, world)) if r != root] or [0])
    if chunk is None:
        k = 1
    elif chunk == "auto":
        k = default_chunk(B, world, root)
    elif chunk == "measure":
        # the engine's own times on this image size: one image alone, and per image inside a batch of (up to) four
        kt = torch.ones(1, dtype=torch.int32, device=device)
        if rank == root and root_ok and world > 1 and peer_max > 1:
            try:
                def clock(n):
                    compute(images[:n], **kwargs)                          # (first call: plans, scratch)
                    if images.is_cuda:
                        torch.cuda.synchronize(images.device)
                    t0 = time.perf_counter()
                    compute(images[:n], **kwargs)
                    if images.is_cuda:
                        torch.cuda.synchronize(images.device)
                    return time.perf_counter() - t0
                nb = min(4, B)
                t1, tn = clock(1), clock(nb)
                kt[0] = measured_chunk(peer_max, t1, (tn - t1) / (nb - 1) if nb > 1 else t1)
            except Exception as e:                                         # noqa: BLE001 -- reported after the exchange
                first_exc = e
        if world > 1:
            dist.broadcast(kt, root, group)
        k = max(1, int(kt.item()))
    else:
        k = int(chunk)
    nsteps = exchange_steps(B, world, root, k)
    chunk_shape = (k,) + tuple(int(v) for v in shape[1:])
    dummy = None

    def dead(cnt):                                                         # a buffer for the steps after a failure
        nonlocal dummy
        if dummy is None:
            dummy = torch.zeros(chunk_shape, dtype=dtype, device=device)
        return dummy[:cnt]

    def plan(t):
        return [(op[0], op[1], op[2], op[3] if len(op) > 3 else 1) for op in exchange_plan(B, world, root, rank, t, k)]

    def post(ops, tensor_of):
        if not ops:
            return []
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, tensor_of(kind, a, n), peer, group) for kind, peer, a, n in ops]
        return dist.batch_isend_irecv(p2p)

    def guarded(fn):                                                       # compute nothing after the first failure
        nonlocal first_exc
        if first_exc is not None:
            return None
        try:
            return fn()
        except Exception as e:                                             # noqa: BLE001 -- kept, raised after the drain
            first_exc = e
            return None

    out_full = None
    if rank == root:
        if root_ok:
            images = images.contiguous()
            out_full = torch.empty_like(images)
        pending = []
        own = list(range(lo, hi))
        per_step = -(-len(own) // nsteps) if nsteps else len(own)

        def root_tensor(kind, a, n):
            if not root_ok:
                return dead(n)
            return images[a:a + n] if kind == "send" else out_full[a:a + n]

        def own_batch(mine):
            def run():
                out_full[mine[0]:mine[-1] + 1] = compute(images[mine[0]:mine[-1] + 1], **kwargs)
            if mine:
                guarded(run)

        for t in range(nsteps):
            pending += post(plan(t), root_tensor)
            own_batch(own[t * per_step:(t + 1) * per_step])                # the root's own shard, spread over the steps, a batch per step
        own_batch(own[nsteps * per_step:])
        for w in pending:
            w.wait()
    else:
        n = hi - lo
        nchunks = -(-n // k) if n else 0
        bufs = [torch.empty(chunk_shape, dtype=dtype, device=device) for _ in range(min(nchunks, 3))]   # ring: arriving / in work / spare
        res = {}
        works = {}
        for t in range(nsteps):
            works[t] = post(plan(t), lambda kind, a, cnt: bufs[((a - lo) // k) % 3][:cnt] if kind == "recv" else res[(a - lo) // k])
            i = t - 1                                                   # chunk i arrived in step t-1: deblur it now, as one batch
            if 0 <= i < nchunks:
                for w in works.pop(t - 1):
                    w.wait()
                cnt = min(k, n - i * k)
                r_i = guarded(lambda: compute(bufs[i % 3][:cnt], **kwargs).contiguous())
                res[i] = r_i if r_i is not None else dead(cnt)          # (after a failure: the step's send still has a buffer)
                res.pop(i - 3, None)                                    # its send (step i+1) was waited for in step i+2
        for ws in works.values():
            for w in ws:
                w.wait()
    if check and world > 1:
        # one status word: the lowest failing rank + 1, 0 if none (control plane; the data path has no collective)
        st = torch.tensor([-(rank + 1) if first_exc is not None else -(world + 1)], dtype=torch.int32, device=device)
        dist.all_reduce(st, op=dist.ReduceOp.MAX, group=group)
        bad = -int(st.item()) - 1
        if first_exc is None and bad < world:
            raise RuntimeError("deblur_from_root: rank %d failed; the batch on rank %d is incomplete" % (bad, root))
    if first_exc is not None:
        raise first_exc
    return out_full if rank == root else None
