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
is the image-by-image exchange of rounds 2-4; a lone 1080p call costs 0.33-0.38 ms against 0.17 ms
per image inside a batch (profiles/), so ``default_chunk`` picks k ~ sqrt(shard / 2): 4 for the 32
images per GPU of BASELINE config 4 (10 steps of ~0.9 ms instead of 34 of 0.33), 1 for config 5's
single 8K image.  Nothing relies on message tags (RCCL ignores them): both sides enumerate the
exchanges of a step in the same order.  Results land directly in the output batch (no staging
buffers on the root).  Unmeasured on N > 1 GPUs (one-GPU lease): gloo runs and one-rank nccl runs only.

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
                     root: int = 0, group=None, chunk: Optional[int] = None, **kwargs):
    """Scatter a (B,C,H,W) batch that lives on `root`, deblur every shard where it lands, gather
    the result on `root` (other ranks return None).  `images` is ignored on non-root ranks;
    `shape` / `dtype` must be given on all ranks; `chunk` = images per exchange step (None: default_chunk; the same
    value on every rank)."""
    import torch
    import torch.distributed as dist
    if compute is None:
        from .deblurring import polyblur_deblurring as compute
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = int(shape[0])
    lo, hi = shard_bounds(B, world, rank)
    k = default_chunk(B, world, root) if chunk is None else int(chunk)
    if device is None:
        device = images.device if (rank == root and images is not None) else torch.device("cpu")
    nsteps = exchange_steps(B, world, root, k)

    def plan(t):
        return [(op[0], op[1], op[2], op[3] if len(op) > 3 else 1) for op in exchange_plan(B, world, root, rank, t, k)]

    def post(ops, tensor_of):
        if not ops:
            return []
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, tensor_of(kind, a, n), peer, group) for kind, peer, a, n in ops]
        return dist.batch_isend_irecv(p2p)

    if rank == root:
        if images is None or tuple(images.shape) != tuple(shape):
            raise ValueError("root must pass the full batch with the announced shape")
        images = images.contiguous()
        out_full = torch.empty_like(images)
        pending = []
        own = list(range(lo, hi))
        per_step = -(-len(own) // nsteps) if nsteps else len(own)
        for t in range(nsteps):
            pending += post(plan(t), lambda kind, a, n: images[a:a + n] if kind == "send" else out_full[a:a + n])
            mine = own[t * per_step:(t + 1) * per_step]            # the root's own shard, spread over the steps, a batch per step
            if mine:
                out_full[mine[0]:mine[-1] + 1] = compute(images[mine[0]:mine[-1] + 1], **kwargs)
        rest = own[nsteps * per_step:]
        if rest:
            out_full[rest[0]:rest[-1] + 1] = compute(images[rest[0]:rest[-1] + 1], **kwargs)
        for w in pending:
            w.wait()
        return out_full

    n = hi - lo
    nchunks = -(-n // k) if n else 0
    chunk_shape = (k,) + tuple(int(v) for v in shape[1:])
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
            res[i] = compute(bufs[i % 3][:cnt], **kwargs).contiguous()
            res.pop(i - 3, None)                                    # its send (step i+1) was waited for in step i+2
    for ws in works.values():
        for w in ws:
            w.wait()
    return None
