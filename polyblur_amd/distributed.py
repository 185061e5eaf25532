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

``deblur_from_root`` moves one image per peer per step, all peers at once: step t is ONE grouped
point-to-point exchange (``batch_isend_irecv`` = ncclGroupStart/End on RCCL) in which the root sends
image t of every peer's shard and receives result t-2 from every peer, so the root's seven xGMI
links carry traffic concurrently and each peer has image t arriving and result t-2 leaving while
it deblurs image t-1 on its compute stream.  Nothing relies on message tags (RCCL ignores them):
both sides enumerate the exchanges of a step in the same order.  Results land directly in the
output batch (no staging buffers on the root).

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


def exchange_plan(batch: int, world: int, root: int, rank: int, step: int):
    """The point-to-point operations of `rank` in exchange step `step`, as (kind, peer, image index) in the
    order both sides enumerate them: the root's list for a step, filtered to one peer, is that peer's list
    with send and recv swapped.  Images travel root -> peer in step t, results come back in step t + 2."""
    ops = []
    peers = [r for r in range(world) if r != root] if rank == root else [rank]
    for r in peers:
        lo, hi = shard_bounds(batch, world, r)
        if lo + step < hi:
            ops.append(("send" if rank == root else "recv", r if rank == root else root, lo + step))
        if step >= 2 and lo + step - 2 < hi:
            ops.append(("recv" if rank == root else "send", r if rank == root else root, lo + step - 2))
    return ops


def exchange_steps(batch: int, world: int, root: int) -> int:
    """Number of exchange steps: the largest peer shard plus the two-step return lag (0 without peers)."""
    sizes = [n for r, n in enumerate(shard_sizes(batch, world)) if r != root]
    return (max(sizes) + 2) if sizes and max(sizes) > 0 else 0


def deblur_from_root(images, shape: Sequence[int], dtype, compute: Optional[Callable] = None, device=None,
                     root: int = 0, group=None, **kwargs):
    """Scatter a (B,C,H,W) batch that lives on `root`, deblur every shard where it lands, gather
    the result on `root` (other ranks return None).  `images` is ignored on non-root ranks;
    `shape` / `dtype` must be given on all ranks."""
    import torch
    import torch.distributed as dist
    if compute is None:
        from .deblurring import polyblur_deblurring as compute
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = int(shape[0])
    lo, hi = shard_bounds(B, world, rank)
    img_shape = (1,) + tuple(int(v) for v in shape[1:])
    if device is None:
        device = images.device if (rank == root and images is not None) else torch.device("cpu")
    nsteps = exchange_steps(B, world, root)

    def post(ops, tensor_of):
        if not ops:
            return []
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, tensor_of(kind, k), peer, group) for kind, peer, k in ops]
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
            pending += post(exchange_plan(B, world, root, rank, t),
                            lambda kind, k: images[k:k + 1] if kind == "send" else out_full[k:k + 1])
            for k in own[t * per_step:(t + 1) * per_step]:         # the root's own shard, spread over the steps
                out_full[k:k + 1] = compute(images[k:k + 1], **kwargs)
        for k in own[nsteps * per_step:]:
            out_full[k:k + 1] = compute(images[k:k + 1], **kwargs)
        for w in pending:
            w.wait()
        return out_full

    n = hi - lo
    bufs = [torch.empty(img_shape, dtype=dtype, device=device) for _ in range(min(n, 3))]   # ring: arriving / in work / spare
    res = {}
    works = {}
    for t in range(nsteps):
        ops = exchange_plan(B, world, root, rank, t)
        works[t] = post(ops, lambda kind, k: bufs[(k - lo) % 3] if kind == "recv" else res[k])
        i = t - 1                                                   # image i arrived in step t-1: deblur it now
        if 0 <= i < n:
            for w in works.pop(t - 1):
                w.wait()
            res[lo + i] = compute(bufs[i % 3], **kwargs).contiguous()
            res.pop(lo + i - 3, None)                               # its send (step i+1) was waited for in step i+2
    for ws in works.values():
        for w in ws:
            w.wait()
    return None
