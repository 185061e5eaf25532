"""Batch sharding across the GPUs of one node.

Images are independent (every reduction in the pipeline is per image), so the batch dimension is
split into contiguous shards, one per rank (= one process per GPU): the first ``B mod n`` ranks get
one extra image.  There is no data-path collective inside the hot path; RCCL over xGMI (the
``nccl`` backend of torch.distributed on ROCm) is used only to move shards from / to the root
rank.  Transfers are per image and non-blocking, so that image k+1 travels while image k is
deblurred and image k-1 returns.

Two usage patterns:
  * shards already resident on every GPU (what bench.py times, "weak" scaling):
        out_local = polyblur_deblurring(x_local, ...)
  * a whole batch on the root rank:
        out = deblur_from_root(x_or_None, shape, dtype, ...)      # returns the full batch on root

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
    out_full = None
    if rank == root:
        if images is None or tuple(images.shape) != tuple(shape):
            raise ValueError("root must pass the full batch with the announced shape")
        out_full = torch.empty_like(images)
        sends = []
        for r in range(world):
            if r == root:
                continue
            a, b = shard_bounds(B, world, r)
            for k in range(a, b):                                   # one message per image: pipelined
                sends.append(dist.isend(images[k:k + 1].contiguous(), dst=r, group=group, tag=k))
        recvs = []
        for r in range(world):
            if r == root:
                continue
            a, b = shard_bounds(B, world, r)
            for k in range(a, b):
                buf = torch.empty(img_shape, dtype=dtype, device=device)
                recvs.append((k, buf, dist.irecv(buf, src=r, group=group, tag=B + k)))
        for k in range(lo, hi):                                     # root's own shard, overlapped with the traffic
            out_full[k:k + 1] = compute(images[k:k + 1].contiguous(), **kwargs)
        for w in sends:
            w.wait()
        for k, buf, w in recvs:
            w.wait()
            out_full[k:k + 1] = buf
        return out_full
    bufs = [torch.empty(img_shape, dtype=dtype, device=device) for _ in range(lo, hi)]
    works = [dist.irecv(bufs[i], src=root, group=group, tag=lo + i) for i in range(hi - lo)]
    back = []
    for i, w in enumerate(works):
        w.wait()                                                    # image i is here; i+1.. still in flight
        res = compute(bufs[i], **kwargs).contiguous()
        back.append((res, dist.isend(res, dst=root, group=group, tag=B + lo + i)))
    for _, w in back:
        w.wait()
    return None
