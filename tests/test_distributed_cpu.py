"""The N > 1 path on CPU: world_size-2 gloo processes, the sharding / scatter / gather logic with the
oracle standing in for the GPU compute (tests may use the oracle; the product path may not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polyblur_amd.distributed import shard_bounds, shard_sizes


def test_shard_bounds():
    for B in (0, 1, 5, 8, 9, 256):
        for n in (1, 2, 3, 8):
            sizes = shard_sizes(B, n)
            assert sum(sizes) == B and max(sizes) - min(sizes) <= 1
            assert sizes == sorted(sizes, reverse=True)                 # the first B % n ranks get the extra image
            pos = 0
            for r in range(n):
                a, b = shard_bounds(B, n, r)
                assert a == pos and b - a == sizes[r]
                pos = b
    assert shard_sizes(256, 8) == [32] * 8 and shard_sizes(8, 8) == [1] * 8   # BASELINE configs 4 and 5
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(x, **kw):
    from oracle import polyblur_ref as ref
    return torch.from_numpy(ref.polyblur_deblurring(x.numpy(), **kw))


def _worker(rank, world, port, B, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polyblur_amd.distributed import deblur_from_root
    from polyblur_amd.synthetic import synthetic_blurry_batch
    kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
    shape = (B, 3, 40, 56)
    x = torch.from_numpy(synthetic_blurry_batch(B, 3, 40, 56, seed0=77)[0]) if rank == 0 else None
    out = deblur_from_root(x, shape, torch.float32, compute=_oracle_compute, **kw)
    if rank == 0:
        np.save(os.path.join(tmp, "dist_out.npy"), out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2])
def test_scatter_compute_gather_two_ranks(tmp_path, B):
    from oracle import polyblur_ref as ref
    from polyblur_amd.synthetic import synthetic_blurry_batch
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "dist_out.npy")
    x = synthetic_blurry_batch(B, 3, 40, 56, seed0=77)[0]
    want = np.concatenate([ref.polyblur_deblurring(x[i:i + 1], n_iter=2, c=0.362, b=0.468, alpha=6, beta=1) for i in range(B)])
    assert got.shape == want.shape and np.array_equal(got, want)       # sharding must not change a single bit
