"""The N > 1 path on CPU: world_size-2 gloo processes, the sharding / scatter / gather logic with the
oracle standing in for the GPU compute (tests may use the oracle; the product path may not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polyblur_amd.distributed import exchange_plan, exchange_steps, shard_bounds, shard_sizes


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


def test_exchange_plan_matches_on_both_sides():
    """tag-free ordering: per step, the root's operations towards a peer are that peer's operations mirrored, in
    the same order; every image travels out once and its result comes back once, two steps later"""
    for B in (0, 1, 2, 5, 8, 9, 33, 256):
        for world in (1, 2, 3, 8):
            for root in {0, world - 1}:
                steps = exchange_steps(B, world, root)
                sent, back = [], []
                for t in range(steps):
                    rops = exchange_plan(B, world, root, root, t)
                    for r in range(world):
                        if r == root:
                            continue
                        mine = exchange_plan(B, world, root, r, t)
                        mirrored = [("recv" if k == "send" else "send", root, i) for k, p, i in rops if p == r]
                        assert mine == mirrored
                    sent += [i for k, p, i in rops if k == "send"]
                    back += [(i, t) for k, p, i in rops if k == "recv"]
                peers_imgs = sorted(i for r in range(world) if r != root for i in range(*shard_bounds(B, world, r)))
                assert sorted(sent) == peers_imgs and sorted(i for i, _ in back) == peers_imgs
                assert not exchange_plan(B, world, root, root, steps)          # nothing left after the last step


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(x, **kw):
    """image by image: NumPy's batched transforms are not bit-identical to its single ones (6 of 11 images here differ by
    5e-7 between a batch call and a single call of the oracle), and this test is about where the images travel"""
    from oracle import polyblur_ref as ref
    return torch.from_numpy(np.concatenate([ref.polyblur_deblurring(x[i:i + 1].numpy(), **kw) for i in range(x.shape[0])]))


def _worker(rank, world, port, B, tmp, root=0, chunk=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polyblur_amd.distributed import deblur_from_root
    from polyblur_amd.synthetic import synthetic_blurry_batch
    kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
    shape = (B, 3, 40, 56)
    x = torch.from_numpy(synthetic_blurry_batch(B, 3, 40, 56, seed0=77)[0]) if rank == root else None
    out = deblur_from_root(x, shape, torch.float32, compute=_oracle_compute, root=root, chunk=chunk, **kw)
    if rank == root:
        np.save(os.path.join(tmp, "dist_out.npy"), out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,world,root,chunk", [(5, 2, 0, 1), (2, 2, 0, None), (7, 3, 2, 1), (1, 2, 1, None), (9, 2, 0, None), (9, 2, 1, 3),
                                                (11, 3, 0, 2)])
def test_scatter_compute_gather(tmp_path, B, world, root, chunk):
    """image by image (chunk 1), the default chunk, and chunks that do not divide the shards: uneven shards, B < world,
    root != 0 -- bit-identical to deblurring every image alone"""
    from oracle import polyblur_ref as ref
    from polyblur_amd.synthetic import synthetic_blurry_batch
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path), root, chunk), nprocs=world, join=True)
    got = np.load(tmp_path / "dist_out.npy")
    x = synthetic_blurry_batch(B, 3, 40, 56, seed0=77)[0]
    want = np.concatenate([ref.polyblur_deblurring(x[i:i + 1], n_iter=2, c=0.362, b=0.468, alpha=6, beta=1) for i in range(B)])
    assert got.shape == want.shape and np.array_equal(got, want)       # sharding must not change a single bit


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` started by hand must become two ranks (one per GPU on a real node): the launcher leg
    is exercised on CPU over gloo -- rendezvous on 127.0.0.1, an all-reduce of ones, rank 0 prints the line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["gpus_requested"] == 2
