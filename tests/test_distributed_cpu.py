"""The N > 1 path on CPU: world_size-2 gloo processes, the sharding / scatter / gather logic with the
oracle standing in for the GPU compute (tests may use the oracle; the product path may not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polyblur_amd.distributed import default_chunk, exchange_plan, exchange_steps, measured_chunk, shard_bounds, shard_sizes


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


@pytest.mark.parametrize("B,world,root,chunk", [(5, 2, 0, 1), (2, 2, 0, None), (7, 3, 2, 1), (1, 2, 1, None), (9, 2, 0, "auto"), (9, 2, 1, 3),
                                                (11, 3, 0, 2)])
def test_scatter_compute_gather(tmp_path, B, world, root, chunk):
    """image by image (chunk 1 = the default, None), the opt-in sqrt rule ("auto"), and chunks that do not divide the shards: uneven shards, B < world,
    root != 0 -- bit-identical to deblurring every image alone"""
    from oracle import polyblur_ref as ref
    from polyblur_amd.synthetic import synthetic_blurry_batch
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path), root, chunk), nprocs=world, join=True)
    got = np.load(tmp_path / "dist_out.npy")
    x = synthetic_blurry_batch(B, 3, 40, 56, seed0=77)[0]
    want = np.concatenate([ref.polyblur_deblurring(x[i:i + 1], n_iter=2, c=0.362, b=0.468, alpha=6, beta=1) for i in range(B)])
    assert got.shape == want.shape and np.array_equal(got, want)       # sharding must not change a single bit


def _failing_worker(rank, world, port, B, tmp, root, chunk, bad_rank, bad_call):
    """compute = a cheap stand-in (x + 1) that raises on `bad_rank` at its `bad_call`-th call: mid-exchange"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polyblur_amd.distributed import deblur_from_root
    calls = [0]

    def compute(x, **kw):
        calls[0] += 1
        if rank == bad_rank and calls[0] == bad_call:
            raise FloatingPointError("injected failure on rank %d, call %d" % (rank, bad_call))
        return x + 1.0

    shape = (B, 3, 8, 8)
    x = torch.arange(B * 3 * 64, dtype=torch.float32).reshape(shape) if rank == root else None
    if bad_call == 0 and rank == root and bad_rank == root:
        x = x[:, :, :4]                                                # the root's argument check fails: nobody may hang either
    what = "ok"
    try:
        out = deblur_from_root(x, shape, torch.float32, compute=compute, root=root, chunk=chunk)
        if rank == root:
            assert torch.equal(out, x + 1.0)
    except FloatingPointError as e:
        what = "own:" + str(e)
    except ValueError as e:
        what = "own:" + str(e)
    except RuntimeError as e:
        what = "other:" + str(e)
    with open(os.path.join(tmp, "rank%d.txt" % rank), "w") as f:
        f.write(what)
    dist.barrier()                                                      # the group still works after a drained failure
    dist.destroy_process_group()


@pytest.mark.parametrize("B,world,root,chunk,bad_rank,bad_call", [
    (9, 2, 0, 1, 1, 2),        # a peer fails on its second image, image by image
    (9, 2, 0, 2, 0, 2),        # the root fails in its own shard, chunks of two
    (11, 3, 1, 2, 2, 1),       # root != 0, the last rank fails on its first chunk
    (12, 3, 0, "auto", 1, 1),  # the default-chunk rule
    (6, 2, 0, 1, 0, 0),        # the root's argument check fails before the first step
    (6, 2, 0, 1, 5, 1),        # nobody fails: the status word says so
])
def test_failure_on_one_rank_drains_the_exchange(tmp_path, B, world, root, chunk, bad_rank, bad_call):
    """VERDICT r5 #7a: a compute error on one rank mid-exchange -- every rank still posts every remaining step, nobody is left
    waiting for a matching send / recv, all ranks learn of it (the failing one raises its own exception, the others
    RuntimeError naming it), and the process group is usable afterwards.  A hang fails the test by timeout."""
    port = _free_port()
    ctx = mp.spawn(_failing_worker, args=(world, port, B, str(tmp_path), root, chunk, bad_rank, bad_call), nprocs=world, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=1.0):
        if time.time() - t0 > 120:
            for p in ctx.processes:
                p.kill()
            pytest.fail("a rank hangs after a failure on rank %d" % bad_rank)
    got = [open(tmp_path / ("rank%d.txt" % r)).read() for r in range(world)]
    if bad_rank >= world:
        assert got == ["ok"] * world
        return
    for r in range(world):
        if r == bad_rank:
            assert got[r].startswith("own:"), got
        else:
            assert got[r].startswith("other:") and ("rank %d failed" % bad_rank) in got[r], got


def test_chunk_rules():
    """image by image is the default; the opt-in rules: sqrt(shard / 2), and the same minimisation from measured times"""
    assert default_chunk(256, 8, 0) == 4 and default_chunk(8, 8, 0) == 1
    assert measured_chunk(32, 0.34, 0.17) == 4 and measured_chunk(32, 0.17, 0.17) == 1 and measured_chunk(1, 1.0, 0.1) == 1
    assert measured_chunk(32, 1.0, 0.05) == 17 and measured_chunk(4, 10.0, 0.01) == 4 and measured_chunk(32, 0.0, 0.0) == 1


def _chunk_arg_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polyblur_amd.distributed import deblur_from_root
    shape = (5, 1, 4, 4)
    x = torch.ones(shape) if rank == 0 else None
    seen = []

    def compute(t, **kw):
        seen.append(int(t.shape[0]))
        return t * 2

    for bad in (0, -1, True, 1.5, "fast"):
        with pytest.raises(ValueError):
            deblur_from_root(x, shape, torch.float32, compute=compute, chunk=bad)
    out = deblur_from_root(x, shape, torch.float32, compute=compute)               # default: image by image
    if rank == 1:
        assert seen == [1, 1], seen
    seen.clear()
    out = deblur_from_root(x, shape, torch.float32, compute=compute, chunk="measure")
    if rank == 0:
        assert torch.equal(out, x * 2)
    dist.barrier()
    dist.destroy_process_group()


def test_chunk_argument_and_default():
    """chunk <= 0 is rejected (it used to divide by zero), the default is k = 1, 'measure' runs end to end"""
    mp.spawn(_chunk_arg_worker, args=(2, _free_port(), ""), nprocs=2, join=True)


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
