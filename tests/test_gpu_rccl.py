"""The RCCL leg of bench.py on the one GPU of the lease: started the way the driver starts the N-rank runs
(torch.distributed.run, RANK / WORLD_SIZE / MASTER_* in the environment), so `nccl` initialisation with a device id, the
all-reduce, the barrier-bracketed timing and the max-over-ranks reduction all execute on an MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_bench_under_torchrun_one_rank():
    args = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-context", "--no-parity"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ranked = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                             "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + args,
                            capture_output=True, text=True, env=env, timeout=600)
    assert ranked.returncode == 0, ranked.stderr[-2000:]
    a, b = _line(plain.stdout), _line(ranked.stdout)
    assert a["rccl_ranks"] == 1 and a["n_gpus"] == 1
    assert b["rccl_ranks"] == 1 and b["n_gpus"] == 1 and b["scaling"] == "weak"
    # the same workload on the same GPU: the RCCL-bracketed run must tell the same story.  Compared on the median device time per
    # step (one event between steps), which the first-run jitter of a 13-ms timed region does not reach: 15 %
    assert abs(b["value_at_median_step"] / a["value_at_median_step"] - 1.0) < 0.15, (a["value_at_median_step"], b["value_at_median_step"])
    assert abs(b["value"] / a["value"] - 1.0) < 0.3, (a["value"], b["value"])


def test_from_root_one_rank_matches_direct_call(tmp_path):
    """deblur_from_root over the nccl backend with a world of one: the degenerate exchange plan (plain device copies)."""
    code = r'''
import os, sys, json, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from polyblur_amd import polyblur_deblurring
from polyblur_amd.distributed import deblur_from_root
from polyblur_amd.synthetic import synthetic_blurry_batch
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
x = torch.from_numpy(synthetic_blurry_batch(3, 3, 120, 160, seed0=5)[0]).to(dev)
kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
a = deblur_from_root(x, tuple(x.shape), torch.float32, device=dev, **kw)
b = polyblur_deblurring(x, **kw)
print(json.dumps({"equal": bool(torch.equal(a, b)), "world": dist.get_world_size()}))
dist.barrier(); dist.destroy_process_group()
''' % REPO
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "from_root_one_rank.py"
    script.write_text(code)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _line(r.stdout)
    assert res == {"equal": True, "world": 1}


def test_c_abi_communicator_world_of_one():
    """pb_comm_* with one rank: no RCCL traffic, the scatter and the gather are device copies of the whole batch; the
    unique id comes from the RCCL library the engine loads on demand."""
    import ctypes as C
    import numpy as np
    import torch
    from polyblur_amd import _capi as capi
    from polyblur_amd.engine import get_engine
    eng = get_engine(0)
    lib = eng.lib
    ident = C.create_string_buffer(128)
    assert lib.pb_comm_unique_id(ident) == 0 and any(ident.raw)
    comm = C.c_void_p()
    eng._check(lib.pb_comm_init(C.byref(comm), eng.ctx, 0, 1, None))
    B, Cc, H, W = 3, 3, 40, 56
    x = torch.rand(B, Cc, H, W, device="cuda")
    shard = torch.zeros_like(x)
    back = torch.zeros_like(x)
    eng._check(lib.pb_comm_scatter(comm, C.c_void_p(x.data_ptr()), C.c_void_p(shard.data_ptr()), capi.PB_F32, B, Cc, H, W, 0))
    eng._check(lib.pb_comm_gather(comm, C.c_void_p(shard.data_ptr()), C.c_void_p(back.data_ptr()), capi.PB_F32, B, Cc, H, W, 0))
    eng.synchronize()
    assert torch.equal(shard, x) and torch.equal(back, x)
    assert lib.pb_comm_destroy(comm) == 0


def test_c_abi_deblur_from_root_world_of_one():
    """pb_comm_deblur_from_root with one rank is the plain batch call (no exchange); with more ranks it walks pb_comm_plan,
    whose order is checked against the Python layer's on the CPU (tests/test_capi_cpu.py) -- the lease has one GPU, so no
    grouped ncclSend / ncclRecv of it has run here."""
    import ctypes as C
    import numpy as np
    import torch
    from polyblur_amd import _capi as capi, polyblur_deblurring
    from polyblur_amd.engine import get_engine
    from polyblur_amd.synthetic import synthetic_blurry_batch
    eng = get_engine(0)
    lib = eng.lib
    comm = C.c_void_p()
    eng._check(lib.pb_comm_init(C.byref(comm), eng.ctx, 0, 1, None))
    x = torch.from_numpy(synthetic_blurry_batch(3, 3, 120, 168, seed0=9)[0]).cuda()
    out = torch.zeros_like(x)
    kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
    o = eng.make_options(**kw)
    eng.set_stream(torch.cuda.current_stream(0).cuda_stream)
    eng._check(lib.pb_comm_deblur_from_root(comm, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), capi.PB_F32, 3, 3, 120, 168, C.byref(o), 0))
    eng.synchronize()
    assert torch.equal(out, polyblur_deblurring(x, **kw))
    assert lib.pb_comm_deblur_from_root(comm, None, C.c_void_p(out.data_ptr()), capi.PB_F32, 3, 3, 120, 168, C.byref(o), 0) != 0   # the root passes the batch
    assert lib.pb_comm_deblur_from_root(comm, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 7, 3, 3, 120, 168, C.byref(o), 0) != 0  # unknown dtype
    assert lib.pb_comm_destroy(comm) == 0


# ---- more than one GPU: the first contact of the world > 1 branches with hardware ---------------------------------------
# (the lease of rounds 1 - 5 shows ONE MI355X, so these are skipped there; the driver's multi-GPU tier runs them)
_TWO_RANK = r'''
import ctypes as C, json, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
from polyblur_amd import polyblur_deblurring, _capi as capi
from polyblur_amd.distributed import deblur_from_root, default_chunk
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank); dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
res = {}
eng = get_engine(rank)
eng.set_stream(torch.cuda.current_stream(rank).cuda_stream)
# the C ABI communicator: rank 0 makes the id, everybody gets it through torch.distributed (any channel would do)
ident = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    buf = C.create_string_buffer(128)
    assert eng.lib.pb_comm_unique_id(buf) == 0
    ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
ident = ident.to(dev); dist.broadcast(ident, 0); ident_b = bytes(ident.cpu().numpy().tobytes())
comm = C.c_void_p()
eng._check(eng.lib.pb_comm_init(C.byref(comm), eng.ctx, rank, world, ident_b))
for B, root, chunk in %(cases)r:
    H, W = 120, 168
    x_np = synthetic_blurry_batch(B, 3, H, W, seed0=40 + B)[0]
    x = torch.from_numpy(x_np).to(dev) if rank == root else None
    # (1) the Python layer over torch.distributed
    got = deblur_from_root(x, (B, 3, H, W), torch.float32, device=dev, root=root, chunk=chunk, **kw)
    # (2) the C ABI over its own communicator
    out = torch.zeros((B, 3, H, W), device=dev) if rank == root else None
    o = eng.make_options(**kw)
    eng._check(eng.lib.pb_comm_set_chunk(comm, 0 if chunk is None else (capi.PB_COMM_CHUNK_AUTO if chunk == "auto" else chunk)))
    eng._check(eng.lib.pb_comm_deblur_from_root(comm, C.c_void_p(x.data_ptr()) if rank == root else None,
                                                C.c_void_p(out.data_ptr()) if rank == root else None, capi.PB_F32, B, 3, H, W, C.byref(o), root))
    eng.synchronize()
    if rank == root:
        # every image what it gets in a call of its own (bit-identical: what an image gets does not depend on its batch)
        want = torch.cat([polyblur_deblurring(x[i:i + 1], **kw) for i in range(B)])
        res["%%d/%%d/%%s" %% (B, root, chunk)] = [bool(torch.equal(got, want)), bool(torch.equal(out, want))]
    dist.barrier()
eng._check(eng.lib.pb_comm_destroy(comm))
gathered = [None] * world
dist.all_gather_object(gathered, res)
if rank == 0:
    merged = {}
    for g in gathered:
        merged.update(g)
    print(json.dumps({"world": world, "results": merged}))
dist.barrier(); dist.destroy_process_group()
'''


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the lease shows one (world > 1 stays unmeasured until a node shows more)")
def test_two_ranks_from_root_python_and_c_abi(tmp_path):
    """uneven shards (5 over 2), B < world (1 over 2), root != 0, image by image and in chunks: deblur_from_root and
    pb_comm_deblur_from_root under torch.distributed.run with 2 ranks over RCCL, bit-equal to the unsharded calls"""
    cases = [(5, 0, 1), (5, 1, 2), (1, 0, None), (1, 1, 1), (9, 0, None), (8, 1, 3), (17, 0, "auto")]
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANK % dict(repo=REPO, cases=cases))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = _line(r.stdout)
    assert res["world"] == 2 and len(res["results"]) == len(cases)
    assert all(all(v) for v in res["results"].values()), res


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs")
def test_bench_two_ranks_resident_and_from_root():
    """bench.py the way the driver starts it at N = 2: the resident line, and the from-root mode (chunked exchange)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in ([], ["--mode", "from_root", "--config", "cfg4", "--batch", "4"]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                            "--no-cpu-baseline", "--no-context", "--no-parity"] + extra, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = _line(r.stdout)
        assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0
