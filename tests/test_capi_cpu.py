"""CPU-side checks of the boundary: the shared library builds/loads, exports every symbol the
header declares, and the host logic (argument validation, layout, error behaviour) works
without a GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from polyblur_amd import _capi as capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.library_path()):
        from polyblur_amd.build import build
        build(verbose=False)
    return capi.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(REPO, "include", "polyblur_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pb_version() == 200


def test_struct_layout_matches_header(lib):
    o = capi.pb_options()
    lib.pb_default_options(ctypes.byref(o))
    # the functional API's defaults, reference deblurring.py:23-25
    assert (o.n_iter, o.n_angles, o.n_interpolated_angles) == (1, 6, 30)
    assert abs(o.c - 0.352) < 1e-7 and abs(o.b - 0.768) < 1e-7 and o.alpha == 2 and o.beta == 3
    assert abs(o.sigma_r - 0.8) < 1e-7 and o.sigma_s == 2.0 and o.q == 0 and o.force_theta_deg == -1.0
    assert (o.remove_halo, o.edgetaping, o.prefilter, o.discard_saturation, o.boundary, o.support) == (0,) * 6
    assert o.ker_size == 25
    assert ctypes.sizeof(capi.pb_blur_info) == capi.INFO_DTYPE.itemsize == 4 * (2 + 13 + 64 + 4 + 2 + 625 + 100 + 832 + 832 + 2 + 100 + 3 + 184)


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from polyblur_amd import polyblur_deblurring
    with pytest.raises(capi.PolyblurHipError):
        polyblur_deblurring(np.zeros((16, 16, 3), np.float32))


def test_argument_validation_before_any_device_work():
    from polyblur_amd import polyblur_deblurring, PolyblurDeblurring
    x = np.zeros((16, 16, 3), np.float32)
    with pytest.raises(ValueError):
        polyblur_deblurring(x, method="nope")
    with pytest.raises(ValueError):
        polyblur_deblurring(x, q=0.5)
    with pytest.raises(NotImplementedError):
        polyblur_deblurring(x, ker_size=51)
    with pytest.raises(NotImplementedError):
        polyblur_deblurring(x, ker_size=31, method="direct_separable")   # (above 25 with edgetaping is built since round 6)
    with pytest.raises(NotImplementedError):
        polyblur_deblurring(x, ker_size=12, method="direct_separable")     # (an even size with edgetaping is built since round 6)
    with pytest.raises(NotImplementedError):
        polyblur_deblurring(x, method="direct_separable", edgetaping=True)
    with pytest.raises(ValueError):
        polyblur_deblurring(np.zeros((4,), np.float32))
    with pytest.raises(TypeError):
        polyblur_deblurring([[0.0]])
    with pytest.raises(ValueError):
        PolyblurDeblurring(patch_decomposition=True)(x)             # patch branch: (B,C,H,W) tensors only
    # the module's defaults differ from the functional ones (reference deblurring.py:266-268)
    import inspect
    f = inspect.signature(polyblur_deblurring).parameters
    m = inspect.signature(PolyblurDeblurring.forward).parameters
    assert (f["b"].default, f["beta"].default, f["sigma_r"].default) == (0.768, 3, 0.8)
    assert (m["b"].default, m["beta"].default, m["sigma_r"].default) == (0.468, 4, 0.4)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under polyblur_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "polyblur_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("on the oracle", ""), f


def test_synthetic_generator_is_deterministic():
    from polyblur_amd.synthetic import synthetic_blurry_batch
    a, pa = synthetic_blurry_batch(2, 3, 40, 56, seed0=5)
    b, pb = synthetic_blurry_batch(2, 3, 40, 56, seed0=5)
    assert np.array_equal(a, b) and pa == pb and a.dtype == np.float32
    assert a.min() >= 0 and a.max() <= 1 and not np.array_equal(a[0], a[1])
    c, pc = synthetic_blurry_batch(1, 3, 40, 56, seed0=5, force_theta_deg=0.0)
    assert pc[0][2] == 0.0


def test_patch_lattice_matches_reference_formulas():
    """deblurring.py:281-299 for the reference's default 400 / 0.25 and a small case"""
    from polyblur_amd.deblurring import patch_grid, kaiser_window_periodic
    g = patch_grid(1080, 1920, (400, 400), 0.25)
    assert (g["step_h"], g["step_w"]) == (300, 300)
    assert (g["new_h"], g["new_w"]) == (1300, 2200) and (g["n_i"], g["n_j"]) == (4, 7)
    assert (g["pad_top"], g["pad_left"]) == (110, 140)
    g = patch_grid(90, 60, (100, 100), 0.25)                # image smaller than one patch
    assert (g["new_h"], g["new_w"], g["n_i"], g["n_j"]) == (100, 100, 1, 1)
    import torch
    for n in (400, 7):
        assert np.abs(kaiser_window_periodic(n) - torch.kaiser_window(n, periodic=True, beta=5.0).numpy()).max() < 1e-6


def test_patch_lattice_without_a_patch_is_refused():
    """an image shorter than patch_size - step along an axis: deblurring.py:284-295 gives new_h < patch_size and no patch
    corner at all (the reference's branch would return zeros); the host says so before anything reaches the GPU"""
    import torch
    from polyblur_amd import PolyblurDeblurring
    from polyblur_amd.deblurring import patch_grid
    assert patch_grid(100, 298, (400, 400), 0.4)["n_i"] == 0
    with pytest.raises(ValueError, match="holds no patch"):
        PolyblurDeblurring(patch_decomposition=True, patch_size=400, patch_overlap=0.4)(torch.zeros(1, 3, 100, 299))
    with pytest.raises(ValueError, match="positive step"):
        PolyblurDeblurring(patch_decomposition=True, patch_size=1, patch_overlap=0.5)(torch.zeros(1, 3, 64, 64))


def test_comm_shards_match_the_python_layer(lib):
    """pb_comm_shard (csrc/comm.hip) and polyblur_amd.distributed.shard_bounds cut a batch the same way."""
    from polyblur_amd.distributed import shard_bounds
    for B in (0, 1, 7, 8, 9, 64, 256):
        for world in (1, 2, 3, 8):
            covered = 0
            for rank in range(world):
                first, count = ctypes.c_int(), ctypes.c_int()
                assert lib.pb_comm_shard(B, world, rank, ctypes.byref(first), ctypes.byref(count)) == 0
                lo, hi = shard_bounds(B, world, rank)
                assert (first.value, first.value + count.value) == (lo, hi)
                covered += count.value
            assert covered == B
    first, count = ctypes.c_int(), ctypes.c_int()
    assert lib.pb_comm_shard(4, 2, 2, ctypes.byref(first), ctypes.byref(count)) != 0      # rank out of range


def test_comm_exchange_plan_matches_the_python_layer(lib):
    """pb_comm_plan / pb_comm_plan_steps (csrc/comm.hip: what pb_comm_deblur_from_root walks) and
    polyblur_amd.distributed.exchange_plan / exchange_steps enumerate the same operations in the same order for every
    (B, world, root, rank, step); and the root's list of a step, filtered to one peer, is that peer's list with send and
    recv swapped (the pairing RCCL needs: it has no tags)."""
    from polyblur_amd.distributed import exchange_plan, exchange_steps
    ops = (ctypes.c_int * (3 * 2 * 8))()
    n = ctypes.c_int()
    for B in (1, 2, 5, 8, 9, 33):
        for world in (1, 2, 3, 8):
            for root in sorted({0, world - 1, world // 2}):
                steps = lib.pb_comm_plan_steps(B, world, root)
                assert steps == exchange_steps(B, world, root)
                for step in range(steps + 1):
                    lists = {}
                    for rank in range(world):
                        assert lib.pb_comm_plan(B, world, root, rank, step, ops, ctypes.byref(n)) == 0
                        got = [("send" if ops[3 * i] else "recv", ops[3 * i + 1], ops[3 * i + 2]) for i in range(n.value)]
                        assert got == exchange_plan(B, world, root, rank, step), (B, world, root, rank, step)
                        lists[rank] = got
                    for peer in range(world):
                        if peer != root:
                            mirror = [("recv" if k == "send" else "send", root, img) for k, p, img in lists[root] if p == peer]
                            assert mirror == lists[peer]
    assert lib.pb_comm_plan_steps(4, 2, 2) < 0 and lib.pb_comm_plan(4, 2, 0, 2, 0, ops, ctypes.byref(n)) != 0


def test_comm_chunked_plan_matches_the_python_layer(lib):
    """the same for exchange steps of k images (pb_comm_plan_chunked / pb_comm_plan_steps_chunked / pb_comm_default_chunk
    against exchange_plan(..., chunk) / exchange_steps(..., chunk) / default_chunk): every (B, world, root, rank, step, k);
    every image of a peer's shard travels out exactly once and comes back exactly once, two steps later"""
    from polyblur_amd.distributed import default_chunk, exchange_plan, exchange_steps, shard_bounds
    ops = (ctypes.c_int * (4 * 2 * 8))()
    n = ctypes.c_int()
    for B in (1, 2, 5, 9, 33, 64, 256):
        for world in (1, 2, 3, 8):
            for root in sorted({0, world - 1}):
                assert lib.pb_comm_default_chunk(B, world, root) == default_chunk(B, world, root)
                for k in sorted({1, 2, 3, 7, default_chunk(B, world, root)}):
                    steps = lib.pb_comm_plan_steps_chunked(B, world, root, k)
                    assert steps == exchange_steps(B, world, root, k)
                    out, back = [], []
                    for step in range(steps + 1):
                        lists = {}
                        for rank in range(world):
                            assert lib.pb_comm_plan_chunked(B, world, root, rank, step, k, ops, ctypes.byref(n)) == 0
                            got = [("send" if ops[4 * i] else "recv", ops[4 * i + 1], ops[4 * i + 2], ops[4 * i + 3]) for i in range(n.value)]
                            want = [(o[0], o[1], o[2], o[3] if len(o) > 3 else 1) for o in exchange_plan(B, world, root, rank, step, k)]
                            assert got == want, (B, world, root, rank, step, k)
                            lists[rank] = got
                        for peer in range(world):
                            if peer != root:
                                mirror = [("recv" if kind == "send" else "send", root, a, c) for kind, p, a, c in lists[root] if p == peer]
                                assert mirror == lists[peer]
                        out += [i for kind, p, a, c in lists[root] if kind == "send" for i in range(a, a + c)]
                        back += [i for kind, p, a, c in lists[root] if kind == "recv" for i in range(a, a + c)]
                    peers = sorted(i for r in range(world) if r != root for i in range(*shard_bounds(B, world, r)))
                    assert sorted(out) == peers and sorted(back) == peers
    assert lib.pb_comm_default_chunk(256, 8, 0) == 4 and lib.pb_comm_default_chunk(8, 8, 0) == 1      # BASELINE configs 4 and 5
    assert lib.pb_comm_plan_steps_chunked(4, 2, 0, 0) < 0


def test_line_length_tiers(lib):
    """include/polyblur_hip.h: 1 = whole lines in LDS, 2 = through a line buffer in device memory, 0 = not taken"""
    want = {1: 0, 2: 1, 4096: 1, 8191: 1, 8192: 1, 8200: 2, 20480: 1, 20736: 2, 40001: 2, 65536: 2, 65537: 0, 9000: 1, 12000: 1, 12001: 2}
    for n, tier in want.items():
        assert lib.pb_fft_length_supported(n) == tier, n
