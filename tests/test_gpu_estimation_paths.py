"""The estimation's two launch sequences and two plan families give the same records.

q == 0, fp32 planes, lines of up to 4096 samples: gray + range partials + row transform run as ONE launch
(csrc/estimate.hip: gray_rows_kernel) instead of gray_minmax_kernel + grad_rows_kernel -- the same sums in the same order,
so every field of the record must be bit-identical (PB_EST_GRAY_ROWS=0 / 1 / 2 select never / default / any line length).
Lines of 4320 / 7680 / 3240 ... samples take radices 18 / 20 / 24 (PB_FFT_EXT_RADIX=0: the greedy plan): a different
factorisation rounds differently, so those are compared within the tolerance of the reference goldens
(tests/test_gpu_parity.py::test_estimate_blur) and against the oracle (blur_estimation.py:18-79)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd.synthetic import synthetic_blurry_batch


def _engine(**env):
    from polyblur_amd.engine import Engine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def engines():
    return {"default": _engine(), "two_launches": _engine(PB_EST_GRAY_ROWS=0), "fused_any": _engine(PB_EST_GRAY_ROWS=2),
            "greedy": _engine(PB_FFT_EXT_RADIX=0)}


def opts(**kw):
    from polyblur_amd.engine import Engine
    return Engine.make_options(**kw)


FIELDS = ("mags", "interp", "theta", "sigma", "rho", "kernel", "gray_min", "gray_max")


@pytest.mark.parametrize("shape", [(1, 3, 211, 157), (2, 1, 64, 96), (3, 3, 120, 4096), (1, 4, 75, 33), (1, 3, 36, 7680),
                                   (2, 3, 1080, 1920)])
def test_fused_gray_rows_is_bit_identical(engines, shape):
    """odd heights (an unpaired last row), one / three / four channels, batches, the longest fused line (4096), and a line of
    7680 samples that only PB_EST_GRAY_ROWS=2 fuses"""
    B, C, H, W = shape
    img, _ = synthetic_blurry_batch(B, C, H, W, seed0=31)
    o = opts(c=0.362, b=0.468)
    a = engines["two_launches"].estimate_blur(img, o)
    for name in ("default", "fused_any"):
        b = engines[name].estimate_blur(img, o)
        for f in FIELDS:
            assert np.array_equal(np.asarray(a[f]), np.asarray(b[f])), (name, f)


@pytest.mark.parametrize("shape", [(1, 3, 4320, 64), (1, 1, 48, 7680), (1, 3, 3240, 40)])
def test_extended_radix_plans_against_greedy_and_oracle(engines, shape):
    B, C, H, W = shape
    img, _ = synthetic_blurry_batch(B, C, H, W, seed0=37)
    o = opts(c=0.362, b=0.468)
    a = engines["greedy"].estimate_blur(img, o)
    b = engines["default"].estimate_blur(img, o)
    assert np.max(np.abs(np.asarray(a["mags"]) - np.asarray(b["mags"]))) < 5e-6
    assert np.array_equal(np.asarray(a["theta"]), np.asarray(b["theta"]))
    assert np.max(np.abs(np.asarray(a["sigma"]) - np.asarray(b["sigma"]))) < 2e-5
    assert np.max(np.abs(np.asarray(a["rho"]) - np.asarray(b["rho"]))) < 2e-5
    # the oracle's estimate of the same image
    _, r = ref.estimate_gaussian_blur(img, c=0.362, b=0.468, return_info=True)
    assert np.max(np.abs(np.asarray(b["mags"])[:, :7] - r["mags"])) < 5e-6
    assert np.max(np.abs(np.asarray(b["sigma"]) - r["sigma"])) < 2e-5 and np.max(np.abs(np.asarray(b["rho"]) - r["rho"])) < 2e-5


@pytest.fixture(scope="module")
def runtime_plan_engine():
    return _engine(PB_COLS_FIXED=0)


@pytest.mark.parametrize("shape,sat", [((1, 3, 2160, 3840), False), ((1, 3, 2160, 3840), True), ((4, 3, 1080, 1920), False),
                                       ((5, 1, 1080, 1920), True), ((1, 3, 4320, 7680), False), ((1, 1, 4320, 512), True),
                                       ((1, 3, 2160, 2000), False), ((1, 3, 2160, 3848), False)])
def test_fixed_plan_columns_are_bit_identical(engines, runtime_plan_engine, shape, sat):
    """csrc/lines_fixed.hip (the column transform of 2160- / 1080- / 4320-point lines as a kernel that holds one plan, its
    tile fetched by LDS-DMA, its twiddle table in LDS) against grad_cols_kernel with the run-time plan (PB_COLS_FIXED=0):
    the same butterflies in the same order -- every field of the record bit-identical, with and without the saturation
    mask; and both within the goldens' tolerance of the oracle (blur_estimation.py:112-134, filters.py:159-186)."""
    B, C, H, W = shape
    nd = min(B, 2)
    img, _ = synthetic_blurry_batch(nd, C, H, W, seed0=41)
    img = np.concatenate([img] * ((B + nd - 1) // nd))[:B]
    if sat:
        img = np.clip(img * 1.35, 0.0, 1.0).astype(np.float32)              # a good part of the image under the mask
    o = opts(c=0.362, b=0.468, discard_saturation=sat)
    a = runtime_plan_engine.estimate_blur(img, o)
    b = engines["default"].estimate_blur(img, o)
    for f in FIELDS:
        assert np.array_equal(np.asarray(a[f]), np.asarray(b[f])), f
    if H * W <= 2160 * 3840:
        _, r = ref.estimate_gaussian_blur(img[:1], c=0.362, b=0.468, discard_saturation=sat, return_info=True)
        assert np.max(np.abs(np.asarray(b["mags"])[:1, :7] - r["mags"])) < 5e-6


@pytest.fixture(scope="module")
def runtime_plan_rows_engine():
    return _engine(PB_ROWS_FIXED=0)


@pytest.mark.parametrize("shape", [(1, 3, 2160, 3840), (1, 1, 33, 3840), (20, 3, 200, 3840), (1, 3, 1080, 1920), (4, 3, 1080, 1920),
                                   (3, 1, 75, 1920), (1, 3, 36, 7680)])
def test_fixed_plan_rows_are_bit_identical(engines, runtime_plan_rows_engine, shape):
    """csrc/lines_fixed.hip's row transforms (3840- / 1920- / 7680-point lines: one plan per kernel, the line padded in LDS)
    against gray_rows_kernel / grad_rows_kernel with the run-time plan (PB_ROWS_FIXED=0): one image and batches (256 and 128
    threads per row pair), one and three channels, an odd height (an unpaired last row) -- every field of the record and
    both gradient planes bit-identical (filters.py:159-186)."""
    B, C, H, W = shape
    nd = min(B, 2)
    img, _ = synthetic_blurry_batch(nd, C, H, W, seed0=43)
    img = np.concatenate([img] * ((B + nd - 1) // nd))[:B]
    o = opts(c=0.362, b=0.468)
    a = runtime_plan_rows_engine.estimate_blur(img, o)
    b = engines["default"].estimate_blur(img, o)
    for f in FIELDS:
        assert np.array_equal(np.asarray(a[f]), np.asarray(b[f])), f
    planes = img.reshape(B * C, H, W)[: min(B * C, 4)]
    gxa, gya = runtime_plan_rows_engine.fourier_gradients(planes)
    gxb, gyb = engines["default"].fourier_gradients(planes)
    assert np.array_equal(gxa, gxb) and np.array_equal(gya, gyb)
    if H * W <= 1080 * 1920:
        rx, ry = ref.spectral_gradients(planes[None])
        assert np.max(np.abs(gxb - rx[0])) < 2e-5 and np.max(np.abs(gyb - ry[0])) < 2e-5


def test_half_gradient_planes_of_an_fp16_call(engines, runtime_plan_engine):
    """an fp16 call with remove_halo keeps grad_img's two planes and every iteration's d/dx of the deblurred
    image as fp16 planes where the compiled-plan line transforms can write them (csrc/filters.hip: halo_kernel's TG) -- z = M /
    (nM + M) is ~1e-5 on an image, so the fp16 rounding of its factors is far below the fp16 output's own rounding.  Against
    the same call with fp32 planes (a context whose column transforms are the run-time-plan kernels: no typed outputs) and
    against the oracle on the fp16-rounded input (deblurring.py:193-208), at the fp16 tolerance of every other test."""
    x, _ = synthetic_blurry_batch(2, 3, 1080, 1920, seed0=47)
    x16 = x.astype(np.float16)
    kw = dict(n_iter=2, c=0.362, b=0.468, alpha=6.0, beta=1.0, remove_halo=True)
    a, ia = engines["default"].polyblur(x16, engines["default"].make_options(**kw), want_info=True)
    b, ib = runtime_plan_engine.polyblur(x16, runtime_plan_engine.make_options(**kw), want_info=True)
    assert np.array_equal(ia["theta"], ib["theta"])
    d = np.abs(a.astype(np.float32) - b.astype(np.float32))
    assert d.max() <= 1.0 / 1024 and np.mean(d > 0) < 1e-3, (float(d.max()), float(np.mean(d > 0)))    # at most one fp16 step, on a handful of samples
    want = ref.polyblur_deblurring(x16[:1].astype(np.float32), n_iter=2, c=0.362, b=0.468, alpha=6, beta=1, remove_halo=True)
    assert np.max(np.abs(a[:1].astype(np.float32) - want)) < 1e-3


@pytest.mark.parametrize("shape", [(1, 3, 720, 1280), (8, 1, 720, 1280), (1, 3, 1440, 2560), (4, 1, 1440, 2560), (1, 1, 1024, 1024),
                                   (6, 1, 1024, 2048), (1, 3, 1200, 1600), (1, 3, 2048, 2048), (3, 1, 2048, 2048), (2, 3, 512, 640),
                                   (12, 1, 512, 640), (1, 3, 960, 768), (8, 1, 960, 768), (1, 1, 3072, 4096), (2, 1, 4096, 3072),
                                   (1, 3, 1600, 800), (6, 1, 1600, 800), (1, 3, 1536, 2048), (1, 3, 1280, 1200), (1, 1, 2560, 1440),
                                   (10, 1, 640, 512), (10, 1, 768, 960), (10, 1, 800, 1024)])
def test_fixed_plans_of_the_common_sizes_are_bit_identical(engines, runtime_plan_engine, runtime_plan_rows_engine, shape):
    """the one-plan kernels of csrc/lines_fixed.hip beyond the BASELINE line lengths (512 ... 4096-point lines: 720p, 1440p, 2K
    ...; narrow tiles for a lone image, wide ones for a batch that fills the chip) against the run-time-plan kernels: records
    and both gradient planes bit-identical, and the gradients within tolerance of the oracle (filters.py:159-186)"""
    B, C, H, W = shape
    nd = min(B, 2)
    img, _ = synthetic_blurry_batch(nd, C, H, W, seed0=53)
    img = np.concatenate([img] * ((B + nd - 1) // nd))[:B]
    o = opts(c=0.362, b=0.468)
    b = engines["default"].estimate_blur(img, o)
    for ref_eng in (runtime_plan_engine, runtime_plan_rows_engine):
        a = ref_eng.estimate_blur(img, o)
        for f in FIELDS:
            assert np.array_equal(np.asarray(a[f]), np.asarray(b[f])), f
    planes = img.reshape(B * C, H, W)[:2]
    gxb, gyb = engines["default"].fourier_gradients(planes)
    gxa, _ = runtime_plan_rows_engine.fourier_gradients(planes)
    _, gya = runtime_plan_engine.fourier_gradients(planes)
    assert np.array_equal(gxa, gxb) and np.array_equal(gya, gyb)
    if H * W <= 1440 * 2560:
        rx, ry = ref.spectral_gradients(planes[None])
        assert np.max(np.abs(gxb - rx[0])) < 2e-5 and np.max(np.abs(gyb - ry[0])) < 2e-5
