"""GPU parity of the one-pass polynomial in its first form (PB_POLY1=1 when the context is created: kernels within the
4-sample halo class only, either tile-spectrum form; 0 = never; the default, 2, is the general form with the composite
filter's own halos: tests/test_gpu_onepass.py).  Under the wrap boundary the reference's deconvolution is ONE filter
a3 K^3 + a2 K^2 + a1 K + b (deblurring.py:139-169); for a dense kernel within the 4-sample halo the composite's halo is at
most 12, so the image's three Horner launches become one window pass with the polynomial's spectrum (csrc/khat.h,
pb_launch_conv_poly).  It must agree with the oracle and with the three-step form, in batches that mix it with every other
body, and in the whole call."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd import _capi as capi
from polyblur_amd.synthetic import synthetic_blurry_batch


@pytest.fixture(scope="module")
def engines():
    from polyblur_amd.engine import Engine
    old = os.environ.get("PB_POLY1")
    os.environ["PB_POLY1"] = "1"
    try:
        one = Engine(0)
    finally:
        if old is None:
            del os.environ["PB_POLY1"]
        else:
            os.environ["PB_POLY1"] = old
    os.environ["PB_POLY1"] = "0"
    try:
        three = Engine(0)
    finally:
        if old is None:
            del os.environ["PB_POLY1"]
        else:
            os.environ["PB_POLY1"] = old
    yield one, three
    one.close()
    three.close()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# (B, C, H, W): a pass of the workgroup form (few window pairs), one of the wave-private form, widths off the 16-byte path
@pytest.mark.parametrize("shape", [(1, 3, 150, 210), (1, 3, 1080, 1920), (2, 1, 301, 517)])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_one_pass_matches_oracle_and_three_steps(engines, shape, dtype):
    one, three = engines
    B = shape[0]
    x, _ = synthetic_blurry_batch(*shape, seed0=81)
    sg, rh = [0.65, 0.5][:B], [0.4, 0.33][:B]
    th = [np.float32(np.deg2rad(30.0)), np.float32(np.deg2rad(105.0))][:B]
    xin = x.astype(dtype)
    outs = []
    for eng in (one, three):
        buf = eng.make_kernels(sg, rh, th, support=capi.PB_SUPPORT_ADAPTIVE)
        info = eng.read_info(buf, B)
        assert all(info["separable"] == 0) and all(info["radius"] == 4)
        outs.append(eng.inverse_filter(xin, buf, 6.0, 1.0, capi.PB_WRAP).astype(np.float32))
    want = ref.inverse_filtering_rank3(xin.astype(np.float32), info["kernel"][:, None], 6.0, 1.0, method="fft")
    tol = 5e-6 if dtype == np.float32 else 6e-4
    assert maxabs(outs[0], want) < tol, maxabs(outs[0], want)
    assert maxabs(outs[0], outs[1]) < (8e-6 if dtype == np.float32 else 1e-3)


def test_one_pass_in_a_mixed_batch_and_other_passes(engines):
    """one image per body: one-pass (radius 4), dense with full halo (three tile-spectrum steps), rank-1 (stencil), rank-1
    within the 4-sample halo (one pass); then
    passes that need the kernel's own spectrum again (zero boundary: three steps; edgetaper blends)"""
    one, three = engines
    x, _ = synthetic_blurry_batch(4, 3, 420, 660, seed0=82)
    # ... and the clamped isotropic kernel later iterations mostly find: rank-1, within the 4-sample halo -- one pass as well
    sg, rh, th = [0.6, 2.5, 2.0, 0.3], [0.4, 1.2, 1.0, 0.3], [np.float32(0.5), np.float32(1.0), np.float32(0.0), np.float32(0.9)]
    buf = one.make_kernels(sg, rh, th, support=capi.PB_SUPPORT_ADAPTIVE)
    info = one.read_info(buf, 4)
    assert list(info["separable"]) == [0, 0, 1, 1] and info["radius"][0] == 4 and info["radius"][1] > 8 and info["radius"][3] == 4
    k = info["kernel"][:, None]
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, method="fft")) < 8e-6
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_ZERO)
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, method="direct")) < 8e-6
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP, edgetaping=True)
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, do_edgetaper=True, method="fft")) < 1e-5
    got = one.inverse_filter(x, buf, 2.0, 3.0, capi.PB_WRAP)                       # other coefficients: other spectra
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 2.0, 3.0, method="fft")) < 8e-6


@pytest.mark.parametrize("shape", [(1, 3, 1080, 1920), (2, 3, 240, 320)])
def test_whole_call_on_a_mildly_blurred_image(engines, shape):
    """the whole call under the adaptive policy on a mildly, obliquely blurred image (the method's own use case): the
    estimates are like sigma 0.6 / rho 0.3 (clamped) at 30 degrees -- dense, within the 4-sample halo.  A context created
    without PB_POLY1 (the general form) takes one pass there as well, under either support policy."""
    from polyblur_amd.engine import Engine
    one, three = engines
    rng = np.random.default_rng(83)
    x = rng.random(shape, dtype=np.float32)
    x = ref.convolve2d(x, ref.gaussian_kernel_2d([np.float32(0.6)] * shape[0], [0.9] * shape[0], [0.5] * shape[0]), method="fft")
    x = np.clip(x, 0, 1).astype(np.float32)
    kw = dict(n_iter=2, c=0.4, b=0.468, alpha=6, beta=1)
    o = one.make_options(support=capi.PB_SUPPORT_ADAPTIVE, **kw)
    got, info = one.polyblur(x, o, want_info=True)
    base, binfo = three.polyblur(x, o, want_info=True)
    assert np.array_equal(info["theta"], binfo["theta"])
    took = (info["radius"] <= 4) & (info["separable"] == 0)
    assert took.any(), (info["radius"], info["separable"], info["sigma"], info["rho"])          # (the case under test occurs)
    assert not np.array_equal(got, base)                                          # (and was evaluated in the other form)
    assert maxabs(got, base) < 2e-5, maxabs(got, base)
    assert maxabs(got, ref.polyblur_deblurring(x, **kw)) < 3e-5
    auto = Engine(0)
    try:
        ad = auto.polyblur(x, o)
        assert not np.array_equal(ad, base) and maxabs(ad, base) < 2e-5 and maxabs(ad, ref.polyblur_deblurring(x, **kw)) < 3e-5
        assert (auto.body_selection(shape[0])[:, 3] == 1).all()                     # (one pass for every image, last iteration)
        # full support: these kernels have radius 8 / 6 there (taps count until they underflow), but the window halo follows
        # the taps that matter to overlap-save (< 1e-10 of the mass beyond them): one pass as well, every tap in the spectrum
        of = one.make_options(support=capi.PB_SUPPORT_FULL, **kw)
        af_, if_ = auto.polyblur(x, of, want_info=True)
        tf_ = three.polyblur(x, of)
        assert (if_["radius"] > 4).all()
        assert not np.array_equal(af_, tf_) and maxabs(af_, tf_) < 2e-5 and maxabs(af_, ref.polyblur_deblurring(x, **kw)) < 3e-5
        # the clamped isotropic estimate sigma = rho = 0.3 (c = 0.2 here), rank-1: its other taps underflow, radius 4 under
        # every policy: one pass
        kw2 = dict(kw, c=0.2)
        of2 = one.make_options(support=capi.PB_SUPPORT_FULL, **kw2)
        a2, i2 = auto.polyblur(x, of2, want_info=True)
        t2 = three.polyblur(x, of2)
        assert (i2["sigma"] == np.float32(0.3)).all() and (i2["radius"] == 4).all() and (i2["separable"] == 1).all()
        assert not np.array_equal(a2, t2) and maxabs(a2, t2) < 2e-5
        assert maxabs(a2, ref.polyblur_deblurring(x, **kw2)) < 2e-5
        # fp16 images: the last iteration stores fp16, its first step fp32 -- a launch of its own runs the one pass
        xh = x.astype(np.float16)
        h1 = auto.polyblur(xh, of2)
        assert np.array_equal(h1, auto.polyblur(xh, of2))
        assert (auto.body_selection(shape[0])[:, 3] == 1).all()
        assert maxabs(h1.astype(np.float32), ref.polyblur_deblurring(xh.astype(np.float32), **kw2)) < 1e-3
    finally:
        auto.close()
