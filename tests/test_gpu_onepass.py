"""GPU parity of the tile-spectrum wave body with per-axis run-time halos and of the general one-pass polynomial, on 64 x 64
and on 128 x 128 windows (csrc/conv_wfft.hip, csrc/conv_w128.hip, csrc/khat.h; round 4).

Under the wrap boundary the reference's deconvolution is ONE filter a3 K^3 + a2 K^2 + a1 K + b (its own 'fft' form,
deblurring.py:139-169).  The engine measures that composite filter's halo per axis (marginals of |taps|, convolution
powers) and takes the polynomial as one window pass wherever that is cheaper than three passes with the kernel's own
halos.  Everything here goes through the C ABI and is checked against the NumPy oracle (reference: deblurring.py:122-169,
211-239; filters.py:14-49), against the three-step form (PB_POLY1=0), image by image against the same image alone, and
at the sizes where windows are ragged, clipped or off the 16-byte path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd import _capi as capi
from polyblur_amd.synthetic import synthetic_blurry_batch


def _engine(mode):
    """a context with PB_POLY1=mode; PB_POLY_MIN_PAIRS128=0 lets images of any size take 128 x 128 windows (by default only
    images of 150 window pairs or more do -- 720p x 3 channels and up; tests/test_gpu_fullsize.py runs those through the default context)"""
    from polyblur_amd.engine import Engine
    old = {k: os.environ.get(k) for k in ("PB_POLY1", "PB_POLY_MIN_PAIRS128")}
    os.environ["PB_POLY1"] = str(mode)
    os.environ["PB_POLY_MIN_PAIRS128"] = "0"
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
    one, three = _engine(3), _engine(0)
    yield one, three
    one.close()
    three.close()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# (halos: the radius beyond which the composite's |tap| mass is < 1e-8, csrc/khat.h KH_HALO_TOL; 1e-10 until round 5)
# (theta deg, sigma, rho) -> what the default context does with it under full support: (form, halo x, halo y); form 0 = three
# steps, 1 = one pass on 64 x 64 windows, 2 = one pass on 128 x 128 windows (csrc/conv_w128.hip)
KERNELS = [
    ((66.0, 2.095, 1.314), (2, 16, 20)),      # the headline's first estimate: a 64 x 64 window would keep 32 x 24 samples
    ((66.0, 1.656, 1.009), (1, 12, 16)),      # its second: 40 x 32 of a 64 x 64 window prices like 104 x 96 of a 128 x 128 one
    ((66.0, 1.240, 0.625), (1, 8, 12)),       # its third
    ((0.0, 1.4, 0.9), (1, 16, 10)),           # rank-1 kernel: its polynomial is not rank-1 -- one pass beats three stencil passes
    ((30.0, 0.65, 0.40), (1, 8, 6)),
    ((0.0, 0.3, 0.3), (1, 4, 4)),             # the clamped isotropic estimate
    ((90.0, 1.2, 0.5), (1, 8, 12)),           # rows much wider than columns
    ((45.0, 3.0, 1.0), (2, 24, 24)),
    ((0.0, 4.0, 4.0), (0, 12, 12)),           # the widest kernel: composite halo 36 -- three stencil passes
]


@pytest.mark.parametrize("shape,dtype", [((1, 3, 1080, 1920), np.float32), ((2, 1, 301, 517), np.float32),
                                         ((1, 3, 150, 210), np.float32), ((1, 3, 520, 776), np.float16),
                                         ((1, 3, 1080, 1920), np.float16)])
def test_polynomial_against_oracle_and_three_steps(engines, shape, dtype):
    one, three = engines
    B = shape[0]
    x, _ = synthetic_blurry_batch(*shape, seed0=91)
    xin = x.astype(dtype)
    tol, tol2 = (5e-6, 8e-6) if dtype == np.float32 else (6e-4, 1e-3)
    for (deg, sg, rh), (poly, hx, hy) in KERNELS:
        th = [np.float32(np.deg2rad(deg))] * B
        outs = []
        for eng in (one, three):
            buf = eng.make_kernels([sg] * B, [rh] * B, th, support=capi.PB_SUPPORT_FULL)
            info = eng.read_info(buf, B)
            outs.append(eng.inverse_filter(xin, buf, 6.0, 1.0, capi.PB_WRAP).astype(np.float32))
            sel = eng.body_selection(B)
            if eng is one:
                assert (sel[:, 3] == poly).all() and ((sel[:, 4] == hx).all() and (sel[:, 5] == hy).all()), (deg, sg, rh, sel)
            else:
                assert (sel[:, 3] == 0).all()
        want = ref.inverse_filtering_rank3(xin.astype(np.float32), info["kernel"][:, None], 6.0, 1.0, method="fft")
        assert maxabs(outs[0], want) < tol, (deg, sg, rh, maxabs(outs[0], want))
        assert maxabs(outs[1], want) < tol, (deg, sg, rh, maxabs(outs[1], want))
        assert maxabs(outs[0], outs[1]) < tol2


def test_mixed_batch_and_each_image_alone(engines):
    """images of both one-pass forms in one batch (128 x 128 and 64 x 64 windows; dense and rank-1 kernels): every image's
    result is bit for bit what it gets alone, and the other boundary model and other coefficients rebuild what they need"""
    one, _ = engines
    x, _ = synthetic_blurry_batch(5, 3, 420, 660, seed0=82)
    sg, rh = [1.656, 0.6, 2.5, 4.0, 4.0], [1.009, 0.4, 1.2, 4.0, 2.0]
    th = [np.float32(np.deg2rad(66.0)), np.float32(0.5), np.float32(1.0), np.float32(0.0), np.float32(np.deg2rad(45.0))]
    buf = one.make_kernels(sg, rh, th, support=capi.PB_SUPPORT_FULL)
    info = one.read_info(buf, 5)
    k = info["kernel"][:, None]
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    sel = one.body_selection(5)
    # 64 x 64 one pass (twice), 128 x 128 one pass, three rank-1 stencil passes, and -- the widest composite there
    # is, halos (32, 32): 64 x 64 tiles -- 128 x 128 one pass again (three tile-spectrum passes until the cost model learnt what
    # a Horner step costs; that form is still what the zero boundary below, PB_POLY1=0 and small images by default take)
    assert sel[:, 3].tolist() == [1, 1, 2, 0, 2] and sel[:, 0].tolist() == [1, 1, 1, 0, 1], sel
    assert sel[4, 4:6].tolist() == [32, 32], sel
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, method="fft")) < 8e-6
    for i in range(5):
        b1 = one.make_kernels(sg[i:i + 1], rh[i:i + 1], th[i:i + 1], support=capi.PB_SUPPORT_FULL, name="one.info")
        alone = one.inverse_filter(x[i:i + 1], b1, 6.0, 1.0, capi.PB_WRAP)
        assert np.array_equal(alone, got[i:i + 1]), i
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_ZERO)                       # zero boundary: three steps, the kernel's own spectrum
    assert (one.body_selection(5)[:, 3] == 0).all()
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, method="direct")) < 8e-6
    got = one.inverse_filter(x, buf, 2.0, 3.0, capi.PB_WRAP)                       # other coefficients: other spectra, other halos
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 2.0, 3.0, method="fft")) < 8e-6
    got = one.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP, edgetaping=True)
    assert maxabs(got, ref.inverse_filtering_rank3(x, k, 6.0, 1.0, do_edgetaper=True, method="fft")) < 1e-5


@pytest.mark.parametrize("shape,dtype,tol", [((1, 3, 1080, 1920), np.float32, 2e-5), ((2, 3, 333, 517), np.float32, 2e-5),
                                             ((1, 3, 536, 712), np.float16, 1e-3), ((3, 1, 97, 131), np.float32, 2e-5)])
def test_whole_call(engines, shape, dtype, tol):
    """the whole call (device-built records: every launch is issued and finds its images on the device), three
    iterations whose kernels shrink -- three steps first, one pass later -- against the oracle with identical direction
    sequences and against the three-step context"""
    one, three = engines
    x, _ = synthetic_blurry_batch(*shape, seed0=17)
    xin = x.astype(dtype)
    kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
    o = one.make_options(**kw)
    got, info = one.polyblur(xin, o, want_info=True)
    sel = one.body_selection(shape[0])
    base, binfo = three.polyblur(xin, o, want_info=True)
    want, winfos = ref.polyblur_deblurring(xin.astype(np.float32), return_info=True, **kw)
    assert np.array_equal(info["theta"], binfo["theta"])
    assert [[float(t) for t in it["theta"]] for it in winfos] == info["theta"].tolist()
    assert maxabs(got.astype(np.float32), want) < tol and maxabs(base.astype(np.float32), want) < tol
    assert sel[:, 0].all()
    # each image alone gets the same bits (what an image gets does not depend on the batch it travels in)
    if shape[0] > 1:
        for i in range(shape[0]):
            assert np.array_equal(one.polyblur(xin[i:i + 1], o), got[i:i + 1]), i


def test_spectra_are_rebuilt_for_a_longer_run_of_records():
    """(advisor, round 3) spectra in the context's scratch are those of a record pointer AND a record count: estimate B
    records at A + B, then B records at A, then filter with the 2 B records at A -- the second half's spectra must be rebuilt,
    not taken from an earlier, larger batch"""
    from polyblur_amd.engine import Engine
    eng = Engine(0)
    try:
        Bh = 2
        x, _ = synthetic_blurry_batch(2 * Bh, 3, 200, 264, seed0=41)
        big, _ = synthetic_blurry_batch(3 * Bh, 3, 200, 264, seed0=51)
        o = eng.make_options(n_iter=1, c=0.362, b=0.468, alpha=6, beta=1)
        eng.polyblur(big, o)                                                   # a larger earlier batch grows the scratch
        buf = eng.info_buffer("run.info", 2 * Bh)
        rec = capi.INFO_DTYPE.itemsize
        import ctypes as C
        xd = eng.to_device("run.x", x)
        img = x[0].size * 4

        def estimate(first):
            eng._check(eng.lib.pb_estimate_blur(eng.ctx, C.c_void_p(xd.ptr + first * img), capi.PB_F32, Bh, 3, 200, 264, C.byref(o),
                                                C.c_void_p(buf.ptr + first * rec)))
        estimate(Bh)
        estimate(0)
        got = eng.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
        info = eng.read_info(buf, 2 * Bh)
        want = ref.inverse_filtering_rank3(x, info["kernel"][:, None], 6.0, 1.0, method="fft")
        assert maxabs(got, want) < 8e-6, maxabs(got, want)
    finally:
        eng.close()
