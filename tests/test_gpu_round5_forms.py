"""Round-5 forms against the forms they replace (same box, same inputs, through the C ABI).

* the parameter kernel's SHORT chain (PB_EST_LEAN, csrc/estimate.hip: blur_params_kernel): the workgroups that form the
  spectra no longer form the record first -- record, selection and output must not change by a bit
  (blur_estimation.py:138-232);
* PolySpec.always (PB_POLY_ALWAYS, csrc/conv.hip: pb_launch_conv_poly): two window launches per polynomial and nothing else
  for large images under the wrap boundary -- against the call that issues every launch its records might need, and
  against the oracle (deblurring.py:139-169);
* the domain-transform row pass with the row in registers (PB_DT_ROWS_REG, csrc/filters.hip: dt_rows_reg_kernel): bit-identical
  to the pass through global memory (domain_transform.py:56-85);
* method='direct' (the zero boundary, filters.py:40-49; the reference's own choice on a GPU, main.py:109-112) as one window
  pass plus three Horner steps over the border ring (PB_ZERO_RING, csrc/conv.hip + conv_wfft.hip: ring_live) against three
  steps over the whole image, and against the oracle."""
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
    return {"default": _engine(), "full_record": _engine(PB_EST_LEAN=0), "every_launch": _engine(PB_POLY_ALWAYS=0),
            "dt_global": _engine(PB_DT_ROWS_REG=0), "dt_rows_wave": _engine(PB_DT_ROWS_REG=2), "dt_rows_regw": _engine(PB_DT_ROWS_REG=3), "dt_cols_sweeps": _engine(PB_DT_COLS_STRIP=0, PB_DT_COLS_COOP=0), "dt_cols_reread": _engine(PB_DT_COLS_STRIP=2, PB_DT_COLS_COOP=0), "dt_cols_threads": _engine(PB_DT_COLS_COOP=0), "dt_cols_coop": _engine(PB_DT_COLS_COOP=2), "taper_three_steps": _engine(PB_POLY_PADDED=0), "taper_full_blends": _engine(PB_TAPER_RING=0), "direct_three_steps": _engine(PB_ZERO_RING=0), "direct_ring": _engine(PB_ZERO_RING_MIN_PAIRS=1)}


KW = dict(c=0.362, b=0.468, alpha=6.0, beta=1.0)
FIELDS = ("mags", "interp", "theta", "sigma", "rho", "kernel", "gray_min", "gray_max", "separable", "radius", "nphase", "kx", "ky")


def _run(eng, x, **kw):
    """(output, per-iteration records: a structured array (n_iter, B) of pb_blur_info)"""
    return eng.polyblur(x, eng.make_options(**kw), want_info=True)


@pytest.mark.parametrize("shape,dtype", [((1, 3, 720, 1280), np.float32), ((2, 3, 1080, 1920), np.float32), ((1, 3, 736, 1290), np.float16),
                                         ((1, 1, 1600, 2000), np.float32)])
def test_short_chain_and_two_launches_change_nothing(engines, shape, dtype):
    """the class PolySpec.always covers (every size since PB_POLY_MIN_PAIRS128 = 1; these are the sizes of its first form,
    150 window pairs and more): the default call, the call whose parameter
    kernel forms the whole record first, and the call that issues every launch are bit-identical -- output, records
    (stencil parts included: the record workgroup forms them beside the short chain) and selections"""
    B, C, H, W = shape
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=91)
    x = x.astype(dtype)
    outs = {}
    for name in ("default", "full_record", "every_launch"):
        out, infos = _run(engines[name], x, n_iter=3, **KW)
        sels = [engines[name].body_selection(B, k).copy() for k in range(3)]
        outs[name] = (out, infos, sels)
    a = outs["default"]
    for name in ("full_record", "every_launch"):
        b = outs[name]
        assert np.array_equal(a[0], b[0]), name
        for ia, ib in zip(a[1], b[1]):
            for f in FIELDS:
                assert np.array_equal(np.asarray(ia[f]), np.asarray(ib[f])), (name, f)
    # every image took a one-pass form in the default call, on the windows the model priced lower
    for k in range(3):
        s = a[2][k]
        assert (s[:, 0] == 1).all() and (s[:, 3] != 0).all(), s
        # (the call that issues every launch keeps the stencil / three-step forms in its model: where it, too, chose one pass,
        # the windows and halos agree)
        t = outs["every_launch"][2][k]
        same = t[:, 3] != 0
        assert np.array_equal(s[same][:, 3:6], t[same][:, 3:6])
        assert np.array_equal(s[:, 2], outs["full_record"][2][k][:, 2])              # `strip`: the record workgroup's word
    # ... and the oracle agrees (fp32: 2e-5 after three iterations, identical theta sequence; fp16 I/O: 1e-3)
    want, winfos = ref.polyblur_deblurring(x.astype(np.float32), n_iter=3, return_info=True, **KW)
    tol = 2e-5 if dtype == np.float32 else 1e-3
    assert np.abs(a[0].astype(np.float32) - want).max() < tol
    for ia, iw in zip(a[1], winfos):
        assert np.array_equal(np.asarray(ia["theta"], np.float32).reshape(-1), np.asarray(iw["theta"], np.float32).reshape(-1))


def test_wide_rank1_kernels_take_one_pass_too(engines):
    """sigma = rho = 4 (a flat image region: the clamp of blur_estimation.py:171-185): the widest composite there is (halo 36)
    -- under PolySpec.always it runs on 128 x 128 windows with 56 x 56 tiles instead of three stencil passes"""
    yy, xx = np.meshgrid(np.arange(800, dtype=np.float64), np.arange(1400, dtype=np.float64), indexing="ij")
    g = 0.5 + 0.4 * np.sin(2 * np.pi * xx / 1400) * np.cos(2 * np.pi * yy / 800)                  # periodic, no edges: the gradient
    x = np.stack([g, 0.9 * g, 0.8 * g])[None].astype(np.float32)                                  # maxima are ~0.005 -> sigma = rho = 4
    out, infos = _run(engines["default"], x, n_iter=2, **KW)
    assert float(infos[0]["sigma"][0]) == 4.0 and float(infos[0]["rho"][0]) == 4.0, (infos[0]["sigma"], infos[0]["rho"])
    s = engines["default"].body_selection(1, 0)
    assert s[0, 0] == 1 and s[0, 3] == 2, s
    want = ref.polyblur_deblurring(x, n_iter=2, **KW)
    assert np.abs(out - want).max() < 2e-5
    out2, _ = _run(engines["every_launch"], x, n_iter=2, **KW)
    assert np.abs(out - out2).max() < 5e-6                                              # (another form: rounding only)


@pytest.mark.parametrize("shape", [(2, 3, 37, 203), (1, 3, 90, 1920), (1, 1, 33, 1024), (3, 3, 5, 2048), (1, 3, 17, 64), (2, 1, 9, 2),
                                   (1, 3, 21, 3840), (2, 3, 3, 4096), (1, 1, 7, 2500), (1, 3, 4, 4100), (1, 3, 3, 7680), (1, 1, 2, 8192),
                                   (2, 3, 5, 257), (1, 3, 6, 1025), (1, 1, 4, 8200)])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_dt_rows_register_form(engines, shape, dtype):
    """the domain transform's row pass with the row in registers (csrc/filters.hip) against the pass through global memory
    (domain_transform.py:56-85) -- the same bits -- and the oracle: one wave per row (dt_rows_reg_kernel: rows of up to 1024, 2048,
    4096 samples) and a row over the four waves of a workgroup (dt_rows_regw_kernel: what few rows get; up to 8192 samples) --
    ragged tails, exactly-full chunks, the widest rows each form takes and one sample beyond, two-sample rows; fp32 and fp16"""
    rng = np.random.default_rng(23)
    x = rng.random(shape, dtype=np.float32).astype(dtype)
    want = engines["dt_global"].dt_recursive_filter(x, 2.0, 0.8, 1)
    want3 = engines["dt_global"].dt_recursive_filter(x, 6.0, 0.4, 3)                    # (later iterations: the in-place pass)
    tol = 5e-6 if dtype == np.float32 else 1e-3
    assert np.abs(want.astype(np.float32) - ref.recursive_filter(x.astype(np.float32), 2.0, 0.8, 1)).max() < tol
    for form in ("default", "dt_rows_wave", "dt_rows_regw"):
        assert np.array_equal(want, engines[form].dt_recursive_filter(x, 2.0, 0.8, 1)), form
        assert np.array_equal(want3, engines[form].dt_recursive_filter(x, 6.0, 0.4, 3)), form


@pytest.mark.parametrize("shape", [(1, 3, 189, 68), (1, 3, 189, 80), (2, 3, 300, 72), (1, 1, 240, 76), (1, 3, 68, 189), (3, 3, 189, 90)])
def test_edgetaper_then_one_pass_on_narrow_images(engines, shape):
    """edgetaping=True on images a few tiles wide: the polynomial behind the blends reads the SECOND set of spectra (the estimation
    builds it under the polynomial's own spec, the first set holds the blends' kernels), and the job list of its window launch
    has to be sized under THAT spec -- sized under the first set's (halos up to 12) it was one pair per row short wherever a
    halo of 16 makes three tiles of what 12 makes two: widths 65 .. 80 lost their second and third planes (found by
    tools/sweep_random.py, case 41: 189 x 68).  Against the oracle, and against the call that issues every launch."""
    B, C, H, W = shape
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=787)
    kw = dict(n_iter=2, edgetaping=True, c=0.31, b=0.43, alpha=6.0, beta=3.0)
    out, infos = _run(engines["default"], x, **kw)
    want, winfos = ref.polyblur_deblurring(x, return_info=True, **kw)
    for ia, iw in zip(infos, winfos):
        assert np.array_equal(np.asarray(ia["theta"], np.float32).reshape(-1), np.asarray(iw["theta"], np.float32).reshape(-1))
    assert np.abs(out - want).max() < 2e-5
    out2, _ = _run(engines["every_launch"], x, **kw)
    assert np.abs(out - out2).max() < 5e-6


@pytest.mark.parametrize("shape", [(1, 3, 189, 72), (2, 3, 400, 600)])
def test_edgetaper_fp16_output_when_every_launch_is_issued(engines, shape):
    """fp16 images with edgetaping=True under PB_POLY_ALWAYS=0: the last iteration's polynomial stores fp16 while its first step
    stores fp32, so the one-pass images wait for a launch of their own behind the three steps (csrc/conv.hip: pb_launch_conv_poly)
    -- which must follow the set of spectra the steps read (the second, behind an edgetaper), not the first set's spec"""
    B, C, H, W = shape
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=787)
    x = x.astype(np.float16)
    kw = dict(n_iter=2, edgetaping=True, c=0.31, b=0.43, alpha=6.0, beta=3.0)
    want = ref.polyblur_deblurring(x.astype(np.float32), **kw)
    for name in ("default", "every_launch"):
        out, _ = _run(engines[name], x, **kw)
        assert np.abs(out.astype(np.float32) - want).max() < 1e-3, name


@pytest.mark.parametrize("shape,dtype", [((1, 3, 720, 1280), np.float32), ((2, 3, 800, 1000), np.float16), ((1, 3, 200, 300), np.float32)])
def test_edgetaper_copies_and_one_pass(engines, shape, dtype):
    """edgetaping=True: the blends copy every tile pair on which alpha is exactly 1 (edgetaper.py:10-23: everything further than
    24 samples from the padded border), and the polynomial that follows takes ONE window pass from the padded, tapered plane
    where the image is large enough -- against three Horner steps there (rounding only), and against the oracle"""
    B, C, H, W = shape
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=12)
    x = x.astype(dtype)
    kw = dict(KW, n_iter=2, edgetaping=True)
    a, ia = _run(engines["default"], x, **kw)
    b, ib = _run(engines["taper_three_steps"], x, **kw)
    c, _ = _run(engines["taper_full_blends"], x, **kw)              # (the second and third blend over the whole plane: the same bits)
    assert np.array_equal(a, c)
    tol = 2e-5 if dtype == np.float32 else 1e-3
    assert np.abs(a.astype(np.float32) - b.astype(np.float32)).max() < (5e-6 if dtype == np.float32 else 1e-3)
    want, winfos = ref.polyblur_deblurring(x.astype(np.float32), return_info=True, **kw)
    assert np.abs(a.astype(np.float32) - want).max() < tol
    for i, w in zip(ia, winfos):
        assert np.array_equal(np.asarray(i["theta"], np.float32).reshape(-1), np.asarray(w["theta"], np.float32).reshape(-1))
    if H >= 720:                                                       # (asserted for the class's original sizes)
        s = engines["default"].body_selection(B, 1)
        assert (s[:, 0] == 1).all() and (s[:, 3] != 0).all(), s


@pytest.mark.parametrize("shape,dtype,extra", [((1, 3, 720, 1280), np.float32, {}), ((2, 3, 1080, 1920), np.float32, {}),
                                               ((1, 3, 736, 1290), np.float16, {}), ((1, 1, 1000, 1500), np.float32, dict(ker_size=13)),
                                               ((1, 3, 800, 1200), np.float32, dict(remove_halo=True)), ((1, 3, 700, 1100), np.float32, dict(edgetaping=True))])
def test_zero_boundary_ring(engines, shape, dtype, extra):
    """the frame within 12 samples of the image border is where the truncation of every Horner step to the padded domain
    matters: it must come out as from three steps over the whole image (rounding only: the interior is one window pass
    instead of three), and the oracle's method='direct' must be matched end to end -- also with a smaller kernel grid
    (pad 6), with halo masking behind the polynomial and with an edgetaper before it"""
    B, C, H, W = shape
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=77)
    x = x.astype(dtype)
    from polyblur_amd import _capi as capi
    kw = dict(KW, n_iter=2, boundary=capi.PB_ZERO, **extra)
    a, ia = _run(engines["direct_ring"], x, **kw)                  # (the ring form at every size: by default only from 4096 three-step window pairs per image)
    b, ib = _run(engines["direct_three_steps"], x, **kw)
    d = np.abs(a.astype(np.float32) - b.astype(np.float32))
    assert d.max() < (5e-6 if dtype == np.float32 else 1e-3), (d.max(), np.unravel_index(d.argmax(), d.shape))
    want, winfos = ref.polyblur_deblurring(x.astype(np.float32), n_iter=2, method="direct", return_info=True, **KW, **extra)
    e = np.abs(a.astype(np.float32) - want)
    assert e.max() < (2e-5 if dtype == np.float32 else 1e-3), (e.max(), np.unravel_index(e.argmax(), e.shape))
    for i, w in zip(ia, winfos):
        assert np.array_equal(np.asarray(i["theta"], np.float32).reshape(-1), np.asarray(w["theta"], np.float32).reshape(-1))
    if not extra.get("edgetaping") and C == 3:                              # (asserted for the class's original sizes)
        s0 = engines["direct_ring"].body_selection(B, 0)
        assert (s0[:, 0] == 1).all() and (s0[:, 3] != 0).all(), s0       # the interior took a one-pass form


def test_zero_boundary_ring_strong_blur(engines):
    """a strongly blurred image (sigma ~ 3.5: composite halos up to 36, three-step halos 12): the widest ring there is"""
    from polyblur_amd import _capi as capi
    from polyblur_amd.synthetic import synthetic_blurry_image
    x = np.stack([synthetic_blurry_image(3, 900, 1300, 5, blur=(3.5, 2.5, 48.0))[0]])
    kw = dict(KW, n_iter=2, boundary=capi.PB_ZERO)
    a, _ = _run(engines["direct_ring"], x, **kw)
    b, _ = _run(engines["direct_three_steps"], x, **kw)
    assert np.abs(a - b).max() < 5e-6
    assert np.abs(a - ref.polyblur_deblurring(x, n_iter=2, method="direct", **KW)).max() < 2e-5
    c, _ = _run(engines["default"], x, **kw)                           # (below the default threshold: three plain steps)
    assert np.array_equal(c, b)


def test_zero_boundary_ring_at_the_default_threshold(engines):
    """the tests above force the ring form with PB_ZERO_RING_MIN_PAIRS=1; at the DEFAULT threshold (4096
    three-step window pairs per image: 2800 x 1600 x 3 is 4200) only bench.py's context entry ran it.  The default context on
    such an image: the interior took a one-pass form, the result is what three steps over the whole image give (rounding
    only) for n_iter = 3, and one iteration matches the oracle's method='direct' (filters.py:40-49; main.py:109-112)."""
    from polyblur_amd import _capi as capi
    x, _ = synthetic_blurry_batch(1, 3, 1600, 2800, seed0=79)
    kw = dict(KW, boundary=capi.PB_ZERO)
    a, _ = _run(engines["default"], x, n_iter=3, **kw)
    s0 = engines["default"].body_selection(1, 0)
    assert (s0[:, 0] == 1).all() and (s0[:, 3] != 0).all(), s0          # the window pass + ring form, not three plain steps
    b, _ = _run(engines["direct_three_steps"], x, n_iter=3, **kw)
    assert np.abs(a - b).max() < 8e-6
    a1, _ = _run(engines["default"], x, n_iter=1, **kw)
    assert np.abs(a1 - ref.polyblur_deblurring(x, n_iter=1, method="direct", **KW)).max() < 2e-5


# ---- which of a plan's radices the line transforms' first / last stage take (csrc/estimate.hip: rows_plan, launch_cols) ----
STAGE_ORDER_SHAPES = [(1, 1, 512, 512), (1, 1, 720, 1280), (1, 1, 1080, 1920), (1, 1, 700, 500), (1, 1, 1200, 1280),
                      (1, 1, 1024, 2048), (1, 1, 2160, 1440), (3, 1, 600, 360)]


@pytest.fixture(scope="module")
def greedy_engine():
    return _engine(PB_FFT_FIRST=0, PB_FFT_FIRST_ROWS=0)


@pytest.mark.parametrize("shape", STAGE_ORDER_SHAPES)
def test_stage_order_of_the_line_transforms(engines, greedy_engine, shape):
    """the smallest radix first (2 at 512, 3 at 720, 5 at 1280 / 1200, 6 at 1080 / 1440, 7 at 700, 8 at 1920 / 2048, 9 at 2160,
    15 at 3840-class lines) is the same transform as the greedy order's -- filters.py:159-186 -- to rounding: both against the
    oracle, and against each other"""
    rng = np.random.default_rng(23)
    x = rng.random(shape, dtype=np.float32)
    rx, ry = ref.spectral_gradients(x)
    scale = max(1.0, float(np.abs(rx).max()), float(np.abs(ry).max()))
    gx, gy = engines["default"].fourier_gradients(x)
    hx, hy = greedy_engine.fourier_gradients(x)
    for a, b in ((gx, rx), (gy, ry), (hx, rx), (hy, ry), (gx, hx), (gy, hy)):
        assert np.abs(a - b).max() < 4e-6 * scale, float(np.abs(a - b).max())


def test_stage_order_does_not_depend_on_the_batch(engines):
    """the order of the stages decides the roundings, so it is a function of the line length alone: an image's records and
    output are the same bits alone (256-thread row workgroups, 512-thread column workgroups) and in a batch (128 / 1024)"""
    B, C, H, W = 12, 3, 360, 640
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=5)
    eng = engines["default"]
    full, infos = _run(eng, x, n_iter=2, **KW)
    for i in (0, 7, 11):
        one, oinfos = _run(eng, x[i:i + 1], n_iter=2, **KW)
        assert np.array_equal(full[i:i + 1], one), i
        for k in range(2):
            for f in ("mags", "theta", "sigma", "rho"):
                assert np.array_equal(np.asarray(infos[k][f])[i], np.asarray(oinfos[k][f])[0]), (i, k, f)


DT_COLUMN_FORMS = ("default", "dt_cols_threads", "dt_cols_reread", "dt_cols_coop")


@pytest.mark.parametrize("shape,dtype", [((1, 3, 720, 1280), np.float32), ((2, 3, 203, 333), np.float16), ((1, 1, 1081, 700), np.float32),
                                         ((3, 3, 64, 97), np.float32), ((1, 3, 65, 40), np.float16), ((1, 1, 1600, 2100), np.float32),
                                         ((2, 3, 128, 70), np.float32), ((1, 3, 161, 300), np.float32), ((2, 3, 203, 336), np.float16),
                                         ((1, 3, 31, 64), np.float32), ((2, 1, 96, 136), np.float16)])
def test_dt_columns_in_strips(engines, shape, dtype):
    """the column pass of the domain transform (domain_transform.py:56-85) in its forms (csrc/filters.hip) against the two sweeps
    through global memory -- the same bits -- and the oracle:
    * one thread per column, the down sweep's values formed again strip by strip from one carry per strip: dt_cols_down_kernel,
      then dt_cols_up_kernel with the weights formed again from J over 16 rows, or -- three fp32 channels, 128 rows and up --
      dt_cols_upw_kernel with the weights the down sweep stored, over 32 rows;
    * few columns (what a small batch gets; forced here): dt_cols_coop_kernel, a workgroup per 16 (one channel: 64) columns that
      streams blocks of 32 rows through LDS -- widths that are and are not whole groups, where rows are not on 16-byte boundaries
      the per-column forms stand in;
    heights that are and are not multiples of the strip, one and three channels, one and three iterations of the filter, a joint
    image of its own"""
    rng = np.random.default_rng(29)
    x = rng.random(shape, dtype=np.float32).astype(dtype)
    jt = rng.random(shape, dtype=np.float32).astype(dtype)
    tol = 5e-6 if dtype == np.float32 else 1e-3
    want = engines["dt_cols_sweeps"].dt_recursive_filter(x, 2.0, 0.8, 1)
    want3 = engines["dt_cols_sweeps"].dt_recursive_filter(x, 6.0, 0.4, 3)
    wantj = engines["dt_cols_sweeps"].dt_recursive_filter(x, 3.0, 0.5, 2, joint=jt)   # (the prefilter's later calls: deblurring.py:80-88)
    assert np.abs(want.astype(np.float32) - ref.recursive_filter(x.astype(np.float32), 2.0, 0.8, 1)).max() < tol
    assert np.abs(wantj.astype(np.float32) - ref.recursive_filter(x.astype(np.float32), 3.0, 0.5, 2, jt.astype(np.float32))).max() < tol
    for form in DT_COLUMN_FORMS:
        assert np.array_equal(want, engines[form].dt_recursive_filter(x, 2.0, 0.8, 1)), form
        assert np.array_equal(want3, engines[form].dt_recursive_filter(x, 6.0, 0.4, 3)), form
        assert np.array_equal(wantj, engines[form].dt_recursive_filter(x, 3.0, 0.5, 2, joint=jt)), form


@pytest.mark.parametrize("kind", ["constant", "nan"])
def test_degenerate_images_under_the_one_pass_class(engines, kind):
    """under PolySpec.always the host vouches for point-symmetric taps and no stencil launch stands behind the
    window launches.  The device still looks (csrc/khat.h: the symmetry reduction is kept, a violation is reported as
    pb_body_selection's second column = -1).  A constant image has no gradient: the reference's estimate is NaN (0 / 0 in
    blur_estimation.py:189-208, the oracle agrees); the engine's clamps (sigma, rho in [0.3, 4], fminf / fmaxf drop a NaN)
    give it the widest isotropic Gaussian instead -- symmetric taps, not flagged, and a polynomial whose taps sum to 1 leaves
    the constant where it is.  With a NaN sample in it the image is garbage in, garbage out; either way the good image in the
    same batch is neither flagged nor changed by a bit, and nothing hangs or faults."""
    eng = engines["default"]
    good, _ = synthetic_blurry_batch(1, 3, 240, 320, seed0=5)
    bad = np.full((1, 3, 240, 320), 0.5, np.float32)
    if kind == "nan":
        bad[0, 1, 100, 100] = np.nan
    x = np.concatenate([good, bad])
    out, _ = _run(eng, x, n_iter=2, **KW)
    sel = eng.body_selection(2, 0)
    assert sel[0, 1] != -1, sel
    if kind == "constant":
        assert sel[1, 1] != -1 and np.max(np.abs(out[1] - 0.5)) < 1e-5, (sel, float(np.max(np.abs(out[1] - 0.5))))
    alone, _ = _run(eng, good, n_iter=2, **KW)
    assert eng.body_selection(1, 0)[0, 1] != -1
    assert np.array_equal(out[:1], alone)
    assert np.isfinite(out[:1]).all()
    want = ref.polyblur_deblurring(good, n_iter=2, c=0.362, b=0.468, alpha=6, beta=1)
    assert np.max(np.abs(out[:1] - want)) < 2e-5
