"""GPU parity of the tile-spectrum body (csrc/conv_fft.hip): dense kernels evaluated per 64 x 64 window in the frequency
domain.  It must agree with the oracle's spatial convolution and with the 2-D stencil body it stands in for, under both
boundary models, every window halo class, every dtype it is built for, and on images smaller than one window."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd import _capi as capi
from polyblur_amd.synthetic import synthetic_blurry_batch


@pytest.fixture()
def eng():
    from polyblur_amd.engine import get_engine
    e = get_engine(0)
    yield e
    e.set_dense_eval("auto", capi.PB_DENSE_MIN_PHASES)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def both_ways(eng, fn):
    eng.set_dense_eval("auto", 0)                            # every dense kernel through the tile-spectrum body
    a = fn()
    eng.set_dense_eval("stencil")
    b = fn()
    return a, b


@pytest.mark.parametrize("sigma,rho,deg", [(3.0, 2.0, 66.0), (4.0, 0.3, 42.0), (1.3, 0.8, 60.0), (0.6, 0.4, 24.0), (2.0, 1.0, 12.0)])
@pytest.mark.parametrize("boundary,method", [(capi.PB_WRAP, "fft"), (capi.PB_ZERO, "direct")])
@pytest.mark.parametrize("support", [capi.PB_SUPPORT_FULL, capi.PB_SUPPORT_ADAPTIVE])
def test_inverse_filter_matches_oracle_and_stencil(eng, sigma, rho, deg, boundary, method, support):
    x, _ = synthetic_blurry_batch(1, 3, 150, 210, seed0=11)
    th = np.float32(deg) * np.float32(np.pi) / np.float32(180)
    k = ref.gaussian_kernel_2d([th], [sigma], [rho])
    buf = eng.make_kernels([sigma], [rho], [th], support=support)
    assert eng.read_info(buf, 1)["separable"][0] == 0
    a, b = both_ways(eng, lambda: eng.inverse_filter(x, buf, 6.0, 1.0, boundary))
    want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method=method)
    assert maxabs(a, want) < 1e-5
    assert maxabs(a, b) < 5e-6


@pytest.mark.parametrize("shape", [(1, 1, 8, 9), (1, 3, 33, 130), (2, 3, 65, 63), (1, 2, 40, 44), (1, 1, 200, 37), (3, 1, 97, 161)])
@pytest.mark.parametrize("boundary", [capi.PB_WRAP, capi.PB_ZERO])
def test_convolve2d_small_and_ragged(eng, shape, boundary):
    rng = np.random.default_rng(5)
    B, C, H, W = shape
    xp = rng.random((B, C, H + 24, W + 24), dtype=np.float32)
    th = np.deg2rad(np.float32([30.0, 75.0, 110.0][:B]))
    buf = eng.make_kernels([2.5, 1.2, 3.5][:B], [1.0, 0.7, 3.0][:B], th)
    a, b = both_ways(eng, lambda: eng.convolve2d(xp, buf, boundary))
    assert maxabs(a, b) < 2e-6


@pytest.mark.parametrize("symmetric", [True, False])
def test_caller_supplied_taps(eng, symmetric):
    # pb_set_kernels: any 25 x 25 taps.  Point-symmetric ones (real spectrum) take the tile-spectrum body, the others
    # keep the stencil body -- either way the result is the oracle's spatial correlation.
    rng = np.random.default_rng(9)
    k = rng.random((1, 25, 25), dtype=np.float32)
    if symmetric:
        k = k + k[:, ::-1, ::-1]
    k /= k.sum()
    xp = rng.random((1, 2, 120 + 24, 90 + 24), dtype=np.float32)
    buf = eng.set_kernels(k)
    a, b = both_ways(eng, lambda: eng.convolve2d(xp, buf, capi.PB_ZERO))
    assert maxabs(a, b) < 2e-6
    want = ref.convolve2d(xp, k[:, None], method="direct")
    assert maxabs(a, want) < 2e-6


def test_window_halo_follows_the_taps_not_the_marginals(eng):
    """The windows' halo is the radius outside which the taps' ABSOLUTE values are negligible (csrc/khat.h).  A point-symmetric
    kernel whose far taps cancel in both marginals -- +a at (-10, +10) and (+10, -10), -a at (+10, +10) and (-10, -10) -- around
    a narrow Gaussian must still get the full halo; and a narrow Gaussian alone, whose radius under full support is 8 (taps
    count until they underflow), runs on the windows of the 4-sample halo with the same result."""
    rng = np.random.default_rng(10)
    g = ref.gaussian_kernel_2d(np.float32([0.5]), [0.6], [0.4])[0]
    k = g.copy()
    for (u, v, sgn) in ((-10, 10, 1), (10, -10, 1), (10, 10, -1), (-10, -10, -1)):
        k[12 + u, 12 + v] += sgn * 0.05
    assert abs(k.sum(0)[22]) < 1e-9 and abs(k.sum(1)[22]) < 1e-9                       # the marginals do not see them
    xp = rng.random((1, 2, 200 + 24, 330 + 24), dtype=np.float32)
    for taps in (k, g):
        buf = eng.set_kernels(taps[None])
        eng.set_dense_eval("auto", 0)                        # every dense point-symmetric kernel through the tile-spectrum body
        try:
            a = eng.convolve2d(xp, buf, capi.PB_ZERO)
            w = eng.convolve2d(xp, buf, capi.PB_WRAP)
        finally:
            eng.set_dense_eval("auto", capi.PB_DENSE_MIN_PHASES)
        assert maxabs(a, ref.convolve2d(xp, taps[None, None], method="direct")) < 2e-6
        assert maxabs(w, ref.convolve2d(xp, taps[None, None], method="fft")) < 2e-6


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float16, 1e-3)])
@pytest.mark.parametrize("extra", [dict(), dict(edgetaping=True), dict(method="direct", remove_halo=True)])
def test_pipeline_both_ways(eng, dtype, tol, extra):
    import torch
    from polyblur_amd import polyblur_deblurring
    x, _ = synthetic_blurry_batch(2, 3, 240, 328, seed0=3)
    xt = torch.from_numpy(x.astype(dtype)).cuda()
    kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1, **extra)
    a, b = both_ways(eng, lambda: polyblur_deblurring(xt, **kw).float().cpu().numpy())
    want = ref.polyblur_deblurring(xt.float().cpu().numpy(), **kw)
    assert maxabs(a, want) < tol
    assert maxabs(a, b) < tol


def test_mixed_batch(eng):
    # one rank-1, one dense-small (below the threshold) and one dense-large kernel in the same launch
    x, _ = synthetic_blurry_batch(3, 3, 130, 170, seed0=2)
    th = np.deg2rad(np.float32([0.0, 24.0, 66.0]))
    sig, rho = [2.0, 0.6, 3.0], [1.0, 0.4, 2.0]
    buf = eng.make_kernels(sig, rho, th, support=capi.PB_SUPPORT_ADAPTIVE)
    eng.set_dense_eval("auto", 36)                           # (0.6, 0.4, 24 deg) has 19 live phases: stencil body
    out = eng.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    k = ref.gaussian_kernel_2d(th, sig, rho)
    want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method="fft")
    assert maxabs(out, want) < 1e-5


def test_host_side_record_facts_follow_the_records(eng):
    # pb_make_kernels / pb_set_kernels let the context remember which bodies a record set needs (a pass then skips the
    # launch nobody needs); whatever rewrites the records -- another build into the same buffer, a raw upload -- must
    # make it forget.  Same buffer, three kinds of records in turn, each result checked against the oracle.
    rng = np.random.default_rng(3)
    xp = rng.random((1, 2, 90 + 24, 130 + 24), dtype=np.float32)
    eng.set_dense_eval("auto", capi.PB_DENSE_MIN_PHASES)

    def check(buf, k):
        out = eng.convolve2d(xp, buf, capi.PB_ZERO)
        assert maxabs(out, ref.convolve2d(xp, k[:, None], method="direct")) < 2e-6

    th0, th1 = np.float32(0.0), np.deg2rad(np.float32(66.0))
    k_rank1 = ref.gaussian_kernel_2d([th0], [2.0], [1.0])
    k_dense = ref.gaussian_kernel_2d([th1], [3.0], [2.0])
    buf = eng.make_kernels([2.0], [1.0], [th0], name="np.facts")          # rank-1: stencil launch only
    check(buf, k_rank1)
    buf = eng.make_kernels([3.0], [2.0], [th1], name="np.facts")          # dense: tile-spectrum launch only
    dense_records = eng.read_info(buf, 1).copy()
    check(buf, k_dense)
    buf = eng.make_kernels([2.0], [1.0], [th0], name="np.facts")
    check(buf, k_rank1)
    buf.upload(dense_records)                                             # raw upload of the dense records over the rank-1 ones
    check(buf, k_dense)


def test_dense_eval_arguments(eng):
    bad = -1                                                                        # PB_ERR_BADARG
    assert eng.lib.pb_set_dense_eval(eng.ctx, 7, 0) == bad                          # unknown mode
    assert eng.lib.pb_set_dense_eval(eng.ctx, capi.PB_DENSE_AUTO, -1) == bad
    assert eng.lib.pb_set_dense_eval(eng.ctx, capi.PB_DENSE_AUTO, 10 ** 6) == bad
    with pytest.raises(KeyError):
        eng.set_dense_eval("spectrum")
    eng.set_dense_eval("stencil")
    eng.set_dense_eval("auto", 0)


def test_full_size_pass_is_deterministic(eng):
    """A 4K pass through the tile-spectrum body, 25 times on the same planes: bit-identical results.  (A 16-byte store whose
    data registers the next instruction rewrote lost single samples in about one launch out of two on MI355X; the location
    moved from launch to launch, so only repetition at full size sees it.)"""
    import ctypes as C
    import torch
    B, Hp, Wp = 1, 2160 + 24, 3840 + 24
    eng.set_dense_eval("auto", 0)
    xp = torch.rand(B, 3, Hp, Wp, device="cuda")
    buf = eng.make_kernels([2.1], [1.3], [np.deg2rad(np.float32(66.0))])
    first = None
    for _ in range(25):
        out = torch.full_like(xp, 7.0)
        eng._check(eng.lib.pb_convolve2d(eng.ctx, C.c_void_p(xp.data_ptr()), C.c_void_p(out.data_ptr()), B, 3, Hp, Wp, buf.ptr, capi.PB_WRAP))
        torch.cuda.synchronize()
        if first is None:
            first = out
        else:
            assert int((out != first).sum().item()) == 0


def test_rewriting_one_record_of_a_cached_set(eng):
    """Record facts are forgotten by RANGE: a B = 4 all-dense set is cached (the host knows that no image needs the
    stencil launch), then record 2 alone is rewritten as a rank-1 kernel through pb_make_kernels -- the whole set must
    forget, or the next pass would skip the stencil launch and leave image 2's output unwritten."""
    import ctypes as C
    B = 4
    x, _ = synthetic_blurry_batch(B, 3, 70, 90, seed0=21)
    eng.set_dense_eval("auto", 0)
    th = np.deg2rad(np.float32([30.0, 40.0, 50.0, 60.0]))
    buf = eng.make_kernels([2.0] * B, [1.0] * B, th, name="np.rewrite")
    first = eng.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    # one record in the middle of the set, rewritten in place as an axis-aligned (rank-1) Gaussian
    fp = C.POINTER(C.c_float)
    one = lambda v: np.ascontiguousarray([v], np.float32).ctypes.data_as(fp)
    eng._check(eng.lib.pb_make_kernels(eng.ctx, 1, one(2.5), one(1.0), one(0.0), capi.PB_SUPPORT_FULL,
                                       C.c_void_p(buf.ptr + 2 * capi.INFO_DTYPE.itemsize)))
    info = eng.read_info(buf, B)
    assert list(info["separable"]) == [0, 0, 1, 0]
    eng.buffer("np.out", x.nbytes).upload(np.full(x.shape, 7.0, np.float32))          # poison: an unwritten image would show
    out = eng.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    k = ref.gaussian_kernel_2d(np.float32([th[0], th[1], 0.0, th[3]]), [2.0, 2.0, 2.5, 2.0], [1.0] * B)
    want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method="fft")
    assert maxabs(out, want) < 1e-5
    assert maxabs(out[[0, 1, 3]], first[[0, 1, 3]]) == 0.0
