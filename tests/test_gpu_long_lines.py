"""Image sides beyond what the spectral derivative keeps in LDS (above 20480 samples, or above 8192 with a prime factor
> 7): the same transform through a line buffer in device memory (csrc/estimate.hip: grad_rows_long_kernel /
grad_cols_long_kernel).  The reference's torch.fft takes any length (filters.py:172-184); the checker is the oracle's
numpy transform.  Thin images keep the oracle fast."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd import _capi as capi
from polyblur_amd.synthetic import synthetic_blurry_batch

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)


@pytest.fixture(scope="module")
def eng():
    from polyblur_amd.engine import get_engine
    return get_engine(0)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.mark.parametrize("shape", [
    (1, 1, 24, 8200),        # rows: Bluestein, power-of-two core of 32768 points
    (1, 1, 8200, 24),        # columns: the same, 8-column tiles
    (1, 2, 21, 20736),       # rows: 2^8 3^4 > 20480, a direct plan; odd height = an unpaired last row
    (2, 1, 20736, 10),       # columns of the same length, a ragged last tile
    (1, 1, 5, 40001),        # rows: core of 131072 points
    (1, 1, 65536, 3),        # the longest line taken, columns; width below one tile
    (1, 1, 4, 65536),        # ... and rows
    (1, 1, 8191, 9001),      # one axis in LDS (8191 = 8191 prime -> Bluestein 16384), the other through memory
])
def test_fourier_gradients_long_lines(eng, shape):
    assert capi.load_library().pb_fft_length_supported(max(shape[2], shape[3])) == 2
    rng = np.random.default_rng(11)
    x = rng.random(shape, dtype=np.float32)
    gx, gy = eng.fourier_gradients(x)
    rx, ry = ref.spectral_gradients(x)
    scale = max(1.0, float(np.abs(rx).max()), float(np.abs(ry).max()))
    assert maxabs(gx, rx) < 6e-6 * scale and maxabs(gy, ry) < 6e-6 * scale, (maxabs(gx, rx), maxabs(gy, ry), scale)


@pytest.mark.parametrize("shape", [(1, 3, 72, 8200), (1, 3, 8200, 72), (2, 1, 40, 20736)])
@pytest.mark.parametrize("method", ["fft", "direct"])
def test_pipeline_on_long_images(shape, method):
    """the whole call: the estimation's fused maxima (column kernel, MODE 1) and every reblurring body on a long, thin image"""
    from polyblur_amd import polyblur_deblurring
    import torch
    x, _ = synthetic_blurry_batch(shape[0], shape[1], shape[2], shape[3], seed0=404)
    out, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), n_iter=2, method=method, return_info=True, **KW)
    want, winfos = ref.polyblur_deblurring(x, n_iter=2, method=method, return_info=True, **KW)
    for a, b in zip(infos, winfos):
        assert np.array_equal(a["theta"], b["theta"])
    assert maxabs(out.cpu().numpy(), want) < 3e-5


def test_halo_and_saturation_on_a_long_image():
    """remove_halo takes the row transform of the deblurred image (deblurring.py:174), discard_saturation masks the
    gradients in the column kernel's maxima (blur_estimation.py:117-118), q > 0 normalises on load"""
    from polyblur_amd import polyblur_deblurring
    import torch
    x, _ = synthetic_blurry_batch(1, 3, 48, 8200, seed0=405)
    x = np.clip(x * 1.3, 0, 1).astype(np.float32)
    kw = dict(n_iter=2, remove_halo=True, discard_saturation=True, q=1e-3, **KW)
    out, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, **kw)
    want, winfos = ref.polyblur_deblurring(x, return_info=True, **kw)
    for a, b in zip(infos, winfos):
        assert np.array_equal(a["theta"], b["theta"])
    assert maxabs(out.cpu().numpy(), want) < 3e-5


def test_lengths_beyond_the_limit_raise():
    from polyblur_amd import polyblur_deblurring
    with pytest.raises(ValueError):
        polyblur_deblurring(np.zeros((4, 65537, 1), np.float32))
