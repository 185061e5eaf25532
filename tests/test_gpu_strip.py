"""GPU parity of the streaming strip body for rank-1 kernels (csrc/conv_strip.hip; opt-in, PB_STRIP=1: measured slower
than the tile body it was meant to replace -- NOTEBOOK.md section 4 -- and kept as the measured experiment behind --experimental).  It must agree
with the oracle and with the tile body under both boundary models, on border and interior strips, in mixed batches."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("PB_TEST_EXPERIMENTAL") != "1",
                                 reason="conv_strip.hip is not in the default library: build with `python -m polyblur_amd.build "
                                        "--experimental` and set PB_TEST_EXPERIMENTAL=1")]

from oracle import polyblur_ref as ref                      # the checker (tests only)
from polyblur_amd import _capi as capi
from polyblur_amd.synthetic import synthetic_blurry_batch


@pytest.fixture(scope="module")
def engines():
    from polyblur_amd.engine import Engine
    old = os.environ.get("PB_STRIP")
    os.environ["PB_STRIP"] = "1"
    try:
        strip = Engine(0)
    finally:
        if old is None:
            del os.environ["PB_STRIP"]
        else:
            os.environ["PB_STRIP"] = old
    tile = Engine(0)
    yield strip, tile
    strip.close()
    tile.close()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.mark.parametrize("shape", [(1, 3, 150, 210), (1, 1, 300, 800), (2, 3, 97, 520)])
@pytest.mark.parametrize("boundary,method", [(capi.PB_WRAP, "fft"), (capi.PB_ZERO, "direct")])
def test_strip_body_matches_oracle_and_tile_body(engines, shape, boundary, method):
    strip, tile = engines
    B = shape[0]
    x, _ = synthetic_blurry_batch(*shape, seed0=71)
    sg, rh, th = [3.0, 2.2][:B], [1.5, 2.2][:B], [np.float32(0.0), np.float32(0.7)][:B]        # axis-aligned; isotropic
    k = ref.gaussian_kernel_2d(th, sg, rh)
    want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method=method)
    outs = []
    for eng in (strip, tile):
        buf = eng.make_kernels(sg, rh, th)
        info = eng.read_info(buf, B)
        assert all(info["separable"] == 1) and all(info["radius"] > 8)
        outs.append(eng.inverse_filter(x, buf, 6.0, 1.0, boundary))
    assert maxabs(outs[0], want) < 1e-5
    assert maxabs(outs[0], outs[1]) < 5e-6


def test_strip_body_in_a_mixed_batch(engines):
    """one image per body: rank-1 with full support (strip), rank-1 with a small kernel (tile), dense (tile-spectrum)"""
    strip, _ = engines
    x, _ = synthetic_blurry_batch(3, 3, 120, 300, seed0=72)
    sg, rh, th = [3.0, 0.6, 2.0], [1.5, 0.6, 1.0], [np.float32(0.0), np.float32(0.0), np.deg2rad(np.float32(30.0))]
    k = ref.gaussian_kernel_2d(th, sg, rh)
    buf = strip.make_kernels(sg, rh, th, support=capi.PB_SUPPORT_ADAPTIVE)
    info = strip.read_info(buf, 3)
    assert list(info["separable"]) == [1, 1, 0] and info["radius"][0] > 8 and info["radius"][1] <= 8
    out = strip.inverse_filter(x, buf, 6.0, 1.0, capi.PB_WRAP)
    want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method="fft")
    assert maxabs(out, want) < 2e-5
