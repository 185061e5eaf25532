"""Deterministic synthetic blurry images for tests and bench.py.

Pure NumPy, no dependency on the engine or on the oracle.  Image ``i`` of a batch is
drawn from ``np.random.default_rng(seed0 + i)``:

* sharp image: 24 random plane-wave sinusoids (1..40 cycles/image, random phase,
  per-channel gain) + 32 random axis-aligned rectangles (alpha 0.5), min-max scaled
  to [0.05, 0.95] -- edges in every direction, so the directional-maximum estimator
  has something to measure;
* blur: sampled anisotropic Gaussian, sigma ~ U[0.6, 3.5], rho = sigma * U[0.33, 1]
  (>= 0.3), theta = 6 deg * randint(0, 30), 25x25, circular (FFT) convolution;
* noise N(0, 0.01^2), clip to [0, 1], float32.
"""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 20260929


def gaussian_psf(sigma: float, rho: float, theta_deg: float, ksize: int = 25) -> np.ndarray:
    """Sum-normalised sampled Gaussian with std ``sigma`` along direction ``theta`` and
    ``rho`` across it (float64 maths, float32 result)."""
    t = np.arange(ksize, dtype=np.float64) - (ksize - 1) // 2
    X, Y = np.meshgrid(t, t, indexing="xy")
    th = -np.deg2rad(theta_deg)
    u = np.cos(th) * X + np.sin(th) * Y
    v = -np.sin(th) * X + np.cos(th) * Y
    k = np.exp(-0.5 * ((u / sigma) ** 2 + (v / rho) ** 2))
    return (k / k.sum()).astype(np.float32)


def synthetic_image(c: int, h: int, w: int, rng: np.random.Generator) -> np.ndarray:
    """One sharp (C,H,W) float32 image in [0.05, 0.95]."""
    yy = (np.arange(h, dtype=np.float32) / np.float32(h))[:, None]
    xx = (np.arange(w, dtype=np.float32) / np.float32(w))[None, :]
    base = np.zeros((h, w), np.float32)
    img = np.zeros((c, h, w), np.float32)
    for _ in range(24):
        f = rng.uniform(1.0, 40.0)
        ang = rng.uniform(0.0, np.pi)
        amp = rng.uniform(0.0, 1.0) / 24.0
        ph = rng.uniform(0.0, 2 * np.pi)
        gains = rng.uniform(0.6, 1.0, size=c).astype(np.float32)
        wave = np.sin(np.float32(2 * np.pi * f) * (np.float32(np.cos(ang)) * xx + np.float32(np.sin(ang)) * yy)
                      + np.float32(ph)).astype(np.float32)
        img += (np.float32(amp) * gains)[:, None, None] * wave[None]
    del base
    for _ in range(32):
        y0, y1 = np.sort(rng.integers(0, h, size=2))
        x0, x1 = np.sort(rng.integers(0, w, size=2))
        val = rng.uniform(0.0, 1.0, size=c).astype(np.float32)
        y1, x1 = max(y1, y0 + 1), max(x1, x0 + 1)
        img[:, y0:y1, x0:x1] = 0.5 * img[:, y0:y1, x0:x1] + 0.5 * val[:, None, None]
    lo, hi = img.min(), img.max()
    img = 0.05 + 0.9 * (img - lo) / max(hi - lo, 1e-12)
    return img.astype(np.float32)


def blur_circular(img: np.ndarray, psf: np.ndarray) -> np.ndarray:
    """Circular convolution of every channel of (C,H,W) with a centred psf."""
    h, w = img.shape[-2:]
    kh, kw = psf.shape
    big = np.zeros((h, w), np.float32)
    iy = (np.arange(kh) - kh // 2) % h                 # centred psf folded onto the torus (images may be
    ix = (np.arange(kw) - kw // 2) % w                 # smaller than the psf)
    np.add.at(big, (iy[:, None], ix[None, :]), psf)
    out = np.fft.irfft2(np.fft.rfft2(img) * np.fft.rfft2(big), s=(h, w))
    return out.astype(np.float32)


def synthetic_blurry_image(c: int, h: int, w: int, seed: int, noise_std: float = 0.01,
                           force_theta_deg=None, blur=None):
    """Returns (blurry (C,H,W) float32, (sigma, rho, theta_deg)).  blur = (sigma, rho, theta_deg) instead of the drawn one."""
    rng = np.random.default_rng(seed)
    sharp = synthetic_image(c, h, w, rng)
    sigma = rng.uniform(0.6, 3.5)
    rho = max(sigma * rng.uniform(0.33, 1.0), 0.3)
    theta = 6.0 * int(rng.integers(0, 30))
    if force_theta_deg is not None:
        theta = float(force_theta_deg)
    if blur is not None:
        sigma, rho, theta = (float(v) for v in blur)
    blurry = blur_circular(sharp, gaussian_psf(sigma, rho, theta))
    blurry = blurry + rng.normal(0.0, noise_std, size=blurry.shape).astype(np.float32)
    return np.clip(blurry, 0.0, 1.0).astype(np.float32), (float(sigma), float(rho), float(theta))


def synthetic_blurry_batch(b: int, c: int, h: int, w: int, seed0: int = DEFAULT_SEED,
                           noise_std: float = 0.01, force_theta_deg=None):
    """Returns ((B,C,H,W) float32, list of true (sigma, rho, theta_deg))."""
    imgs, params = [], []
    for i in range(b):
        im, p = synthetic_blurry_image(c, h, w, seed0 + i, noise_std, force_theta_deg)
        imgs.append(im)
        params.append(p)
    return np.stack(imgs).astype(np.float32), params
