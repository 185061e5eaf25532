"""CPU oracle for the Polyblur hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch NumPy (fp32) restatement of the algorithm in the reference
``teboli/polyblur`` package, written from the math in SURVEY.md section 8a.  Every
function cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product (``polyblur_amd``) never does.

Pinning: the reference ships no tests or golden vectors for this path (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by ``tests/golden/make_golden.py`` (imports
``/root/reference``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every stage and the full pipeline against
them (max-abs <= 2e-6 for stages, see the test for the per-case bounds).

Layout convention: images are ``(B, C, H, W)`` float32 arrays ("planar"), like the
reference's torch tensors.  The public entry point also accepts ``(H, W)`` /
``(H, W, C)`` arrays, like the reference does for ndarrays.

Deliberate deviations from the reference (all documented in SURVEY.md section 2.2):
  * edgetaper's alpha uses a per-image maximum (reference: max over the whole
    batch, ``edgetaper.py:15,21``) -- identical for B == 1;
  * ``method='direct'`` works for B > 1 (reference: shape error, ``filters.py:45-49``);
    it is the loop of B == 1 calls;
  * halo masking reproduces the reference's ``grad_y * grad_y`` product
    (``deblurring.py:174``) bit-for-bit in intent ("bug-compatible").
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


class _Fft:
    """The transforms the oracle uses.  Default: numpy.fft (what the goldens were pinned with).
    ``set_fft_workers(n)`` switches to scipy.fft's multi-threaded pocketfft -- same transforms, rounding-level
    differences -- for the all-cores leg of bench.py's cpu_baseline; ``set_fft_workers(None)`` switches back."""
    workers = None

    def __getattr__(self, name):
        if self.workers is None or name in ("fftshift", "ifftshift"):
            return getattr(np.fft, name)
        import functools
        import scipy.fft
        return functools.partial(getattr(scipy.fft, name), workers=self.workers)


_fft = _Fft()


def set_fft_workers(n):
    _Fft.workers = None if n is None else int(n)


__all__ = [
    "polyblur_deblurring", "PolyblurDeblurring", "spectral_gradients",
    "spectral_gradients_1d", "estimate_gaussian_blur", "gaussian_kernel_2d",
    "polynomial_deconvolution", "inverse_filtering_rank3", "edgetaper",
    "edgetaper_weights", "halo_masking", "bilateral_filter",
    "recursive_filter", "correlate_same_zero", "circular_convolve",
]


# ----------------------------------------------------------------------------
# layout helpers                                   (utils.py:8-31, utils.py:48-61)
# ----------------------------------------------------------------------------

def to_planar(img: np.ndarray) -> np.ndarray:
    """(H,W) or (H,W,C) ndarray -> (1,C,H,W) float32 copy.  utils.py:8-21 +
    deblurring.py:48.  NB: uint8 input is cast, not rescaled (utils.py:15-16)."""
    a = np.asarray(img)
    if a.ndim == 2:
        p = a[None, None]
    elif a.ndim == 3:
        p = np.moveaxis(a, 2, 0)[None]
    else:
        raise ValueError("expected (H,W) or (H,W,C) array, got shape %r" % (a.shape,))
    return np.ascontiguousarray(p, dtype=F32)


def from_planar(x: np.ndarray) -> np.ndarray:
    """(1,C,H,W) -> (H,W) or (H,W,C).  utils.py:24-31 (squeeze, then CHW->HWC)."""
    s = np.squeeze(x)
    if s.ndim == 2:
        return np.ascontiguousarray(s)
    return np.ascontiguousarray(np.moveaxis(s, 0, -1))


def replicate_pad(x: np.ndarray, r: int) -> np.ndarray:
    """Edge-replicate pad of the two last axes by r.  utils.py:48-53."""
    pad = [(0, 0)] * (x.ndim - 2) + [(r, r), (r, r)]
    return np.pad(x, pad, mode="edge")


def crop(x: np.ndarray, r: int) -> np.ndarray:
    """Inverse of the pad.  utils.py:56-61."""
    return x[..., r:-r, r:-r] if r > 0 else x


# ----------------------------------------------------------------------------
# spectral gradients                                         (filters.py:159-186)
# ----------------------------------------------------------------------------

def _centered_freqs(n: int) -> np.ndarray:
    # filters.py:175-176: (arange(n) - n//2) / n, in the fftshift-ed ordering.
    return ((np.arange(n) - n // 2) / n).astype(F32)


def spectral_gradients(x: np.ndarray):
    """Fourier-interpolated image gradient, the reference's 2-D formulation.

    filters.py:172-184: U = fftshift(fft2(x)); gx = Re ifft2(ifftshift(2 pi f_w (i U)));
    gy likewise with f_h.  Returns (gx, gy), each the shape of x, float32.
    """
    x = np.asarray(x, dtype=F32)
    h, w = x.shape[-2:]
    U = _fft.fftshift(_fft.fft2(x), axes=(-2, -1))
    iU = (-U.imag + 1j * U.real).astype(np.complex64)          # = i * U
    fw = _centered_freqs(w).reshape((1,) * (x.ndim - 1) + (w,))
    fh = _centered_freqs(h).reshape((1,) * (x.ndim - 2) + (h, 1))
    two_pi = F32(2.0 * np.pi)
    gx = _fft.ifft2(_fft.ifftshift((two_pi * fw) * iU, axes=(-2, -1))).real
    gy = _fft.ifft2(_fft.ifftshift((two_pi * fh) * iU, axes=(-2, -1))).real
    return gx.astype(F32), gy.astype(F32)


def spectral_derivative_multiplier(n: int) -> np.ndarray:
    """The length-n multiplier D[k] = 2 pi i f_k / 1 in *unshifted* FFT order with the
    Nyquist bin (even n) zeroed: the Nyquist term is purely imaginary after the inverse
    transform of a real signal and is dropped by ``real()`` in filters.py:180,183."""
    k = np.arange(n)
    f = np.where(k <= (n - 1) // 2, k, k - n).astype(np.float64) / n
    if n % 2 == 0:
        f[n // 2] = 0.0
    return (2j * np.pi * f).astype(np.complex64)


def spectral_gradients_1d(x: np.ndarray):
    """Separable restatement of :func:`spectral_gradients`: gx is the
    row-wise, gy the column-wise 1-D periodic spectral derivative.  This is the form
    the HIP kernels implement; tests check it equals the 2-D form to rounding."""
    x = np.asarray(x, dtype=F32)
    h, w = x.shape[-2:]
    dw = spectral_derivative_multiplier(w)
    dh = spectral_derivative_multiplier(h)
    gx = _fft.ifft(_fft.fft(x, axis=-1) * dw, axis=-1).real
    gy = _fft.ifft(_fft.fft(x, axis=-2) * dh[:, None], axis=-2).real
    return gx.astype(F32), gy.astype(F32)


# ----------------------------------------------------------------------------
# blur estimation                                          (blur_estimation.py)
# ----------------------------------------------------------------------------

def saturation_mask(gray: np.ndarray, discard: bool, threshold: float = 0.99):
    """blur_estimation.py:83-88."""
    if discard:
        return gray > F32(threshold)
    return np.zeros(gray.shape, dtype=bool)


def range_normalize(gray: np.ndarray, q: float = 0.0):
    """(x - lo) / (hi - lo) clamped to [0,1]; lo/hi are per-(image,channel) min/max or
    the q / 1-q quantiles.  blur_estimation.py:92-109.  Returns (normalised, lo, hi)."""
    b, c = gray.shape[:2]
    flat = gray.reshape(b, c, -1)
    if q > 0:
        lo = np.quantile(flat, q, axis=-1, method="linear").astype(F32)
        hi = np.quantile(flat, 1.0 - q, axis=-1, method="linear").astype(F32)
    else:
        lo = flat.min(axis=-1)
        hi = flat.max(axis=-1)
    lo = lo.reshape(b, c, 1, 1).astype(F32)
    hi = hi.reshape(b, c, 1, 1).astype(F32)
    out = np.clip((gray - lo) / (hi - lo), F32(0), F32(1)).astype(F32)
    return out, lo, hi


def directional_maxima(gx: np.ndarray, gy: np.ndarray, n_angles: int = 6) -> np.ndarray:
    """m_k = max over (C,H,W) of |cos(t_k) gx - sin(t_k) gy|, t_k = k pi / n_angles,
    k = 0..n_angles.  blur_estimation.py:122-134 (the channel mean at :127-128 is over
    a size-1 axis for the gray image).  Returns (B, n_angles+1) float32."""
    gxm = gx.mean(axis=1, dtype=F32)
    gym = gy.mean(axis=1, dtype=F32)
    ang = np.linspace(0.0, np.pi, n_angles + 1).astype(F32)
    out = np.empty((gx.shape[0], n_angles + 1), dtype=F32)
    for k, t in enumerate(ang):
        ck, sk = F32(np.cos(t, dtype=F32)), F32(np.sin(t, dtype=F32))
        out[:, k] = np.abs(ck * gxm - sk * gym).reshape(gx.shape[0], -1).max(axis=1)
    return out


def keys_cubic_weights(x_new: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Row-normalised Keys (a = -0.5) cubic weights on |x_new - x|, no edge handling,
    row sums regularised by +1e-5.  blur_estimation.py:138-147.  Returns (Nnew, Nold)."""
    d = np.abs(x_new[:, None].astype(F32) - x[None, :].astype(F32)).astype(F32)
    near = d < 1
    far = (d >= 1) & (d < 2)
    wf = ((F32(-0.5) * d + F32(2.5)) * d - F32(4)) * d + F32(2)
    wn = (F32(1.5) * d - F32(2.5)) * d * d + F32(1)
    w = far.astype(F32) * wf + near.astype(F32) * wn
    w = w / (w.sum(axis=-1, keepdims=True, dtype=F32) + F32(1e-5))
    return w.astype(F32)


def angle_grids(n_angles: int = 6, n_interpolated_angles: int = 30):
    """deblurring.py:62-63: both grids are truncated to integers (``.long()``)."""
    thetas = np.linspace(0, 180, n_angles + 1).astype(np.int64)
    step = 180 / n_interpolated_angles
    interp = np.arange(0, 180, step).astype(np.int64)
    return thetas, interp


def dominant_direction(mags: np.ndarray, thetas: np.ndarray, interp_thetas: np.ndarray):
    """blur_estimation.py:151-167.  Returns (m_normal, m_ortho, theta_rad, interp, i_min)
    with shapes (B,), (B,), (B,), (B,N), (B,)."""
    n_int = interp_thetas.shape[-1]
    # :156-157 -- both grids are divided by N (=30), NOT by the angular step
    w = keys_cubic_weights(interp_thetas.astype(F32) / F32(n_int), thetas.astype(F32) / F32(n_int))
    interp = (mags.astype(F32) @ w.T).astype(F32)                        # (B,N)
    i_min = np.argmin(interp, axis=-1)
    theta_deg = interp_thetas[i_min]                                      # integer degrees
    m_normal = np.take_along_axis(interp, i_min[:, None], axis=-1)[:, 0]
    ortho_deg = (theta_deg + 90) % 180
    i_ortho = (ortho_deg / (180 / n_int)).astype(np.int64)                # :165 truncation
    m_ortho = np.take_along_axis(interp, i_ortho[:, None], axis=-1)[:, 0]
    theta = theta_deg.astype(F32) * F32(np.pi) / F32(180)
    return m_normal, m_ortho, theta.astype(F32), interp, i_min


def gaussian_std_from_magnitudes(m_normal, m_ortho, c: float, b: float):
    """sigma = sqrt(clamp(c^2/(m^2+1e-8) - b^2, 0.09, 16)).  blur_estimation.py:171-185."""
    cc, bb = F32(c * c), F32(b * b)

    def one(m):
        m = m.astype(F32)
        v = cc / (m * m + F32(1e-8)) - bb
        return np.sqrt(np.clip(v, F32(0.09), F32(16.0))).astype(F32)

    return one(m_normal), one(m_ortho)


def gaussian_kernel_2d(theta, sigma, rho, ksize: int = 25) -> np.ndarray:
    """Sampled, rotated, sum-normalised Gaussian.  blur_estimation.py:189-232.
    theta (radians), sigma, rho: (B,) arrays.  Returns (B, ksize, ksize) float32 with
    k[i, j] = exp(-(A X^2 + 2 B X Y + C Y^2)/2), X = t[j], Y = t[i]."""
    theta = -np.asarray(theta, dtype=F32)                                 # :194
    sigma = np.asarray(sigma, dtype=F32)
    rho = np.asarray(rho, dtype=F32)
    co, si = np.cos(theta, dtype=F32), np.sin(theta, dtype=F32)
    i1 = F32(1) / (sigma * sigma)
    i2 = F32(1) / (rho * rho)
    a00 = co * co * i1 + si * si * i2                                     # :204-207
    a01 = si * co * (i1 - i2)
    a11 = co * co * i2 + si * si * i1
    t = (np.arange(ksize) - (ksize - 1) // 2).astype(F32)                 # :222
    X = t[None, None, :]
    Y = t[None, :, None]
    quad = (a00[:, None, None] * X * X + F32(2) * a01[:, None, None] * X * Y
            + a11[:, None, None] * Y * Y)
    k = np.exp(F32(-0.5) * quad, dtype=F32)
    return (k / k.sum(axis=(-2, -1), keepdims=True, dtype=F32)).astype(F32)


def separable_xt_kernels(theta, sigma, rho, ksize: int = 25):
    """The x-t separable APPROXIMATION of gaussian_kernel_2d -- method='direct_separable'.

    Intent of the reference's dead side-car separable_gaussian2d.cpp:91-183 (a 1-D Gaussian along x followed by a
    1-D Gaussian along an oblique line, sampled with linear interpolation -- Geusebroek et al.), whose own
    formulas do not run (`tan_phi` at :103 drops the cos*sin factor; the Python stub filters.py:96-98 returns
    its input).  The decomposition is derived from the quadratic form the live kernel uses, so that it
    approximates the SAME Gaussian: with q = A X^2 + 2 B X Y + C Y^2 (A, B, C as in blur_estimation.py:204-207),
        q = A (X + (B/A) Y)^2 + (C - B^2/A) Y^2
    = a Gaussian in X of variance 1/A, then one in Y of variance A/(AC - B^2) taken along the line
    X = -(B/A) Y.  The axis with the larger coefficient (the narrower Gaussian) goes first -- X if A >= C, else the
    roles of X and Y are swapped -- which keeps the line within 45 degrees of the second axis (|shear| <= 1).
    Every 1-D kernel is sampled on -r..r and normalised to sum 1; the oblique one puts, for each offset i along
    its axis, the weights g[i] (1-f) and g[i] f on the two samples next to the line (f = fractional part).

    Returns (K1, K2), two (B, ksize, ksize) float32 correlation kernels; K ~= K2 applied after K1.
    PARITY UNPINNED: there is no running reference for this path; the GPU engine is tested against THIS
    restatement, and its distance to the exact kernel is what the tests state as the method's tolerance."""
    theta = -np.asarray(theta, dtype=F32)
    sigma = np.asarray(sigma, dtype=F32)
    rho = np.asarray(rho, dtype=F32)
    co, si = np.cos(theta, dtype=F32), np.sin(theta, dtype=F32)
    i1 = F32(1) / (sigma * sigma)
    i2 = F32(1) / (rho * rho)
    a00 = co * co * i1 + si * si * i2
    a01 = si * co * (i1 - i2)
    a11 = co * co * i2 + si * si * i1
    r = (ksize - 1) // 2
    t = np.arange(-r, r + 1).astype(F32)
    B = theta.shape[0]
    K1 = np.zeros((B, ksize, ksize), F32)
    K2 = np.zeros((B, ksize, ksize), F32)
    for n in range(B):
        A, Bq, C = a00[n], a01[n], a11[n]
        x_first = bool(A >= C)                     # the narrower axis first: |B| <= sqrt(AC) <= max(A, C), so |shear| <= 1
        p, o = (A, C) if x_first else (C, A)       # first axis coefficient, other axis coefficient
        det = p * o - Bq * Bq
        g1 = np.exp(F32(-0.5) * p * t * t, dtype=F32)
        g1 = (g1 / g1.sum(dtype=F32)).astype(F32)
        g2 = np.exp(F32(-0.5) * (det / p) * t * t, dtype=F32)
        g2 = (g2 / g2.sum(dtype=F32)).astype(F32)
        shear = F32(-Bq / p)
        for i in range(-r, r + 1):
            if x_first:
                K1[n, r, r + i] = g1[r + i]
            else:
                K1[n, r + i, r] = g1[r + i]
            pos = F32(shear * F32(i))
            m = int(np.floor(pos))
            f = F32(pos - F32(m))
            for mm, wgt in ((m, F32(g2[r + i] * (F32(1) - f))), (m + 1, F32(g2[r + i] * f))):
                if abs(mm) <= r and wgt != 0:
                    if x_first:
                        K2[n, r + i, r + mm] += wgt
                    else:
                        K2[n, r + mm, r + i] += wgt
    return K1, K2


def estimate_gaussian_blur(img: np.ndarray, c: float, b: float, q: float = 0.0,
                           n_angles: int = 6, n_interpolated_angles: int = 30,
                           ker_size: int = 25, discard_saturation: bool = False,
                           multichannel: bool = False, return_info: bool = False):
    """blur_estimation.py:18-79 for the cases that run in the reference: RGB (always
    gray, :36-37) or single-channel input.  Returns kernel (B,1,k,k) [, info dict]."""
    x = np.asarray(img, dtype=F32)
    if x.shape[1] == 3 or not multichannel:
        gray = x.mean(axis=1, keepdims=True, dtype=F32)
    elif x.shape[1] == 1:
        gray = x
    else:
        raise NotImplementedError("per-channel kernels crash in the reference for C not in {1,3} "
                                  "(blur_estimation.py:67 overwrites `thetas`)")
    thetas, interp_thetas = angle_grids(n_angles, n_interpolated_angles)
    mask = saturation_mask(gray, discard_saturation)
    norm, lo, hi = range_normalize(gray, q)
    gx, gy = spectral_gradients(norm)
    gx[mask] = 0                                                           # :117-118
    gy[mask] = 0
    mags = directional_maxima(gx, gy, n_angles)
    m_n, m_o, theta, interp, i_min = dominant_direction(mags, thetas, interp_thetas)
    sigma, rho = gaussian_std_from_magnitudes(m_n, m_o, c, b)
    kernel = gaussian_kernel_2d(theta, sigma, rho, ker_size)[:, None]
    if return_info:
        info = dict(mags=mags, interp=interp, i_min=i_min, theta=theta, sigma=sigma,
                    rho=rho, lo=lo.reshape(-1), hi=hi.reshape(-1))
        return kernel, info
    return kernel


# ----------------------------------------------------------------------------
# convolution primitives                                        (filters.py:14-49)
# ----------------------------------------------------------------------------

def _per_image_kernel(kernel: np.ndarray, b: int):
    k = np.asarray(kernel, dtype=F32)
    if k.ndim == 2:
        k = np.broadcast_to(k, (b, 1) + k.shape)
    elif k.ndim == 3:
        k = k[:, None]
    if k.shape[1] != 1:
        raise NotImplementedError("oracle supports one kernel per image (B,1,h,w)")
    return k


def correlate_same_zero(x: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """Cross-correlation with zero 'same' padding, one kernel per image applied to every
    channel.  filters.py:40-49 (F.conv2d is a correlation; B == 1 semantics looped)."""
    x = np.asarray(x, dtype=F32)
    b, ch, h, w = x.shape
    k = _per_image_kernel(kernel, b)
    kh, kw = k.shape[-2:]
    # padding='same': kh - 1 rows in all, (kh - 1) // 2 of them in front -- for an even kernel one fewer than behind
    ty, tx = (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, [(0, 0), (0, 0), (ty, kh - 1 - ty), (tx, kw - 1 - tx)])
    out = np.zeros_like(x)
    for i in range(kh):
        for j in range(kw):
            tap = k[:, 0, i, j].reshape(b, 1, 1, 1)
            if not np.any(tap):
                continue
            out += tap * xp[:, :, i:i + h, j:j + w]
    return out


def psf_to_otf(kernel: np.ndarray, shape) -> np.ndarray:
    """Zero-embed the psf at the top-left, roll its centre to (0,0), FFT.
    filters.py:255-273."""
    k = np.asarray(kernel, dtype=F32)
    kh, kw = k.shape[-2:]
    big = np.zeros(k.shape[:-2] + tuple(shape), dtype=F32)
    big[..., :kh, :kw] = k
    big = np.roll(big, (-(kh // 2), -(kw // 2)), axis=(-2, -1))
    return _fft.fft2(big).astype(np.complex64)


def circular_convolve(x: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """``convolve2d(..., method='fft')``: circular-pad by the kernel radius, multiply in
    Fourier, crop (filters.py:31-35).  Equals a circular convolution on the un-padded
    domain whenever the image is larger than the kernel."""
    x = np.asarray(x, dtype=F32)
    k = _per_image_kernel(kernel, x.shape[0])
    r = k.shape[-1] // 2
    xp = np.pad(x, [(0, 0), (0, 0), (r, r), (r, r)], mode="wrap")
    X = _fft.fft2(xp).astype(np.complex64)
    K = psf_to_otf(k, xp.shape[-2:])
    return crop(_fft.ifft2(K * X).real.astype(F32), r)


def convolve2d(x, kernel, method="direct"):
    """filters.py:14-37 dispatcher (2-D kernels only)."""
    if method == "direct":
        return correlate_same_zero(x, kernel)
    if method == "fft":
        return circular_convolve(x, kernel)
    raise ValueError("Convolution method %s is not implemented" % method)


# ----------------------------------------------------------------------------
# polynomial deconvolution                                 (deblurring.py:113-169)
# ----------------------------------------------------------------------------

def polynomial_coefficients(alpha: float, beta: float):
    """deblurring.py:133-135 / :162-164 -> (a3, a2, a1, beta)."""
    return alpha / 2 - beta + 2, 3 * beta - alpha - 6, 5 - 3 * beta + alpha / 2, beta


def polynomial_deconvolution(x: np.ndarray, kernel: np.ndarray, alpha: float, beta: float,
                             method: str = "fft") -> np.ndarray:
    """y = a3 K^3 x + a2 K^2 x + a1 K x + beta x by Horner.
    'fft' : deblurring.py:141-169 (circular over the given domain);
    'direct' : deblurring.py:122-138 (three zero-padded correlations)."""
    x = np.asarray(x, dtype=F32)
    a3, a2, a1, b0 = polynomial_coefficients(alpha, beta)
    if method == "fft":
        Y = _fft.fft2(x).astype(np.complex64)
        K = psf_to_otf(_per_image_kernel(kernel, x.shape[0]), x.shape[-2:])
        X = F32(a3) * Y
        X = K * X + F32(a2) * Y
        X = K * X + F32(a1) * Y
        X = K * X + F32(b0) * Y
        return _fft.ifft2(X).real.astype(F32)
    if method == "direct":
        t = F32(a3) * x
        t = correlate_same_zero(t, kernel) + F32(a2) * x
        t = correlate_same_zero(t, kernel) + F32(a1) * x
        return correlate_same_zero(t, kernel) + F32(b0) * x
    if method == "direct_separable":               # kernel = (K1, K2) of separable_xt_kernels: K ~= K2 after K1
        k1, k2 = kernel
        blur = lambda v: correlate_same_zero(correlate_same_zero(v, k1), k2)
        t = F32(a3) * x
        t = blur(t) + F32(a2) * x
        t = blur(t) + F32(a1) * x
        return blur(t) + F32(b0) * x
    raise ValueError("%s not implemented" % method)


# ----------------------------------------------------------------------------
# edgetaper                                                      (edgetaper.py)
# ----------------------------------------------------------------------------

def _autocorr_weight(proj: np.ndarray, n: int) -> np.ndarray:
    # edgetaper.py:11-15: z = ifft(|fft(p, n-1)|^2), append z[0], 1 - z/max(z)
    z = _fft.fft(proj.astype(F32), n - 1, axis=-1)
    z = _fft.ifft(np.abs(z) ** 2, axis=-1).real.astype(F32)
    z = np.concatenate([z, z[..., :1]], axis=-1)
    return (F32(1) - z / z.max(axis=-1, keepdims=True)).astype(F32)   # per-image max


def edgetaper_weights(kernel: np.ndarray, shape) -> np.ndarray:
    """alpha = v1 (x) v2 with v the normalised circular autocorrelation of the kernel's
    axis projections.  edgetaper.py:10-23.  kernel (B,1,k,k) -> alpha (B,1,H,W)."""
    k = np.asarray(kernel, dtype=F32)
    v1 = _autocorr_weight(k.sum(axis=-1, dtype=F32), shape[0])     # project over columns -> rows
    v2 = _autocorr_weight(k.sum(axis=-2, dtype=F32), shape[1])
    return (v1[..., :, None] * v2[..., None, :]).astype(F32)


def edgetaper(x: np.ndarray, kernel: np.ndarray, n_tapers: int = 3, method: str = "fft"):
    """x <- alpha x + (1-alpha) (K*x), three times.  edgetaper.py:26-33."""
    x = np.asarray(x, dtype=F32)
    alpha = edgetaper_weights(_per_image_kernel(kernel, x.shape[0]), x.shape[-2:])
    for _ in range(n_tapers):
        blurred = convolve2d(x, kernel, method=method)
        x = alpha * x + (F32(1) - alpha) * blurred
    return x.astype(F32)


# ----------------------------------------------------------------------------
# halo masking                                             (deblurring.py:172-208)
# ----------------------------------------------------------------------------

def halo_masking(x: np.ndarray, y: np.ndarray, grad_x=None) -> np.ndarray:
    """deblurring.py:193-208 with helpers :173-190.  M = -gx*ox - gy*gy (sic, :174);
    nM = sum_{H,W}(gx^2+gy^2); z = clamp(M/(nM+M), min=0); out = y + z (x - y)."""
    if grad_x is None:
        gx, gy = spectral_gradients(x)
    else:
        gx, gy = grad_x
    ox, _oy = spectral_gradients(y)
    M = (-gx * ox) + (-gy * gy)
    nM = (gx * gx + gy * gy).sum(axis=(-2, -1), keepdims=True, dtype=F32)
    z = np.maximum(M / (nM + M), F32(0))
    return (y + z * (x - y)).astype(F32)


# ----------------------------------------------------------------------------
# edge-aware filters              (filters.py:107-148, domain_transform.py:6-85)
# ----------------------------------------------------------------------------

def bilateral_filter(x: np.ndarray, ksize: int = 5, sigma_spatial: float = 5.0,
                     sigma_color: float = 0.1) -> np.ndarray:
    """5x5 per-channel bilateral, replicate pad, J/(W+1e-5).  filters.py:107-148."""
    x = np.asarray(x, dtype=F32)
    r = ksize // 2
    h, w = x.shape[-2:]
    xp = replicate_pad(x, r)
    t = np.arange(-(ksize // 2), ksize // 2 + 1).astype(F32)      # filters.py:109
    var2 = F32(2 * sigma_color * sigma_color)
    num = np.zeros_like(x)
    den = np.zeros_like(x)
    for i in range(ksize):
        for j in range(ksize):
            gw = np.exp(-(t[j] * t[j] + t[i] * t[i]) / F32(2 * sigma_spatial * sigma_spatial), dtype=F32)
            s = xp[..., i:i + h, j:j + w]
            d = s - x
            wgt = np.exp(-d * d / var2, dtype=F32) * gw
            num += wgt * s
            den += wgt
    return (num / (den + F32(1e-5))).astype(F32)


def recursive_filter(I: np.ndarray, sigma_s: float = 60, sigma_r: float = 0.4,
                     num_iterations: int = 3, joint_image=None) -> np.ndarray:
    """Domain-transform recursive filter (Gastal & Oliveira RF).
    domain_transform.py:6-63; scan body :78-83."""
    I = np.asarray(I, dtype=F32)
    J = I if joint_image is None else np.asarray(joint_image, dtype=F32)
    dx = np.abs(np.diff(J, axis=-1)).sum(axis=1, dtype=F32)              # (B,H,W-1)  :27,31
    dy = np.abs(np.diff(J, axis=-2)).sum(axis=1, dtype=F32)              # (B,H-1,W)  :28,33
    dx = np.pad(dx, [(0, 0), (0, 0), (1, 0)])                             # :32
    dy = np.pad(dy, [(0, 0), (1, 0), (0, 0)])                             # :34
    ratio = F32(sigma_s / sigma_r)
    dHdx = (F32(1) + ratio * dx).astype(F32)                              # :37
    dVdy = (F32(1) + ratio * dy).astype(F32)                              # :38
    N = num_iterations
    out = I.copy()
    for i in range(N):
        sigma_i = sigma_s * math.sqrt(3) * 2 ** (N - (i + 1)) / math.sqrt(4 ** N - 1)   # :50
        a = F32(math.exp(-math.sqrt(2) / sigma_i))                                       # :53
        Vx = np.power(a, dHdx, dtype=F32)[:, None]                       # (B,1,H,W)
        w = out.shape[-1]
        for c in range(1, w):                                              # :78-79
            out[..., c] += Vx[..., c] * (out[..., c - 1] - out[..., c])
        for c in range(w - 2, -1, -1):                                     # :82-83
            out[..., c] += Vx[..., c + 1] * (out[..., c + 1] - out[..., c])
        Vy = np.power(a, dVdy, dtype=F32)[:, None]
        h = out.shape[-2]
        for r in range(1, h):
            out[..., r, :] += Vy[..., r, :] * (out[..., r - 1, :] - out[..., r, :])
        for r in range(h - 2, -1, -1):
            out[..., r, :] += Vy[..., r + 1, :] * (out[..., r + 1, :] - out[..., r, :])
    return out


def _first_greater(pos_ext: np.ndarray, thr: np.ndarray) -> np.ndarray:
    """First index j with pos_ext[j] > thr[k], per k (pos_ext ascending, last entry is the sentinel)."""
    return np.searchsorted(pos_ext, thr, side="right")


def _nc_box_rows(F: np.ndarray, ct: np.ndarray, box_radius) -> np.ndarray:
    """One horizontal normalized-convolution box pass.  NC.cpp:50-139 with its per-image (B=1)
    semantics applied to every image of the batch (the reference's `find` NC.cpp:10-47 returns the
    wrong index for batch > 1); any channel count (the reference hard-codes 3, NC.cpp:128-130).
    F: (B,C,H,W) float32, ct: (B,H,W) float32 transformed-domain positions."""
    B, C, H, W = F.shape
    r = F32(box_radius)
    l_pos = (ct - r).astype(F32)                                           # :65-66
    u_pos = (ct + r).astype(F32)
    # torch.cumsum on CPU accumulates float32 inputs in double and rounds each prefix to float32
    sat = np.zeros((B, C, H, W + 1), F32)                                  # :113-114
    sat[..., 1:] = np.cumsum(F, axis=-1, dtype=np.float64).astype(F32)
    out = np.empty_like(F)
    for b in range(B):
        for y in range(H):
            pos_ext = np.concatenate([ct[b, y], np.asarray([65535.0], F32)])        # :83-84, 2^16 - 1 = "infinity"
            li = _first_greater(pos_ext, l_pos[b, y])                                # :95-108
            ui = _first_greater(pos_ext, u_pos[b, y])
            den = ((ui - li).astype(np.int64).astype(F32) + F32(0.0001)).astype(F32)   # :134 (int64 + double scalar -> float32)
            out[b, :, y] = (sat[b, :, y][:, ui] - sat[b, :, y][:, li]) / den            # :128-134
    return out


def normalized_convolution(I: np.ndarray, sigma_s: float = 60, sigma_r: float = 0.4,
                           num_iterations: int = 3) -> np.ndarray:
    """Domain-transform normalized convolution (Gastal & Oliveira NC), the variant the reference's author
    recommends over RF for parallel hardware (RF.cpp:7-11).  NC.cpp:143-204."""
    I = np.asarray(I, dtype=F32)
    dx = np.abs(np.diff(I, axis=-1)).sum(axis=1, dtype=F32)               # :157-169
    dy = np.abs(np.diff(I, axis=-2)).sum(axis=1, dtype=F32)
    dx = np.pad(dx, [(0, 0), (0, 0), (1, 0)])
    dy = np.pad(dy, [(0, 0), (1, 0), (0, 0)])
    ratio = F32(F32(sigma_s) / F32(sigma_r))                               # float / float in C++  :173
    dHdx = (F32(1) + ratio * dx).astype(F32)
    dVdy = (F32(1) + ratio * dy).astype(F32)
    ct_H = np.cumsum(dHdx, axis=2, dtype=np.float64).astype(F32)          # :177-178
    ct_V = np.cumsum(dVdy, axis=1, dtype=np.float64).astype(F32)
    ct_Vt = np.ascontiguousarray(ct_V.transpose(0, 2, 1))                  # :181
    N = num_iterations
    F = I.copy()
    for i in range(N):
        # C++ float arithmetic: sqrt/pow return double, the product is rounded to float on assignment  :194-197
        sigma_i = F32(float(F32(sigma_s)) * math.sqrt(3) * math.pow(2, N - (i + 1)) / math.sqrt(math.pow(4, N) - 1))
        radius = F32(math.sqrt(3) * float(sigma_i))
        F = _nc_box_rows(F, ct_H, radius)                                  # :199
        Ft = np.ascontiguousarray(F.transpose(0, 1, 3, 2))                 # :200
        Ft = _nc_box_rows(Ft, ct_Vt, radius)                               # :202
        F = np.ascontiguousarray(Ft.transpose(0, 1, 3, 2))                 # :203
    return F


def edge_aware_filtering(x: np.ndarray, sigma_s: float, sigma_r: float, prefilter: str = "bilateral"):
    """deblurring.py:99-110.  The reference's live path is the bilateral filter (:108);
    the domain-transform call is the commented-out line :107 and is what BASELINE's
    config 3 asks for -- selectable here."""
    if prefilter == "bilateral":
        smooth = bilateral_filter(x)
    elif prefilter == "domain_transform":
        smooth = recursive_filter(x, sigma_s=sigma_s, sigma_r=sigma_r, num_iterations=1)
    elif prefilter == "normalized_convolution":
        smooth = normalized_convolution(x, sigma_s=sigma_s, sigma_r=sigma_r, num_iterations=1)
    else:
        raise ValueError("unknown prefilter %r" % prefilter)
    return smooth, (x - smooth).astype(F32)


# ----------------------------------------------------------------------------
# non-blind step and the driver                 (deblurring.py:211-239, :23-96)
# ----------------------------------------------------------------------------

def inverse_filtering_rank3(x: np.ndarray, kernel: np.ndarray, alpha: float = 2, b: float = 4,
                            remove_halo: bool = False, do_edgetaper: bool = False,
                            grad_img=None, method: str = "direct") -> np.ndarray:
    """deblurring.py:211-239: pad -> [edgetaper] -> polynomial -> crop -> [halo] -> clamp."""
    r = np.asarray(kernel[0] if method == "direct_separable" else kernel).shape[-1] // 2
    xp = replicate_pad(np.asarray(x, dtype=F32), r)
    if do_edgetaper:
        if method == "direct_separable":
            raise NotImplementedError("edgetaping is not defined for the separable approximation")
        xp = edgetaper(xp, kernel, method=method)
    y = crop(polynomial_deconvolution(xp, kernel, alpha, b, method=method), r)
    if remove_halo:
        y = halo_masking(crop(xp, r), y, grad_img)
    return np.clip(y, F32(0), F32(1)).astype(F32)


def polyblur_deblurring(img, n_iter=1, c=0.352, b=0.768, alpha=2, beta=3, sigma_r=0.8, sigma_s=2.0,
                        ker_size=25, q=0.0, n_angles=6, n_interpolated_angles=30, remove_halo=False,
                        edgetaping=False, prefiltering=False, discard_saturation=False,
                        multichannel_kernel=False, method="fft", verbose=False,
                        prefilter="bilateral", return_info=False):
    """deblurring.py:23-96.  ``prefilter`` and ``return_info`` are oracle-only extras."""
    a = np.asarray(img)
    as_image = a.ndim in (2, 3)
    x = to_planar(a) if as_image else np.asarray(a, dtype=F32)
    if x.ndim != 4:
        raise ValueError("expected (H,W), (H,W,C) or (B,C,H,W)")
    if method not in ("fft", "direct", "direct_separable"):
        raise ValueError("method %r is not runnable in the reference" % (method,))
    grad_img = spectral_gradients(x)                                       # :61
    pred = x
    infos = []
    for _ in range(n_iter):                                                # :68
        kernel, info = estimate_gaussian_blur(pred, c=c, b=b, q=q, n_angles=n_angles,
                                              n_interpolated_angles=n_interpolated_angles,
                                              ker_size=ker_size, discard_saturation=discard_saturation,
                                              multichannel=multichannel_kernel, return_info=True)
        info["kernel"] = kernel[:, 0]
        if method == "direct_separable":           # the opt-in x-t approximation of the same Gaussian (unpinned)
            k1, k2 = separable_xt_kernels(info["theta"], info["sigma"], info["rho"], ker_size)
            kernel = (k1[:, None], k2[:, None])
        if prefiltering:                                                   # :80-84
            smooth, detail = edge_aware_filtering(pred, sigma_s, sigma_r, prefilter)
            pred = inverse_filtering_rank3(smooth, kernel, alpha=alpha, b=beta, remove_halo=remove_halo,
                                           do_edgetaper=edgetaping, grad_img=grad_img, method=method)
            pred = pred + detail
        else:
            pred = inverse_filtering_rank3(pred, kernel, alpha=alpha, b=beta, remove_halo=remove_halo,
                                           do_edgetaper=edgetaping, grad_img=grad_img, method=method)
        pred = np.clip(pred, F32(0), F32(1)).astype(F32)                   # :88
        info["image"] = pred
        infos.append(info)
    out = from_planar(pred) if as_image else pred
    return (out, infos) if return_info else out


# --------------------------------------------------------------------------------------------
# 8-bit file edge of the CLI flow (main.py:80-82,146).  The conversions live in scikit-image
# (requirements.txt:4 pins 0.19.2), which is absent here and from /root/reference, so they are
# restated from its published algorithm (skimage/util/dtype.py `_convert`): unsigned -> float32 is
# `np.multiply(image, 1. / 255, dtype=np.float32)`; float -> uint8 is `np.multiply(image, 255)`,
# `np.rint`, `np.clip(0, 255)`, cast.  Parity of this edge is UNPINNED against scikit-image itself.
# --------------------------------------------------------------------------------------------
def img_as_float32_from_ubyte(img: np.ndarray) -> np.ndarray:
    assert img.dtype == np.uint8
    return np.multiply(img, 1.0 / 255, dtype=np.float32)


def img_as_ubyte_from_float(img: np.ndarray) -> np.ndarray:
    out = np.multiply(np.asarray(img, np.float32), 255, dtype=np.float32)
    np.rint(out, out=out)
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)


def polyblur_deblurring_uint8(img: np.ndarray, **kwargs) -> np.ndarray:
    """main.py:80-82,146: uint8 (H,W)/(H,W,C) image -> float32 -> polyblur_deblurring -> uint8."""
    return img_as_ubyte_from_float(polyblur_deblurring(img_as_float32_from_ubyte(img), **kwargs))


def kaiser_window_periodic(n: int, beta: float = 5.0) -> np.ndarray:
    """torch.kaiser_window(n, periodic=True, beta=5) (deblurring.py:352)."""
    if n == 1:
        return np.ones(1, F32)
    r = 2.0 * np.arange(n, dtype=np.float64) / n - 1.0
    return (np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / np.i0(beta)).astype(F32)


def patchwise_deblurring(images: np.ndarray, patch_size=(400, 400), patch_overlap=0.25, **kwargs) -> np.ndarray:
    """The patch branch of PolyblurDeblurring.forward (deblurring.py:269-340), FIX-FORWARD: the reference
    raises NameError on the undefined `handling_saturation` (:289,:315) and indexes a batch of restored
    patches as [n::batch_size] (:335), which is only right for B == 1.  This restatement drops the
    saturation branch and indexes per image; it is NOT pinned by reference goldens (the reference
    cannot run this branch) -- parity for this piece is "unpinned"."""
    x = np.asarray(images, dtype=F32)
    h, w = x.shape[-2:]
    if h % 2 == 1:
        x, h = x[..., :-1, :], h - 1
    if w % 2 == 1:
        x, w = x[..., :, :-1], w - 1
    ph, pw = patch_size
    step_h, step_w = int(ph * (1 - patch_overlap)), int(pw * (1 - patch_overlap))
    new_h = int(np.ceil((h - ph) / step_h) * step_h) + ph
    new_w = int(np.ceil((w - pw) / step_w) * step_w) + pw
    pl, pr = int(np.floor((new_w - w) / 2)), int(np.ceil((new_w - w) / 2))
    pt, pb = int(np.floor((new_h - h) / 2)), int(np.ceil((new_h - h) / 2))
    xp = np.pad(x, [(0, 0), (0, 0), (pt, pb), (pl, pr)], mode="edge")
    window = (kaiser_window_periodic(ph)[:, None] * kaiser_window_periodic(pw)[None, :]).astype(F32)
    acc = np.zeros_like(xp)
    wsum = np.zeros(xp.shape[-2:], F32)
    for i0 in range(0, new_h - ph + 1, step_h):
        for j0 in range(0, new_w - pw + 1, step_w):
            res = polyblur_deblurring(np.ascontiguousarray(xp[..., i0:i0 + ph, j0:j0 + pw]), **kwargs)
            acc[..., i0:i0 + ph, j0:j0 + pw] += res * window
            wsum[i0:i0 + ph, j0:j0 + pw] += window
    out = np.clip(acc / (wsum + F32(1e-8)), F32(0), F32(1)).astype(F32)
    return np.ascontiguousarray(out[..., pt:pt + h, pl:pl + w])


class PolyblurDeblurring:
    """deblurring.py:250-347.  The patch branch is the fix-forward restatement above (the reference's
    raises NameError).  Note the differing defaults (:266-268)."""

    def __init__(self, patch_decomposition=False, patch_size=400, patch_overlap=0.25, batch_size=1):
        self.patch_decomposition = patch_decomposition
        self.patch_size = (patch_size, patch_size)
        self.patch_overlap = patch_overlap
        self.batch_size = batch_size

    def __call__(self, images, n_iter=1, c=0.352, b=0.468, alpha=2, beta=4, sigma_s=2, ker_size=25,
                 sigma_r=0.4, q=0.0, n_angles=6, n_interpolated_angles=30, remove_halo=False,
                 edgetaping=False, prefiltering=False, discard_saturation=False,
                 multichannel_kernel=False, method="fft", device=None):
        kw = dict(n_iter=n_iter, c=c, b=b, alpha=alpha, beta=beta, ker_size=ker_size, sigma_s=sigma_s, sigma_r=sigma_r,
                  remove_halo=remove_halo, edgetaping=edgetaping, prefiltering=prefiltering,
                  discard_saturation=discard_saturation, multichannel_kernel=multichannel_kernel, method=method, q=q,
                  n_angles=n_angles, n_interpolated_angles=n_interpolated_angles)
        if self.patch_decomposition:
            return patchwise_deblurring(images, self.patch_size, self.patch_overlap, **kw)
        return polyblur_deblurring(images, **kw)

    forward = __call__
