"""Drop-in for the reference's public API (polyblur/deblurring.py:23-25 and :250-347).

    from polyblur_amd import polyblur_deblurring, PolyblurDeblurring

Same names, keyword arguments, defaults (including the functional/module defaults that
differ in the reference), and the same in/out type rule:

* ``np.ndarray`` (H,W) or (H,W,C)  ->  ``np.ndarray`` of the same shape, float32
  (deblurring.py:46-48,93-94);
* ``torch.Tensor`` (B,C,H,W)       ->  ``torch.Tensor`` on the same device, same dtype.
  ROCm tensors are used in place (zero copy, on torch's current stream); CPU tensors
  are staged through the GPU.

The work is done by the HIP engine behind include/polyblur_hip.h; there is no CPU
implementation here.  ``method`` keeps the reference's meaning as a *boundary model*:
'fft' = circular reblurring on the replicate-padded domain (the reference default),
'direct' = zero-padded reblurring -- both evaluated by the same reblurring pass (separable or 2-D
stencil, or -- dense kernels -- overlap-save on 64x64 windows inside LDS; no image-sized transform
exists on the device).

Extras that the reference does not have (all keyword-only, defaults keep reference
behaviour): ``support`` ('full' | 'adaptive'), ``prefilter`` ('bilateral' |
'domain_transform' -- which edge-aware filter ``prefiltering=True`` uses; the reference's
live code path is the bilateral one, deblurring.py:107-108), ``return_info``, ``temporaries``
('fp32' | 'fp16': float16 tensors only -- store the two Horner temporaries as fp16).
"""
from __future__ import annotations

from time import time

import numpy as np

from . import _capi as capi
from .engine import get_engine

# 'direct_separable': the reference's own separable path raises NameError (blur_estimation.py:77, filters.py:71,78); its
# intent -- a 1-D pass followed by an oblique 1-D pass with linear interpolation, separable_gaussian2d.cpp:91-183 -- is an
# opt-in APPROXIMATION of the 25x25 kernel here (zero boundary like 'direct'); rank-1 kernels are exact either way
_METHODS = {"fft": capi.PB_WRAP, "direct": capi.PB_ZERO, "direct_separable": capi.PB_ZERO}
_SUPPORT = {"full": capi.PB_SUPPORT_FULL, "adaptive": capi.PB_SUPPORT_ADAPTIVE}
_PREFILTER = {"bilateral": capi.PB_PREFILTER_BILATERAL, "domain_transform": capi.PB_PREFILTER_DOMAIN_TRANSFORM,
              "normalized_convolution": capi.PB_PREFILTER_NORMALIZED_CONVOLUTION}


def _check_image_size(h, w):
    """Line lengths the spectral derivative takes (include/polyblur_hip.h: pb_fft_length_supported): whole lines in LDS up
    to 20480 samples (8192 when a prime factor exceeds 7), through a line buffer in device memory up to 65536."""
    lib = capi.load_library()
    for n, what in ((int(h), "height"), (int(w), "width")):
        if n >= 2 and not lib.pb_fft_length_supported(n):
            raise ValueError("image %s %d is beyond the engine's spectral derivative: sides up to 65536 are supported" % (what, n))


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


def _build_options(C, n_iter, c, b, alpha, beta, sigma_r, sigma_s, ker_size, q, n_angles, n_interpolated_angles,
                   remove_halo, edgetaping, prefiltering, discard_saturation, multichannel_kernel, method, support,
                   prefilter, force_theta_deg=-1.0, temporaries="fp32"):
    if method == "direct_separable" and edgetaping:
        raise NotImplementedError("edgetaping is not defined for method='direct_separable'")
    if method not in _METHODS:
        raise ValueError("%s not implemented" % method)          # reference: deblurring.py:119 (never raised there)
    if temporaries not in ("fp32", "fp16"):
        raise ValueError("temporaries must be 'fp32' or 'fp16'")
    if support not in _SUPPORT:
        raise ValueError("support must be 'full' or 'adaptive'")
    if prefilter not in _PREFILTER:
        raise ValueError("prefilter must be 'bilateral', 'domain_transform' or 'normalized_convolution'")
    if not (isinstance(ker_size, (int, np.integer)) and 2 <= ker_size <= capi.PB_KSIZE_MAX):
        # up to 25 the kernel lives in a 25 x 25 record; 26 .. 49 take the large-kernel pass (csrc/conv_big.hip); the replicate
        # pad is ker_size // 2 (even sizes: off-centre, as the reference's grid arange(k) - (k - 1) // 2 and its two convolution
        # paths place them).  sigma is clamped to 4, so 49 holds +-6 sigma: nothing beyond it is built
        raise NotImplementedError("ker_size must be between 2 and 49")
    if ker_size % 2 == 0 and method == "direct_separable":
        raise NotImplementedError("an even ker_size is built for the 'fft' / 'direct' methods only (not with 'direct_separable')")
    if ker_size > capi.PB_KSIZE and method == "direct_separable":
        raise NotImplementedError("a ker_size above 25 is built for the 'fft' / 'direct' methods only (not with 'direct_separable')")
    if not (0 <= q < 0.5):
        raise ValueError("q must be in [0, 0.5)")
    if multichannel_kernel and C not in (1, 3):
        raise NotImplementedError("per-channel kernels crash in the reference for C not in {1,3}")
    if not (1 <= n_angles <= capi.PB_MAX_ANGLES - 1 and 1 <= n_interpolated_angles <= capi.PB_MAX_INTERP):
        raise ValueError("n_angles / n_interpolated_angles out of range")
    from .engine import Engine
    return Engine.make_options(n_iter=n_iter, c=c, b=b, alpha=alpha, beta=beta, sigma_r=sigma_r, sigma_s=sigma_s, q=q,
                               n_angles=n_angles, n_interpolated_angles=n_interpolated_angles, remove_halo=remove_halo,
                               edgetaping=edgetaping,
                               prefilter=_PREFILTER[prefilter] if prefiltering else capi.PB_PREFILTER_NONE,
                               discard_saturation=discard_saturation, boundary=_METHODS[method],
                               support=_SUPPORT[support], force_theta_deg=force_theta_deg, ker_size=ker_size,
                               separable_approx=(method == "direct_separable"), half_temporaries=(temporaries == "fp16"))


def _info_to_dicts(info, n_angles, n_interp):
    if info is None:
        return None
    out = []
    for it in range(info.shape[0]):
        rec = info[it]
        out.append(dict(mags=rec["mags"][:, :n_angles + 1].copy(), interp=rec["interp"][:, :n_interp].copy(),
                        i_min=rec["i_min"].copy(), theta=rec["theta"].copy(), sigma=rec["sigma"].copy(),
                        rho=rec["rho"].copy(), kernel=rec["kernel"].copy(), separable=rec["separable"].copy(),
                        radius=rec["radius"].copy(), nphase=rec["nphase"].copy(), lo=rec["gray_min"].copy(),
                        hi=rec["gray_max"].copy()))
    return out


def _print_stage_times(prof, wall):
    """verbose=True: device time per stage from hipEvents around every launch (the reference prints host
    wall-clock per stage without synchronising, deblurring.py:59-90)."""
    ms = {k: v[0] for k, v in prof.items()}
    est = ms["gray"] + ms["grad_rows"] + ms["grad_cols"] + ms["params"]
    print('-- blur estimation:   %1.5f  (gray/min-max %1.5f, row derivative %1.5f, column derivative + maxima %1.5f, '
          'parameters %1.5f)' % (est / 1e3, ms["gray"] / 1e3, ms["grad_rows"] / 1e3, ms["grad_cols"] / 1e3, ms["params"] / 1e3))
    deb = ms["conv"] + ms["halo"] + ms["prefilter"] + ms["other"]
    print('-- deblurring:        %1.5f  (%d stencil passes %1.5f, halo masking %1.5f, prefilter %1.5f, other %1.5f)'
          % (deb / 1e3, prof["conv"][1], ms["conv"] / 1e3, ms["halo"] / 1e3, ms["prefilter"] / 1e3, ms["other"] / 1e3))
    print('-- polyblur (hip):    %1.5f s wall' % wall)


def polyblur_deblurring(img, n_iter=1, c=0.352, b=0.768, alpha=2, beta=3, sigma_r=0.8, sigma_s=2.0, ker_size=25, q=0.0,
                        n_angles=6, n_interpolated_angles=30, remove_halo=False, edgetaping=False, prefiltering=False,
                        discard_saturation=False, multichannel_kernel=False, method='fft', verbose=False, *,
                        support='full', prefilter='bilateral', return_info=False, device=None, temporaries='fp32'):
    """Blind deblurring of ``img`` -- see the module docstring; reference deblurring.py:23-96."""
    start = time()
    if isinstance(img, np.ndarray):
        # utils.to_tensor + unsqueeze (deblurring.py:46-48): HWC -> (1,C,H,W) float32 copy
        if img.ndim == 2:
            x = img[None, None]
        elif img.ndim == 3:
            x = np.moveaxis(img, 2, 0)[None]
        else:
            raise ValueError("expected an (H,W) or (H,W,C) array, got shape %r" % (img.shape,))
        x = np.ascontiguousarray(x, dtype=np.float32)
        _check_image_size(*x.shape[-2:])
        opts = _build_options(x.shape[1], n_iter, c, b, alpha, beta, sigma_r, sigma_s, ker_size, q, n_angles,
                              n_interpolated_angles, remove_halo, edgetaping, prefiltering, discard_saturation,
                              multichannel_kernel, method, support, prefilter)
        eng = get_engine(0 if device is None else int(device))
        eng.set_stream(0)                                   # host arrays: the device's default stream
        if verbose:
            eng.profile_begin()
        try:
            res = eng.polyblur(x, opts, want_info=return_info)
        finally:                                            # (a failed call must not leave the thread's context profiling)
            prof = eng.profile_end() if verbose else None
        if verbose:
            _print_stage_times(prof, time() - start)
        out, info = res if return_info else (res, None)
        # utils.to_array (utils.py:24-31): squeeze, CHW -> HWC
        out = np.squeeze(out)
        if out.ndim == 3:
            out = np.ascontiguousarray(np.moveaxis(out, 0, -1))
        return (out, _info_to_dicts(info, n_angles, n_interpolated_angles)) if return_info else out

    if not _is_torch_tensor(img):
        raise TypeError("img must be a numpy.ndarray or a torch.Tensor")
    import torch
    if img.dim() != 4:
        raise ValueError("expected a (B,C,H,W) tensor, got shape %r" % (tuple(img.shape),))
    if img.dtype not in (torch.float32, torch.float16):
        raise TypeError("tensor dtype must be float32 or float16 (the reference is float32-only)")
    _check_image_size(*img.shape[-2:])
    if temporaries == "fp16" and img.dtype != torch.float16:
        raise ValueError("temporaries='fp16' applies to float16 images only")
    opts = _build_options(img.shape[1], n_iter, c, b, alpha, beta, sigma_r, sigma_s, ker_size, q, n_angles,
                          n_interpolated_angles, remove_halo, edgetaping, prefiltering, discard_saturation,
                          multichannel_kernel, method, support, prefilter, temporaries=temporaries)
    dtype = capi.PB_F32 if img.dtype == torch.float32 else capi.PB_F16
    if img.is_cuda:
        dev = img.device.index if img.device.index is not None else torch.cuda.current_device()
        eng = get_engine(dev)
        xin = img.contiguous()
        out = torch.empty_like(xin)
        with torch.cuda.device(dev):
            eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            if verbose:
                eng.profile_begin()
            try:
                info = eng.polyblur_ptr(xin.data_ptr(), out.data_ptr(), dtype, xin.shape, opts, want_info=return_info)
            finally:                                        # (a failed call must not leave the thread's context profiling)
                prof = eng.profile_end() if verbose else None
    else:
        eng = get_engine(0 if device is None else int(device))
        eng.set_stream(0)
        if verbose:
            eng.profile_begin()
        arr = img.detach().contiguous().numpy()
        try:
            res = eng.polyblur(arr, opts, want_info=return_info)
        finally:
            prof = eng.profile_end() if verbose else None
        o, info = res if return_info else (res, None)
        out = torch.from_numpy(o)
    if verbose:
        _print_stage_times(prof, time() - start)
    return (out, _info_to_dicts(info, n_angles, n_interpolated_angles)) if return_info else out


def polyblur_deblurring_uint8(img, n_iter=1, c=0.352, b=0.768, alpha=2, beta=3, sigma_r=0.8, sigma_s=2.0, ker_size=25,
                              q=0.0, n_angles=6, n_interpolated_angles=30, remove_halo=False, edgetaping=False,
                              prefiltering=False, discard_saturation=False, multichannel_kernel=False, method='fft',
                              verbose=False, *, support='full', prefilter='bilateral', return_info=False, device=None):
    """8-bit images in, 8-bit images out: the reference CLI's
    ``img_as_ubyte(polyblur_deblurring(img_as_float32(imread(...)), ...))`` (main.py:80-82,146) with both
    conversions done inside the first / last device kernels (scikit-image 0.19.2 semantics:
    ``v * float32(1/255)`` on load, ``clip(rint(v * 255), 0, 255)`` on store; fp32 in between).

    ``numpy.ndarray`` uint8 (H,W) / (H,W,C) -> same shape uint8;  ``torch.Tensor`` uint8 (B,C,H,W) -> same."""
    start = time()
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8:
            raise TypeError("expected a uint8 image, got %s" % img.dtype)
        if img.ndim == 2:
            x = img[None, :, :, None]
        elif img.ndim == 3:
            x = img[None]
        else:
            raise ValueError("expected an (H,W) or (H,W,C) array, got shape %r" % (img.shape,))
        opts = _build_options(x.shape[3], n_iter, c, b, alpha, beta, sigma_r, sigma_s, ker_size, q, n_angles,
                              n_interpolated_angles, remove_halo, edgetaping, prefiltering, discard_saturation,
                              multichannel_kernel, method, support, prefilter)
        eng = get_engine(0 if device is None else int(device))
        eng.set_stream(0)
        res = eng.polyblur_u8_hwc(x, opts, want_info=return_info)
        out, info = res if return_info else (res, None)
        out = out.reshape(img.shape)
    else:
        if not _is_torch_tensor(img):
            raise TypeError("img must be a numpy.ndarray or a torch.Tensor")
        import torch
        if img.dim() != 4 or img.dtype != torch.uint8:
            raise TypeError("expected a (B,C,H,W) uint8 tensor")
        opts = _build_options(img.shape[1], n_iter, c, b, alpha, beta, sigma_r, sigma_s, ker_size, q, n_angles,
                              n_interpolated_angles, remove_halo, edgetaping, prefiltering, discard_saturation,
                              multichannel_kernel, method, support, prefilter)
        if img.is_cuda:
            dev = img.device.index if img.device.index is not None else torch.cuda.current_device()
            eng = get_engine(dev)
            xin = img.contiguous()
            out = torch.empty_like(xin)
            with torch.cuda.device(dev):
                eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                info = eng.polyblur_ptr(xin.data_ptr(), out.data_ptr(), capi.PB_U8, xin.shape, opts, want_info=return_info)
        else:
            eng = get_engine(0 if device is None else int(device))
            eng.set_stream(0)
            res = eng.polyblur(img.detach().contiguous().numpy(), opts, want_info=return_info)
            o, info = res if return_info else (res, None)
            out = torch.from_numpy(o)
    if verbose:
        print('-- polyblur (hip, uint8): %1.5f s' % (time() - start))
    return (out, _info_to_dicts(info, n_angles, n_interpolated_angles)) if return_info else out


def kaiser_window_periodic(n: int, beta: float = 5.0) -> np.ndarray:
    """torch.kaiser_window(n, periodic=True, beta) (deblurring.py:352): the symmetric window of
    length n+1 without its last sample."""
    if n == 1:
        return np.ones(1, np.float32)
    k = np.arange(n, dtype=np.float64)
    r = 2.0 * k / n - 1.0
    return (np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / np.i0(beta)).astype(np.float32)


def patch_grid(h: int, w: int, patch_size, overlap: float):
    """The reference's patch lattice (deblurring.py:281-299): steps, padded size, pads, counts."""
    ph, pw = patch_size
    step_h, step_w = int(ph * (1 - overlap)), int(pw * (1 - overlap))
    new_h = int(np.ceil((h - ph) / step_h) * step_h) + ph
    new_w = int(np.ceil((w - pw) / step_w) * step_w) + pw
    pad_top, pad_left = int(np.floor((new_h - h) / 2)), int(np.floor((new_w - w) / 2))
    n_i = len(range(0, new_h - ph + 1, step_h))
    n_j = len(range(0, new_w - pw + 1, step_w))
    return dict(ph=ph, pw=pw, step_h=step_h, step_w=step_w, new_h=new_h, new_w=new_w, pad_top=pad_top,
                pad_left=pad_left, n_i=n_i, n_j=n_j)


def _device_index(device):
    """`device` of PolyblurDeblurring.forward (deblurring.py:322-330: where the patches are deblurred) -> GPU index."""
    if device is None:
        return None
    if isinstance(device, int):
        return device
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError("device=%r: the engine runs on a GPU only (there is no CPU implementation)" % (device,))
    return d.index if d.index is not None else torch.cuda.current_device()


def _patchwise_deblurring(images, patch_size, overlap, batch_size, kwargs, device=None):
    """PolyblurDeblurring.forward, patch branch (deblurring.py:269-340, fix-forward)."""
    import ctypes as C

    import torch
    if not _is_torch_tensor(images) or images.dim() != 4:
        raise ValueError("patch decomposition expects a (B,C,H,W) tensor")
    if images.dtype not in (torch.float32, torch.float16):
        raise TypeError("tensor dtype must be float32 or float16")
    if kwargs.pop("return_info", False):
        raise TypeError("return_info is not available with patch_decomposition=True (one record set per patch group)")
    # (the lattice first: a plain error for what the reference's arithmetic leaves without a single patch -- an image shorter than
    #  patch_size - step along an axis gives new_h < patch_size at deblurring.py:284-285 and an empty I_coords at :294: the
    #  reference's branch, could it run, would return zeros there)
    if min(patch_size) < 2 or int(patch_size[0] * (1 - overlap)) < 1 or int(patch_size[1] * (1 - overlap)) < 1:
        raise ValueError("patch size / overlap incompatible: patches of at least 2x2 with a positive step are needed")
    g0 = patch_grid(images.shape[-2] // 2 * 2, images.shape[-1] // 2 * 2, patch_size, overlap)
    if g0["n_i"] < 1 or g0["n_j"] < 1:
        raise ValueError("image %dx%d is smaller than patch_size - step along an axis: the reference's lattice (deblurring.py:284-295) "
                         "holds no patch for it -- use a smaller patch_size or patch_decomposition=False" % (images.shape[-2], images.shape[-1]))
    was_cuda = images.is_cuda
    want = _device_index(device)
    if was_cuda:
        dev = images.device.index if images.device.index is not None else torch.cuda.current_device()
        if want is not None and want != dev:
            raise ValueError("images live on cuda:%d but device=cuda:%d was requested" % (dev, want))
    else:
        dev = 0 if want is None else want                   # reference: patches.to(device) ... .cpu() (deblurring.py:322-330)
    x = images if was_cuda else images.to("cuda:%d" % dev)
    h, w = x.shape[-2:]
    if h % 2 == 1:                                      # :273-279 make the size even
        x = x[..., :-1, :]
        h -= 1
    if w % 2 == 1:
        x = x[..., :, :-1]
        w -= 1
    x = x.contiguous()
    B, Cc = x.shape[:2]
    g = patch_grid(h, w, patch_size, overlap)
    if g["ph"] < 2 or g["pw"] < 2 or g["step_h"] < 1 or g["step_w"] < 1:
        raise ValueError("patch size / overlap incompatible: patches of at least 2x2 with a positive step are needed")
    eng = get_engine(dev)
    dtype = capi.PB_F32 if x.dtype == torch.float32 else capi.PB_F16
    n_p = g["n_i"] * g["n_j"]
    wy = torch.from_numpy(kaiser_window_periodic(g["ph"])).to(x.device)
    wx = torch.from_numpy(kaiser_window_periodic(g["pw"])).to(x.device)
    restored = torch.empty((n_p * B, Cc, g["ph"], g["pw"]), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        group = max(1, int(batch_size))
        for first in range(0, n_p, group):                       # :310 groups of `batch_size` patches
            count = min(group, n_p - first)
            patches = torch.empty((count * B, Cc, g["ph"], g["pw"]), dtype=x.dtype, device=x.device)
            eng._check(eng.lib.pb_extract_patches(eng.ctx, x.data_ptr(), patches.data_ptr(), dtype, B, Cc, h, w, g["ph"],
                                                  g["pw"], g["step_h"], g["step_w"], g["n_i"], g["n_j"], g["pad_top"],
                                                  g["pad_left"], first, count))
            restored[first * B:(first + count) * B] = polyblur_deblurring(patches, **kwargs)
        out = torch.empty_like(x)
        eng._check(eng.lib.pb_overlap_add(eng.ctx, restored.data_ptr(), out.data_ptr(), dtype, B, Cc, h, w, g["ph"], g["pw"],
                                          g["step_h"], g["step_w"], g["n_i"], g["n_j"], g["pad_top"], g["pad_left"],
                                          C.c_void_p(wy.data_ptr()), C.c_void_p(wx.data_ptr())))
        torch.cuda.current_stream(dev).synchronize()             # wy / wx / restored must outlive the kernels
    return out if was_cuda else out.cpu()


class _ModuleBase:
    pass


try:                                      # be an nn.Module when torch is importable, like the reference
    import torch.nn as _nn
    _Base = _nn.Module
except Exception:                         # pragma: no cover
    _Base = _ModuleBase


class PolyblurDeblurring(_Base):
    """Stateless module wrapper, reference deblurring.py:250-347.

    ``patch_decomposition=True`` raises ``NameError`` in the reference (undefined
    ``handling_saturation``, deblurring.py:289).  Here it is built "fix-forward" (SURVEY 8f row 1):
    the saturation branch is dropped and the restored patches of a batch are indexed per image;
    everything else -- even-size crop, replicate padding to the patch grid, periodic Kaiser(beta=5)
    window, overlap-add normalised by the summed window + 1e-8, clamp, crop -- follows :269-340.
    Note the defaults of ``forward`` differ from the functional API's (deblurring.py:266-268).
    """

    def __init__(self, patch_decomposition=False, patch_size=400, patch_overlap=0.25, batch_size=1):
        super().__init__()
        self.batch_size = batch_size
        self.patch_decomposition = patch_decomposition
        self.patch_size = (patch_size, patch_size)
        self.patch_overlap = patch_overlap

    def forward(self, images, n_iter=1, c=0.352, b=0.468, alpha=2, beta=4, sigma_s=2, ker_size=25, sigma_r=0.4,
                q=0.0, n_angles=6, n_interpolated_angles=30, remove_halo=False, edgetaping=False, prefiltering=False,
                discard_saturation=False, multichannel_kernel=False, method='fft', device=None, **extras):
        if self.patch_decomposition:
            return _patchwise_deblurring(images, self.patch_size, self.patch_overlap, self.batch_size,
                                         dict(n_iter=n_iter, c=c, b=b, alpha=alpha, beta=beta, ker_size=ker_size,
                                              sigma_s=sigma_s, sigma_r=sigma_r, remove_halo=remove_halo,
                                              edgetaping=edgetaping, prefiltering=prefiltering,
                                              discard_saturation=discard_saturation,
                                              multichannel_kernel=multichannel_kernel, method=method, q=q,
                                              n_angles=n_angles, n_interpolated_angles=n_interpolated_angles, **extras),
                                         device=device)
        return polyblur_deblurring(images, n_iter=n_iter, c=c, b=b, alpha=alpha, beta=beta, ker_size=ker_size,
                                   sigma_s=sigma_s, sigma_r=sigma_r, remove_halo=remove_halo, edgetaping=edgetaping,
                                   prefiltering=prefiltering, discard_saturation=discard_saturation,
                                   multichannel_kernel=multichannel_kernel, method=method, q=q, n_angles=n_angles,
                                   n_interpolated_angles=n_interpolated_angles, device=_device_index(device), **extras)

    if _Base is _ModuleBase:              # pragma: no cover
        __call__ = forward
