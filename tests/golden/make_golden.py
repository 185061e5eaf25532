#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
The reference is imported read-only with a stub for the missing scikit-image module
(SURVEY.md Appendix A); nothing from it is copied -- the fixtures hold inputs and
the reference's outputs only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures (all float32 unless noted):
  stages_{A,B,C}.npz     every L1 function on small synthetic inputs
  kernel_grid.npz        create_gaussian_filter over a (sigma, rho, theta) grid
  pipeline_peacock.npz   n_iter=3 on pictures/peacock_defocus.png, per-iteration
                         intermediates, methods fft + direct
  pipeline_variants.npz  option variants on a 160x224 crop
  pipeline_strongblur.npz  border-semantics stress (sigma=3.5 blur)
  pipeline_batch.npz     B=2 batch through method='fft'
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sk = types.ModuleType("skimage")
sk.img_as_float32 = lambda x: np.asarray(x, np.float32) / (255.0 if np.asarray(x).dtype == np.uint8 else 1.0)
sys.modules["skimage"] = sk
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import torch  # noqa: E402

torch.set_num_threads(8)
from polyblur import polyblur_deblurring, PolyblurDeblurring  # noqa: E402
from polyblur import blur_estimation, filters, edgetaper, domain_transform, utils, deblurring  # noqa: E402
from PIL import Image  # noqa: E402

from polyblur_amd.synthetic import synthetic_image, synthetic_blurry_batch  # noqa: E402

T = torch.from_numpy


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.dtype.is_floating_point else t.detach().cpu().numpy()


class Recorder:
    """Wrap the estimation sub-routines (looked up through the module globals by
    gaussian_blur_estimation) to capture per-iteration intermediates."""

    NAMES = ["compute_gradient_magnitudes", "find_maximal_blur_direction",
             "compute_gaussian_parameters", "create_gaussian_filter", "normalize"]

    def __init__(self):
        self.log = []
        self.orig = {}

    def __enter__(self):
        for n in self.NAMES:
            self.orig[n] = getattr(blur_estimation, n)
            setattr(blur_estimation, n, self._wrap(n, self.orig[n]))
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(blur_estimation, n, f)

    def _wrap(self, name, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            self.log.append((name, out))
            return out
        return inner

    def per_iteration(self):
        its, cur = [], {}
        for name, out in self.log:
            if name == "normalize":
                cur = {}
                its.append(cur)
            elif name == "compute_gradient_magnitudes":
                cur["mags"] = npy(out)
            elif name == "find_maximal_blur_direction":
                cur["m_normal"], cur["m_ortho"], cur["theta"] = (npy(out[0])[:, 0], npy(out[1])[:, 0], npy(out[2])[:, 0])
            elif name == "compute_gaussian_parameters":
                cur["sigma"], cur["rho"] = npy(out[0])[:, 0], npy(out[1])[:, 0]
            elif name == "create_gaussian_filter":
                cur["kernel"] = npy(out)[:, 0]
        return its


def run_pipeline(x, **kw):
    """Run the reference driver on a (B,C,H,W) float32 array one iteration at a time is
    not possible (state is internal), so run it whole with the recorder attached, and
    additionally re-run with n_iter = 1..N to get the image after each iteration."""
    n_iter = kw.pop("n_iter")
    with Recorder() as rec:
        out = npy(polyblur_deblurring(T(x), n_iter=n_iter, **kw))
    its = rec.per_iteration()
    assert len(its) == n_iter
    imgs = [npy(polyblur_deblurring(T(x), n_iter=k, **kw)) for k in range(1, n_iter)] + [out]
    return out, its, imgs


def pack_iterations(d, prefix, its, imgs=None):
    for n, it in enumerate(its):
        for k, v in it.items():
            d["%s/it%d/%s" % (prefix, n, k)] = v
        if imgs is not None:
            d["%s/it%d/image" % (prefix, n)] = imgs[n]


def interp_from_mags(mags):
    thetas = torch.linspace(0, 180, 7).unsqueeze(0).long()
    ith = torch.arange(0, 180, 6.0).unsqueeze(0).long()
    return npy(blur_estimation.cubic_interpolator(ith / 30, thetas / 30, T(mags)))


def stage_fixture(x, name, direct=True):
    """All L1 functions on one synthetic input x (B,C,H,W)."""
    d = {"x": x}
    tx = T(x)
    gx, gy = filters.fourier_gradients(tx)
    d["grad_x"], d["grad_y"] = npy(gx), npy(gy)
    gray = tx.mean(dim=1, keepdims=True) if x.shape[1] == 3 else tx
    d["gray"] = npy(gray)
    norm = blur_estimation.normalize(gray, q=0.0)
    d["normalized"] = npy(norm)
    mask = blur_estimation.get_saturation_mask(gray, False)
    g = blur_estimation.compute_gradients(norm, mask)
    d["norm_grad_x"], d["norm_grad_y"] = npy(g[0]), npy(g[1])
    mags = blur_estimation.compute_gradient_magnitudes(g)
    d["mags"] = npy(mags)
    d["interp"] = interp_from_mags(d["mags"])
    thetas = torch.linspace(0, 180, 7).unsqueeze(0).long()
    ith = torch.arange(0, 180, 6.0).unsqueeze(0).long()
    mn, mo, th = blur_estimation.find_maximal_blur_direction(mags, thetas, ith)
    d["m_normal"], d["m_ortho"], d["theta"] = npy(mn)[:, 0], npy(mo)[:, 0], npy(th)[:, 0]
    sg, rh = blur_estimation.compute_gaussian_parameters(mn, mo, c=0.362, b=0.468)
    d["sigma"], d["rho"] = npy(sg)[:, 0], npy(rh)[:, 0]
    ker = blur_estimation.create_gaussian_filter(th, sg, rh, ksize=25)
    d["kernel"] = npy(ker)[:, 0]
    # a deliberately wide, rotated kernel for the convolution stages (border stress)
    B = x.shape[0]
    th2 = torch.tensor([[np.deg2rad(24.0 + 18 * i)] for i in range(B)], dtype=torch.float32)
    sg2 = torch.tensor([[2.2 + 0.4 * i] for i in range(B)], dtype=torch.float32)
    rh2 = torch.tensor([[0.9 + 0.2 * i] for i in range(B)], dtype=torch.float32)
    kw = blur_estimation.create_gaussian_filter(th2, sg2, rh2, ksize=25)
    d["kwide"] = npy(kw)[:, 0]
    d["kwide_params"] = np.stack([npy(sg2)[:, 0], npy(rh2)[:, 0], npy(th2)[:, 0]], 1)
    xp = utils.pad_with_kernel(tx, kw)
    for kname, kk in (("kest", ker), ("kwide", kw)):
        d["poly_fft_%s" % kname] = npy(deblurring.compute_polynomial_fft(xp, kk, 6.0, 1.0))
        if direct and B == 1:
            d["poly_direct_%s" % kname] = npy(deblurring.compute_polynomial_direct(xp, kk, 6.0, 1.0))
        d["inv_fft_%s" % kname] = npy(deblurring.inverse_filtering_rank3(tx, kk, alpha=6.0, b=1.0, method="fft"))
    if B == 1:
        d["conv_fft_kwide"] = npy(filters.convolve2d(xp, kw, method="fft"))
        d["conv_direct_kwide"] = npy(filters.convolve2d(xp, kw, method="direct"))
        d["taper_alpha_kwide"] = npy(edgetaper.edgetaper_alpha(kw, xp.shape[-2:]))
        d["taper_fft_kwide"] = npy(edgetaper.edgetaper(xp, kw, method="fft"))
        d["taper_direct_kwide"] = npy(edgetaper.edgetaper(xp, kw, method="direct"))
        y = deblurring.inverse_filtering_rank3(tx, kw, alpha=6.0, b=1.0, method="fft")
        d["halo_kwide"] = npy(deblurring.halo_masking(tx, y, (gx, gy)))
    d["bilateral"] = npy(filters.bilateral_filter(tx))
    d["rf_n1"] = npy(domain_transform.recursive_filter(tx, sigma_s=2.0, sigma_r=0.8, num_iterations=1))
    d["rf_n3"] = npy(domain_transform.recursive_filter(tx, sigma_s=60, sigma_r=0.4, num_iterations=3))
    np.savez_compressed(os.path.join(HERE, name), **d)
    print("wrote", name, len(d), "arrays")


def main():
    kw = dict(c=0.362, b=0.468, alpha=6, beta=1)

    # ---- stage fixtures on synthetic inputs -------------------------------------
    xa = synthetic_blurry_batch(1, 3, 96, 128, seed0=101)[0]
    xb = synthetic_blurry_batch(1, 1, 63, 95, seed0=202)[0]
    xc = synthetic_blurry_batch(2, 3, 64, 80, seed0=303)[0]
    stage_fixture(xa, "stages_A.npz")
    stage_fixture(xb, "stages_B.npz")
    stage_fixture(xc, "stages_C.npz")

    # ---- kernel grid -------------------------------------------------------------
    sig = np.array([0.3, 0.55, 1.0, 2.0, 4.0], np.float32)
    degs = np.array([0, 6, 24, 30, 42, 90, 174], np.float32)
    S, R, D = np.meshgrid(sig, sig, degs, indexing="ij")
    th = (T(D.reshape(-1, 1)) * np.pi / 180).float()
    kg = blur_estimation.create_gaussian_filter(th, T(S.reshape(-1, 1)), T(R.reshape(-1, 1)), ksize=25)
    np.savez_compressed(os.path.join(HERE, "kernel_grid.npz"), sigma=S.reshape(-1), rho=R.reshape(-1),
                        theta=npy(th)[:, 0], kernels=npy(kg)[:, 0])
    print("wrote kernel_grid.npz")

    # ---- peacock: the reference's own demo (config 1) --------------------------------
    img = np.asarray(Image.open(os.path.join(REF, "pictures/peacock_defocus.png")))[..., :3].astype(np.float32) / 255.0
    x = np.ascontiguousarray(np.moveaxis(img, 2, 0)[None])
    d = {}
    for method in ("fft", "direct"):
        out, its, imgs = run_pipeline(x, n_iter=3, method=method, **kw)
        pack_iterations(d, method, its)
        d["%s/out" % method] = out
        d["%s/it0_image_crop" % method] = imgs[0][..., 100:164, 200:264]
        d["%s/it1_image_crop" % method] = imgs[1][..., 100:164, 200:264]
    # ndarray in / ndarray out (HWC) through the functional API and the module
    nd = polyblur_deblurring(img, n_iter=3, **kw).astype(np.float32)
    assert nd.shape == img.shape and np.array_equal(np.moveaxis(nd, 2, 0)[None], d["fft/out"])
    d["ndarray_out_is_fft_out_hwc"] = np.array(1)
    np.savez_compressed(os.path.join(HERE, "pipeline_peacock.npz"), **d)
    print("wrote pipeline_peacock.npz")

    # ---- option variants on a crop ----------------------------------------------------
    xc = np.ascontiguousarray(x[..., 170:330, 240:464])          # (1,3,160,224)
    d = {"x": xc}
    variants = {
        "plain": {},
        "edgetaping": dict(edgetaping=True),
        "remove_halo": dict(remove_halo=True),
        "prefiltering": dict(prefiltering=True),
        "discard_saturation": dict(discard_saturation=True),
        "q1e-4": dict(q=1e-4),
        "all": dict(edgetaping=True, remove_halo=True, prefiltering=True),
    }
    for vname, opts in variants.items():
        for method in ("fft", "direct"):
            out, its, imgs = run_pipeline(xc, n_iter=3, method=method, **kw, **opts)
            pack_iterations(d, "%s/%s" % (vname, method), its)
            d["%s/%s/out" % (vname, method)] = out
    # saturated variant: scale up so that a visible fraction exceeds 0.99
    xs = np.clip(xc * 1.6, 0, 1).astype(np.float32)
    d["x_sat"] = xs
    for method in ("fft",):
        out, its, _ = run_pipeline(xs, n_iter=2, method=method, discard_saturation=True, **kw)
        pack_iterations(d, "sat/%s" % method, its)
        d["sat/%s/out" % method] = out
    # single-channel input
    xg = np.ascontiguousarray(xc[:, 1:2])
    out, its, _ = run_pipeline(xg, n_iter=2, method="fft", **kw)
    pack_iterations(d, "gray/fft", its)
    d["gray/fft/out"] = out
    # default parameters of the functional API and of the module (they differ)
    d["defaults/functional"] = npy(polyblur_deblurring(T(xc)))
    d["defaults/module"] = npy(PolyblurDeblurring()(T(xc)))
    d["module_n2"] = npy(PolyblurDeblurring()(T(xc), n_iter=2, alpha=6, beta=1))
    d["gray_ndarray_hw"] = polyblur_deblurring(xc[0, 1], n_iter=1, **kw).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "pipeline_variants.npz"), **d)
    print("wrote pipeline_variants.npz")

    # ---- strong blur: fft vs direct differ by 3.6e-2 at the border ---------------------
    ks = blur_estimation.create_gaussian_filter(torch.tensor([[np.deg2rad(30.0)]]).float(),
                                                torch.tensor([[3.5]]), torch.tensor([[1.5]]), ksize=25)
    xb = npy(filters.convolve2d(T(x[..., 100:356, 150:470].copy()), ks, method="fft")).clip(0, 1)
    d = {"x": xb}
    for method in ("fft", "direct"):
        out, its, _ = run_pipeline(xb, n_iter=3, method=method, **kw)
        pack_iterations(d, method, its)
        d["%s/out" % method] = out
    np.savez_compressed(os.path.join(HERE, "pipeline_strongblur.npz"), **d)
    print("wrote pipeline_strongblur.npz")

    # ---- batch (fft only: 'direct' is broken for B>1 in the reference) ------------------
    xb2, _ = synthetic_blurry_batch(3, 3, 120, 168, seed0=404)
    d = {"x": xb2}
    out, its, _ = run_pipeline(xb2, n_iter=3, method="fft", **kw)
    pack_iterations(d, "fft", its)
    d["fft/out"] = out
    # the same images one by one (B == 1 semantics) -- must equal the batched run
    d["fft/out_single"] = np.concatenate([npy(polyblur_deblurring(T(xb2[i:i + 1]), n_iter=3, method="fft", **kw))
                                          for i in range(3)])
    np.savez_compressed(os.path.join(HERE, "pipeline_batch.npz"), **d)
    print("wrote pipeline_batch.npz")

    # ---- fp16-rounded inputs, oracle in fp32 -------------------------------
    xh = xc.astype(np.float16).astype(np.float32)
    d = {"x": xh}
    out, its, _ = run_pipeline(xh, n_iter=3, method="fft", **kw)
    pack_iterations(d, "fft", its)
    d["fft/out"] = out
    np.savez_compressed(os.path.join(HERE, "pipeline_fp16in.npz"), **d)
    print("wrote pipeline_fp16in.npz")


if __name__ == "__main__":
    main()
