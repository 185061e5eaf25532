"""Host-side driver of the C ABI: one `Engine` (= one pb_ctx) per GPU.

Everything here is plumbing: device buffers, option marshalling, numpy <-> device copies.
The arithmetic all happens in libpolyblur_hip.so.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _capi as capi
from ._capi import PolyblurHipError

_DT = {np.dtype(np.float32): capi.PB_F32, np.dtype(np.float16): capi.PB_F16, np.dtype(np.uint8): capi.PB_U8}


class DeviceBuffer:
    """A hipMalloc'ed buffer owned by an Engine (used when the caller hands us host data)."""

    def __init__(self, engine: "Engine", nbytes: int):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        engine._check(engine.lib.pb_malloc(engine.ctx, C.byref(p), C.c_size_t(max(self.nbytes, 1))))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.pb_memcpy_h2d(self.engine.ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.pb_memcpy_d2h(self.engine.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.engine.lib.pb_free(self.engine.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = capi.load_library()
        ctx = C.c_void_p()
        rc = self.lib.pb_create(C.byref(ctx), int(device), C.c_void_p(stream or 0))
        if rc != 0:
            raise PolyblurHipError("pb_create(device=%d) failed with %s: no usable MI355X/HIP device "
                                   "(the engine has no CPU fallback)" % (device, capi.STATUS.get(rc, rc)))
        self.ctx = ctx
        self.device = device
        self._pool = {}

    # ---- infrastructure ------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.pb_last_error_string(self.ctx)
            raise PolyblurHipError("%s: %s" % (capi.STATUS.get(rc, rc), msg.decode() if msg else ""))

    def close(self):
        if self.ctx:
            for b in self._pool.values():
                b.free()
            self._pool.clear()
            self.lib.pb_destroy(self.ctx)
            self.ctx = None

    def set_stream(self, stream: int):
        self._check(self.lib.pb_set_stream(self.ctx, C.c_void_p(stream or 0)))

    def synchronize(self):
        self._check(self.lib.pb_synchronize(self.ctx))

    def set_dense_eval(self, mode: str = "auto", min_phases: int = 16):
        """How dense (non rank-1) kernels are evaluated: 'stencil' = always the 2-D stencil body; 'auto' = kernels with
        at least `min_phases` live stencil phases take the tile-spectrum body (pb_set_dense_eval)."""
        m = {"stencil": capi.PB_DENSE_STENCIL, "auto": capi.PB_DENSE_AUTO}[mode]
        self._check(self.lib.pb_set_dense_eval(self.ctx, m, int(min_phases)))

    def body_selection(self, B: int, iteration: int = -1) -> np.ndarray:
        """(B, 6) int32: per image {tile-spectrum body, halo class, strip, one-pass polynomial, halo x, halo y} of iteration
        `iteration` of the most recent polyblur call (-1: of the most recent estimation / reblurring pass; pb_body_selection)."""
        out = np.zeros((B, 6), np.int32)
        self._check(self.lib.pb_body_selection(self.ctx, int(iteration), out.ctypes.data_as(capi.C.POINTER(capi.C.c_int)), int(B)))
        return out

    def workspace_bytes(self) -> int:
        return int(self.lib.pb_workspace_bytes(self.ctx))

    def buffer(self, name: str, nbytes: int) -> DeviceBuffer:
        """A named, reusable device buffer (grows on demand)."""
        b = self._pool.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = DeviceBuffer(self, nbytes)
            self._pool[name] = b
        return b

    def to_device(self, name: str, arr: np.ndarray) -> DeviceBuffer:
        return self.buffer(name, arr.nbytes).upload(arr)

    @staticmethod
    def make_options(n_iter=1, c=0.352, b=0.768, alpha=2, beta=3, sigma_r=0.8, sigma_s=2.0, q=0.0, n_angles=6,
                     n_interpolated_angles=30, remove_halo=False, edgetaping=False, prefilter=capi.PB_PREFILTER_NONE,
                     discard_saturation=False, boundary=capi.PB_WRAP, support=capi.PB_SUPPORT_FULL,
                     force_theta_deg=-1.0, ker_size=capi.PB_KSIZE, separable_approx=False, half_temporaries=False) -> capi.pb_options:
        o = capi.pb_options()
        o.n_iter = int(n_iter); o.c = float(c); o.b = float(b); o.alpha = float(alpha); o.beta = float(beta)
        o.sigma_s = float(sigma_s); o.sigma_r = float(sigma_r); o.q = float(q); o.n_angles = int(n_angles)
        o.n_interpolated_angles = int(n_interpolated_angles); o.remove_halo = int(bool(remove_halo))
        o.edgetaping = int(bool(edgetaping)); o.prefilter = int(prefilter)
        o.discard_saturation = int(bool(discard_saturation)); o.boundary = int(boundary); o.support = int(support)
        o.force_theta_deg = float(force_theta_deg)
        o.ker_size = int(ker_size)
        o.separable_approx = int(bool(separable_approx))
        o.half_temporaries = int(bool(half_temporaries))
        return o

    # ---- raw-pointer entry points (device pointers as ints) --------------------------------
    def polyblur_ptr(self, in_ptr: int, out_ptr: int, dtype: int, shape, opts: capi.pb_options, want_info=False):
        B, Cc, H, W = (int(v) for v in shape)
        info = None
        ip = None
        if want_info and opts.n_iter > 0:
            info = np.zeros((opts.n_iter, B), dtype=capi.INFO_DTYPE)
            ip = info.ctypes.data
        self._check(self.lib.pb_polyblur_batch(self.ctx, in_ptr, out_ptr, dtype, B, Cc, H, W, C.byref(opts), ip))
        return info

    # ---- numpy conveniences (host arrays in, host arrays out) -------------------------------
    def polyblur(self, x: np.ndarray, opts: capi.pb_options, want_info=False):
        x = np.ascontiguousarray(x)
        if x.dtype not in _DT:
            x = x.astype(np.float32)
        din = self.to_device("np.in", x)
        dout = self.buffer("np.out", x.nbytes)
        info = self.polyblur_ptr(din.ptr, dout.ptr, _DT[x.dtype], x.shape, opts, want_info)
        out = dout.download(x.shape, x.dtype)
        return (out, info) if want_info else out

    def polyblur_u8_hwc(self, imgs: np.ndarray, opts: capi.pb_options, want_info=False):
        """(B,H,W,C) uint8 host images -> deblurred (B,H,W,C) uint8: the bytes go up as they are, are planarised,
        deblurred (8-bit load / store fused into the first / last kernels) and re-interleaved on the device."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        B, H, W, Cc = imgs.shape
        hwc = self.to_device("np.u8.hwc", imgs)
        chw = self.buffer("np.u8.chw", imgs.nbytes)
        res = self.buffer("np.u8.out", imgs.nbytes)
        self._check(self.lib.pb_u8_deinterleave(self.ctx, hwc.ptr, chw.ptr, B, Cc, H, W))
        info = self.polyblur_ptr(chw.ptr, res.ptr, capi.PB_U8, (B, Cc, H, W), opts, want_info)
        self._check(self.lib.pb_u8_interleave(self.ctx, res.ptr, hwc.ptr, B, Cc, H, W))
        out = hwc.download(imgs.shape, np.uint8)
        return (out, info) if want_info else out

    def info_buffer(self, name: str, B: int) -> DeviceBuffer:
        return self.buffer(name, B * capi.INFO_DTYPE.itemsize)

    def make_kernels(self, sigma, rho, theta_rad, support=capi.PB_SUPPORT_FULL, name="np.info") -> DeviceBuffer:
        s = np.ascontiguousarray(sigma, np.float32).reshape(-1)
        r = np.ascontiguousarray(rho, np.float32).reshape(-1)
        t = np.ascontiguousarray(theta_rad, np.float32).reshape(-1)
        B = s.size
        buf = self.info_buffer(name, B)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.pb_make_kernels(self.ctx, B, s.ctypes.data_as(fp), r.ctypes.data_as(fp),
                                             t.ctypes.data_as(fp), int(support), buf.ptr))
        return buf

    def set_kernels(self, taps, support=capi.PB_SUPPORT_FULL, name="np.info") -> DeviceBuffer:
        k = np.ascontiguousarray(taps, np.float32).reshape(-1, capi.PB_KSIZE, capi.PB_KSIZE)
        buf = self.info_buffer(name, k.shape[0])
        self._check(self.lib.pb_set_kernels(self.ctx, k.shape[0], k.ctypes.data_as(C.POINTER(C.c_float)),
                                            int(support), buf.ptr))
        return buf

    def read_info(self, buf: DeviceBuffer, B: int) -> np.ndarray:
        self.synchronize()
        return buf.download((B,), capi.INFO_DTYPE)

    def estimate_blur(self, x: np.ndarray, opts: capi.pb_options) -> np.ndarray:
        x = np.ascontiguousarray(x)
        B, Cc, H, W = x.shape
        din = self.to_device("np.in", x)
        buf = self.info_buffer("np.info", B)
        self._check(self.lib.pb_estimate_blur(self.ctx, din.ptr, _DT[x.dtype], B, Cc, H, W, C.byref(opts), buf.ptr))
        return self.read_info(buf, B)

    def fourier_gradients(self, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        H, W = x.shape[-2:]
        P = int(np.prod(x.shape[:-2])) if x.ndim > 2 else 1
        din = self.to_device("np.in", x)
        gx = self.buffer("np.gx", x.nbytes)
        gy = self.buffer("np.gy", x.nbytes)
        self._check(self.lib.pb_fourier_gradients(self.ctx, din.ptr, P, H, W, gx.ptr, gy.ptr))
        self.synchronize()
        return gx.download(x.shape, np.float32), gy.download(x.shape, np.float32)

    def inverse_filter(self, x: np.ndarray, info: DeviceBuffer, alpha, beta, boundary=capi.PB_WRAP, edgetaping=False,
                       remove_halo=False, grad0=None) -> np.ndarray:
        x = np.ascontiguousarray(x)
        B, Cc, H, W = x.shape
        din = self.to_device("np.in", x)
        dout = self.buffer("np.out", x.nbytes)
        g0x = g0y = None
        if remove_halo:
            g0x = self.to_device("np.g0x", np.ascontiguousarray(grad0[0], np.float32)).ptr
            g0y = self.to_device("np.g0y", np.ascontiguousarray(grad0[1], np.float32)).ptr
        self._check(self.lib.pb_inverse_filter(self.ctx, din.ptr, dout.ptr, _DT[x.dtype], B, Cc, H, W, info.ptr,
                                               float(alpha), float(beta), int(boundary), int(bool(edgetaping)),
                                               int(bool(remove_halo)), g0x, g0y))
        self.synchronize()
        return dout.download(x.shape, x.dtype)

    def convolve2d(self, xp: np.ndarray, info: DeviceBuffer, boundary=capi.PB_WRAP) -> np.ndarray:
        xp = np.ascontiguousarray(xp, np.float32)
        B, Cc, Hp, Wp = xp.shape
        din = self.to_device("np.in", xp)
        dout = self.buffer("np.out", xp.nbytes)
        self._check(self.lib.pb_convolve2d(self.ctx, din.ptr, dout.ptr, B, Cc, Hp, Wp, info.ptr, int(boundary)))
        self.synchronize()
        return dout.download(xp.shape, np.float32)

    def edgetaper(self, xp: np.ndarray, info: DeviceBuffer, boundary=capi.PB_WRAP) -> np.ndarray:
        xp = np.ascontiguousarray(xp, np.float32)
        B, Cc, Hp, Wp = xp.shape
        din = self.to_device("np.in", xp)
        dout = self.buffer("np.out", xp.nbytes)
        self._check(self.lib.pb_edgetaper(self.ctx, din.ptr, dout.ptr, B, Cc, Hp, Wp, info.ptr, int(boundary)))
        self.synchronize()
        return dout.download(xp.shape, np.float32)

    def halo_mask(self, x, y, g0x, g0y) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        B, Cc, H, W = x.shape
        dx = self.to_device("np.in", x)
        dy = self.to_device("np.y", np.ascontiguousarray(y, np.float32))
        dgx = self.to_device("np.g0x", np.ascontiguousarray(g0x, np.float32))
        dgy = self.to_device("np.g0y", np.ascontiguousarray(g0y, np.float32))
        dout = self.buffer("np.out", x.nbytes)
        self._check(self.lib.pb_halo_mask(self.ctx, dx.ptr, dy.ptr, dgx.ptr, dgy.ptr, dout.ptr, B, Cc, H, W))
        self.synchronize()
        return dout.download(x.shape, np.float32)

    def dt_recursive_filter(self, x: np.ndarray, sigma_s=60.0, sigma_r=0.4, num_iterations=3, joint=None) -> np.ndarray:
        x = np.ascontiguousarray(x)
        B, Cc, H, W = x.shape
        din = self.to_device("np.in", x)
        dj = self.to_device("np.joint", np.ascontiguousarray(joint, x.dtype)).ptr if joint is not None else None
        dout = self.buffer("np.out", x.nbytes)
        self._check(self.lib.pb_dt_recursive_filter(self.ctx, din.ptr, dj, dout.ptr, _DT[x.dtype], B, Cc, H, W,
                                                    float(sigma_s), float(sigma_r), int(num_iterations)))
        self.synchronize()
        return dout.download(x.shape, x.dtype)

    def dt_normalized_convolution(self, x: np.ndarray, sigma_s=60.0, sigma_r=0.4, num_iterations=3) -> np.ndarray:
        x = np.ascontiguousarray(x)
        B, Cc, H, W = x.shape
        din = self.to_device("np.in", x)
        dout = self.buffer("np.out", x.nbytes)
        self._check(self.lib.pb_dt_normalized_convolution(self.ctx, din.ptr, dout.ptr, _DT[x.dtype], B, Cc, H, W,
                                                          float(sigma_s), float(sigma_r), int(num_iterations)))
        self.synchronize()
        return dout.download(x.shape, x.dtype)

    def bilateral5(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x)
        B, Cc, H, W = x.shape
        din = self.to_device("np.in", x)
        dout = self.buffer("np.out", x.nbytes)
        self._check(self.lib.pb_bilateral5(self.ctx, din.ptr, dout.ptr, _DT[x.dtype], B, Cc, H, W))
        self.synchronize()
        return dout.download(x.shape, x.dtype)

    def profile_begin(self):
        self._check(self.lib.pb_profile_begin(self.ctx))

    def profile_end(self):
        """-> {tag: (total_ms, launches)} for the launches issued since profile_begin()."""
        ms = (C.c_float * len(capi.PROF_TAGS))()
        cnt = (C.c_int * len(capi.PROF_TAGS))()
        self._check(self.lib.pb_profile_end(self.ctx, ms, cnt))
        return {t: (float(ms[i]), int(cnt[i])) for i, t in enumerate(capi.PROF_TAGS)}

    def time_inner_loop(self, in_ptr: int, out_ptr: int, dtype: int, shape, info_ptr: int, alpha, beta,
                        boundary=capi.PB_WRAP, reps=10) -> float:
        B, Cc, H, W = (int(v) for v in shape)
        ms = C.c_float(0)
        self._check(self.lib.pb_time_inner_loop(self.ctx, in_ptr, out_ptr, dtype, B, Cc, H, W, info_ptr, float(alpha),
                                                float(beta), int(boundary), int(reps), C.byref(ms)))
        return float(ms.value)


_tls = threading.local()


class _ThreadEngines:
    """The engines of one host thread; closed (contexts destroyed, scratch freed) when the thread ends."""

    def __init__(self):
        self.by_device = {}

    def __del__(self):
        for e in self.by_device.values():
            try:
                e.close()
            except Exception:
                pass


def get_engine(device: int = 0) -> Engine:
    """The calling thread's Engine for a GPU (created on first use).  A pb_ctx is not thread-safe and owns its
    scratch buffers, so every host thread gets its own -- kept in thread-local storage, so that a thread that ends takes
    its contexts (and their HBM) with it and a recycled thread id never inherits another thread's engine; within a
    thread, calls on different streams are ordered by pb_set_stream (the new stream waits for the work queued on the
    previous one)."""
    box = getattr(_tls, "engines", None)
    if box is None:
        box = _tls.engines = _ThreadEngines()
    e = box.by_device.get(int(device))
    if e is None or e.ctx is None:
        e = Engine(device)
        box.by_device[int(device)] = e
    return e
