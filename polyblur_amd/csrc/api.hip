// C ABI of libpolyblur_hip.so (include/polyblur_hip.h): context, scratch memory, and the
// drivers that chain the kernels of conv.hip / estimate.hip / filters.hip on one stream.
// Mirrors polyblur_deblurring's main loop (reference deblurring.py:58-96) and
// inverse_filtering_rank3 (deblurring.py:211-239).
#include <cstdlib>
#include <cstring>

#include "common.h"

// (g_dtype: the type of the gradient planes gx, gy, ox -- PB_F32, or PB_F16 inside an fp16 call of the pipeline)
int pb_grad_energy(pb_ctx *ctx, const void *gx, const void *gy, float *nM, int P, long HW, int g_dtype = PB_F32);
int pb_halo_apply(pb_ctx *ctx, const void *x, int x_dtype, int x_pitch, long x_plane, const float *y, const void *gx,
                  const void *gy, const void *ox, const float *nM, void *out, int out_dtype, int P, int H, int W,
                  int clamp01, const void *recomb_cur = nullptr, int recomb_cur_dtype = 0, const float *recomb_smooth = nullptr,
                  int g_dtype = PB_F32);
int pb_recombine(pb_ctx *ctx, const float *y, const void *cur, int cur_dtype, const float *smooth, void *out, int out_dtype,
                 long n);
int pb_bilateral5_impl(pb_ctx *ctx, const void *in, int in_dtype, void *out, int out_dtype, int P, int H, int W);
int pb_dt_filter_impl(pb_ctx *ctx, const void *in, const void *joint, int dtype, float *out, int B, int C, int H, int W,
                      float sigma_s, float sigma_r, int num_iterations);
int pb_nc_filter_impl(pb_ctx *ctx, const void *in, int dtype, float *out, int B, int C, int H, int W, float sigma_s,
                      float sigma_r, int num_iterations);
int pb_convert_from_float(pb_ctx *ctx, const float *in, void *out, int dtype, long n);
int pb_u8_layout(pb_ctx *ctx, const unsigned char *in, unsigned char *out, int B, int C, int H, int W, int to_planar);
int pb_convert_to_float(pb_ctx *ctx, const void *in, int dtype, float *out, long n);
int pb_extract_patches_impl(pb_ctx *ctx, const void *img, void *patches, int dtype, int B, int C, int H, int W, int ph, int pw,
                            int step_h, int step_w, int n_j, int pad_top, int pad_left, int first, int count);
int pb_overlap_add_impl(pb_ctx *ctx, const void *patches, void *out, int dtype, int B, int C, int H, int W, int ph, int pw,
                        int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left, const float *win_y,
                        const float *win_x);

#ifndef PB_EXPERIMENTAL
// (conv_xt.hip -- both 1-D passes of the x-t approximation in one launch -- is a measured experiment of the --experimental
// build: slower than the exact path it approximates.  The default build runs method='direct_separable' as two launches
// of the general body over the two sparse records.)
int pb_launch_conv_xt(pb_ctx *, const ConvPass &) { return PB_ERR_UNSUPPORTED; }
#endif

int pb_fail(pb_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

void *pb_scratch(pb_ctx *ctx, const char *name, size_t bytes) {
    ScratchBuf &b = ctx->scratch[name];
    if (b.bytes >= bytes && b.p) return b.p;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = (bytes + 255) & ~(size_t)255;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        pb_fail(ctx, PB_ERR_NOMEM, "scratch '%s': hipMalloc(%zu) failed", name, want);
        return nullptr;
    }
    b.bytes = want;
    return b.p;
}

static size_t dsize(int dtype) { return dtype == PB_F16 ? 2 : (dtype == PB_U8 ? 1 : 4); }
static int pitch4(int w) { return (w + 3) & ~3; }

// Every environment knob of the library, read ONCE per context (pb_create) into pb_ctx -- no kernel launcher looks at the
// environment.  They exist to compare forms on the same box (bench.py's context entries, tools/), not to configure a
// deployment: the defaults are the product.
//   PB_DENSE_EVAL=stencil|<n>   dense kernels never take the tile-spectrum body | from n live stencil phases on (16)
//   PB_FFT_BODY=wg|wave         which form of the tile-spectrum body runs three-step passes (wave; fp16 temporaries --
//                               pb_options.half_temporaries -- always take the workgroup form, the only one built for them)
//   PB_POLY1=0..3               one-pass polynomial: 0 never, 1 4-sample halo class only, 2 + 64 x 64 windows with the
//                               composite's halos, 3 (default) + 128 x 128 windows
//   PB_POLY_GAIN, PB_POLY_MIN_AREA, PB_POLY_COST128, PB_POLY_MIN_PAIRS128   cost model of the forms (common.h: 0.7, 768, 8, 1)
//   PB_ZERO_RING_ASIDE=0        ... its first two ring steps behind the window pass instead of beside it (side stream)
//   PB_ZERO_RING_MIN_PAIRS=<n>  ... the ring form only for images of at least n three-step window pairs (4096)
//   PB_ZERO_RING=0              method='direct' keeps three Horner steps over the whole image
//   PB_TAPER_RING=0             the second and third blend of an edgetaper over the whole plane, not over the border ring
//   PB_POLY_PADDED=0            the polynomial after an edgetaper keeps three Horner steps
//   PB_POLY_ALWAYS=0            never PolySpec.always (issue every launch the records might need)
//   PB_EST_GRAY_ROWS=0|1|2      gray + range + row transform in one launch: never | fp32 lines up to 4096 | any line in LDS
//   PB_EST_LEAN=0               the parameter kernel forms the whole record before the spectra (no short chain)
//   PB_DT_ROWS_REG=0            the domain-transform row pass through global memory instead of registers
//   PB_DT_COLS_STRIP=0          the domain-transform column pass as two sweeps through global memory instead of strips
//   PB_FFT_EXT_RADIX=0          greedy transform plans only (radices up to 16)
//   PB_FFT_FIRST / _ROWS=0|r    the column / row transform's first and last stage: the greedy plan's order, or radix r (default: by line length)
//   PB_FFT_LOGNB, PB_WAVE_MIN_JOBS   shapes of the column-transform / wave-body launches
//   PB_COLS_FIXED=0, PB_ROWS_FIXED=0   the line transforms always by the run-time-plan kernels (estimate.hip), also where
//                               lines_fixed.hip holds the plan (the tests' bit-identity reference)
//   PB_XT=2                     the x-t approximation through two launches of the general body
//   PB_STRIP, PB_STRIP_SEG, PB_EST_OVERLAP   measured experiments: read in --experimental builds only
// (retired in round 6, their alternatives measured and dropped in NOTEBOOK.md: PB_ROWS_NT, PB_COLS_WIDE, PB_MAIN_STREAM_BODY,
// PB_SIDE_STREAM, PB_SIDE_MIN_TILES -- the fields keep their defaults)
static void pb_read_knobs(pb_ctx *ctx) {
    auto geti = [](const char *n, int &v) { if (const char *e = getenv(n)) v = atoi(e); };
    auto getl = [](const char *n, long &v) { if (const char *e = getenv(n)) v = atol(e); };
    auto getf = [](const char *n, float &v) { if (const char *e = getenv(n)) v = (float)atof(e); };
    if (const char *e = getenv("PB_DENSE_EVAL")) {
        if (e[0] == 's') ctx->fft_min_phases = -1;
        else if (e[0] >= '0' && e[0] <= '9') ctx->fft_min_phases = atoi(e);
    }
    if (const char *e = getenv("PB_FFT_BODY")) ctx->fft_wave = (e[0] == 'w' && e[1] == 'g') ? 0 : 1;
    if (const char *e = getenv("PB_XT")) ctx->xt_two_launch = e[0] == '2';
#ifdef PB_EXPERIMENTAL
    geti("PB_STRIP", ctx->strip_mode); geti("PB_STRIP_SEG", ctx->strip_seg);
#endif
    geti("PB_EST_GRAY_ROWS", ctx->est_gray_rows); geti("PB_EST_LEAN", ctx->est_lean); geti("PB_DT_ROWS_REG", ctx->dt_rows_reg); geti("PB_DT_COLS_STRIP", ctx->dt_cols_strip); geti("PB_DT_COLS_COOP", ctx->dt_cols_coop);
    geti("PB_FFT_EXT_RADIX", ctx->fft_ext_radix); geti("PB_FFT_FIRST", ctx->fft_first); geti("PB_FFT_FIRST_ROWS", ctx->fft_first_rows); geti("PB_FFT_LOGNB", ctx->fft_lognb); geti("PB_COLS_FIXED", ctx->cols_fixed); geti("PB_ROWS_FIXED", ctx->rows_fixed);
    getl("PB_WAVE_MIN_JOBS", ctx->wave_min_jobs);
    geti("PB_POLY1", ctx->poly_mode); getf("PB_POLY_GAIN", ctx->poly_gain); geti("PB_POLY_MIN_AREA", ctx->poly_min_area);
    getf("PB_POLY_COST128", ctx->poly_cost128); getl("PB_POLY_MIN_PAIRS128", ctx->poly_min_pairs128);
    geti("PB_POLY_ALWAYS", ctx->poly_always); geti("PB_POLY_PADDED", ctx->poly_padded); geti("PB_TAPER_RING", ctx->taper_ring); geti("PB_ZERO_RING", ctx->zero_ring); getl("PB_ZERO_RING_MIN_PAIRS", ctx->zero_ring_min_pairs); geti("PB_ZERO_RING_ASIDE", ctx->zero_ring_aside);
}

extern "C" {

int pb_version(void) { return PB_VERSION; }

void pb_default_options(pb_options *o) {
    // deblurring.py:23-25
    memset(o, 0, sizeof(*o));
    o->n_iter = 1; o->c = 0.352f; o->b = 0.768f; o->alpha = 2.f; o->beta = 3.f;
    o->sigma_r = 0.8f; o->sigma_s = 2.0f; o->q = 0.f; o->n_angles = 6; o->n_interpolated_angles = 30;
    o->remove_halo = 0; o->edgetaping = 0; o->prefilter = PB_PREFILTER_NONE; o->discard_saturation = 0;
    o->boundary = PB_WRAP; o->support = PB_SUPPORT_FULL; o->force_theta_deg = -1.f; o->ker_size = PB_KSIZE;
}

int pb_create(pb_ctx **out, int device, void *stream) {
    if (!out) return PB_ERR_BADARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PB_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return PB_ERR_HIP;
    pb_ctx *ctx = new pb_ctx();
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream);
    pb_read_knobs(ctx);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming) != hipSuccess) { delete ctx; return PB_ERR_HIP; }
    {
        if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
            ctx->aux = nullptr;                                  // (the engine works without it)
        }
    }
    *out = ctx;
    return PB_OK;
}

int pb_set_dense_eval(pb_ctx *ctx, int mode, int min_phases) {
    if (!ctx || (mode != PB_DENSE_STENCIL && mode != PB_DENSE_AUTO) || min_phases < 0 || min_phases > PB_MAX_PHASES + 1)
        return PB_ERR_BADARG;
    ctx->fft_min_phases = mode == PB_DENSE_STENCIL ? -1 : min_phases;
    pb_forget_records(ctx, nullptr, 0);
    return PB_OK;
}

int pb_destroy(pb_ctx *ctx) {
    if (!ctx) return PB_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->scratch) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto &kv : ctx->plans) {
        FftPlan &p = kv.second;
        if (p.tw) (void)hipFree(p.tw);
        if (p.drev) (void)hipFree(p.drev);
        if (p.chirp) (void)hipFree(p.chirp);
        if (p.bfilt_rev) (void)hipFree(p.bfilt_rev);
        if (p.dnat) (void)hipFree(p.dnat);
    }
    if (ctx->interp_w) (void)hipFree(ctx->interp_w);
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->evpool) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    delete ctx;
    return PB_OK;
}

int pb_set_stream(pb_ctx *ctx, void *stream) {
    if (!ctx) return PB_ERR_BADARG;
    hipStream_t next = static_cast<hipStream_t>(stream);
    if (next != ctx->stream) {
        // the context's scratch buffers may still be in use by work queued on the old stream: order the new
        // stream behind it (device-side dependency, no host wait)
        PB_HIP(hipSetDevice(ctx->device));
        if (hipEventRecord(ctx->ev_switch, ctx->stream) != hipSuccess || hipStreamWaitEvent(next, ctx->ev_switch, 0) != hipSuccess) {
            // (the old stream may have been destroyed by its owner: nothing can be ordered behind it any more -- wait
            // for the device instead and adopt the new stream all the same)
            (void)hipGetLastError();
            PB_HIP(hipDeviceSynchronize());
        }
        ctx->stream = next;
    }
    return PB_OK;
}

int pb_synchronize(pb_ctx *ctx) {
    if (!ctx) return PB_ERR_BADARG;
    PB_HIP(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

const char *pb_last_error_string(pb_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

size_t pb_workspace_bytes(pb_ctx *ctx) {
    size_t t = 0;
    if (ctx) for (auto &kv : ctx->scratch) t += kv.second.bytes;
    return t;
}

int pb_malloc(pb_ctx *ctx, void **dptr, size_t bytes) {
    if (!ctx || !dptr) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) return pb_fail(ctx, PB_ERR_NOMEM, "pb_malloc(%zu) failed", bytes);
    return PB_OK;
}
int pb_free(pb_ctx *ctx, void *dptr) {
    if (!ctx) return PB_ERR_BADARG;
    {
        // what the context has cached about records is a hint and goes wholesale; which caller-supplied record sets hold taps that
        // are not point-symmetric is not (common.h: FlipSet): only the sets inside the allocation being freed are dropped
        auto keep = std::move(ctx->flip_sets);
        pb_forget_records(ctx, nullptr, 0);
        hipDeviceptr_t base = nullptr; size_t size = 0;
        if (dptr && hipMemGetAddressRange(&base, &size, dptr) == hipSuccess) {
            const char *lo = static_cast<const char *>(static_cast<void *>(base)), *hi = lo + size;
            for (auto it = keep.begin(); it != keep.end();) {
                const char *a = static_cast<const char *>(it->first);
                if (a >= lo && a < hi) it = keep.erase(it); else ++it;
            }
        } else {
            (void)hipGetLastError();
            keep.erase(dptr);
        }
        ctx->flip_sets = std::move(keep);
    }
    PB_HIP(hipStreamSynchronize(ctx->stream));
    PB_HIP(hipFree(dptr));
    return PB_OK;
}
int pb_memcpy_h2d(pb_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return PB_ERR_BADARG;
    pb_forget_range(ctx, dst, bytes);                     // (the copy may overwrite records the context has cached facts about)
    PB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}
int pb_memcpy_d2h(pb_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return PB_ERR_BADARG;
    PB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// inverse filtering (deblurring.py:211-239)
// ---------------------------------------------------------------------------------------------
namespace {

struct Geometry {
    int B, C, H, W, P, pad, Hp, Wp, pp;     // pad = ker_size / 2; pp = pitch of padded fp32 planes
    long pplane;                        // elements per padded plane
    long HW;
    // method='direct_separable': records of the two 1-D passes that stand in for every reblurring, and the plane between them
    const pb_blur_info *sep1 = nullptr, *sep2 = nullptr;
    float *sep_u = nullptr;
    // pb_options.half_temporaries: the two Horner temporaries are stored as fp16 (fp32 accumulation, fp32 x operand)
    void *t1h = nullptr, *t2h = nullptr;
    // ker_size above 25: the taps on the ker_size grid (conv_big.hip), rebuilt after every estimation
    const float *big_taps = nullptr;
    int big_ksize = 0;
    // the records are point-symmetric Gaussians the estimation of THIS call builds on an odd ker_size grid (PolySpec.always)
    bool est_gaussians = false;
    // ... and every one of them takes the window form of a single pass (PolySpec.always == 2): the edgetaper's three blends
    // issue the wave body's launch and nothing else
    bool taper_windows = false;
};

Geometry geometry(int B, int C, int H, int W, int pad = PB_KRAD) {
    Geometry g;
    g.B = B; g.C = C; g.H = H; g.W = W; g.P = B * C; g.pad = pad;
    g.Hp = H + 2 * pad; g.Wp = W + 2 * pad; g.pp = pitch4(g.Wp);
    g.pplane = (long)g.Hp * g.pp;
    g.HW = (long)H * W;
    return g;
}

ConvPass base_pass(const Geometry &g, const pb_blur_info *info, int boundary) {
    ConvPass p;
    memset(&p, 0, sizeof(p));
    p.H = g.H; p.W = g.W; p.pad = g.pad; p.C = g.C; p.P = g.P; p.info = info; p.boundary = boundary;
    p.scale = 1.f; p.coef = 0.f; p.epilogue = EPI_HORNER; p.clamp01 = 0;
    return p;
}
void set_in_virtual(ConvPass &p, const Geometry &g, const void *ptr, int dtype) {
    p.in = ptr; p.in_kind = SRC_VIRTUAL; p.in_dtype = dtype; p.in_pitch = g.W; p.in_plane = g.HW;
}
void set_in_padded(ConvPass &p, const Geometry &g, const void *ptr, int dtype = PB_F32) {
    p.in = ptr; p.in_kind = SRC_PADDED; p.in_dtype = dtype; p.in_pitch = g.pp; p.in_plane = g.pplane;
}
void set_x_virtual(ConvPass &p, const Geometry &g, const void *ptr, int dtype) {
    p.x = ptr; p.x_kind = SRC_VIRTUAL; p.x_dtype = dtype; p.x_pitch = g.W; p.x_plane = g.HW;
}
void set_x_padded(ConvPass &p, const Geometry &g, const float *ptr) {
    p.x = ptr; p.x_kind = SRC_PADDED; p.x_dtype = PB_F32; p.x_pitch = g.pp; p.x_plane = g.pplane;
}
void set_out_padded(ConvPass &p, const Geometry &g, void *ptr, int dtype = PB_F32) {
    p.out = ptr; p.out_kind = OUT_PADDED; p.out_dtype = dtype; p.out_pitch = g.pp; p.out_plane = g.pplane;
}
void set_out_interior(ConvPass &p, const Geometry &g, void *ptr, int dtype) {
    p.out = ptr; p.out_kind = OUT_INTERIOR; p.out_dtype = dtype; p.out_pitch = g.W; p.out_plane = g.HW;
}

// Three edgetaper blends on the padded domain (edgetaper.py:26-33).  src is an un-padded image
// (virtual replicate pad).  Returns the padded fp32 result in *result (one of the two scratch planes).
int run_edgetaper(pb_ctx *ctx, const Geometry &g, const void *src, int src_dtype, const pb_blur_info *info, int boundary,
                  float *pa, float *pb, float **result) {
    ConvPass p = base_pass(g, info, boundary);
    p.epilogue = EPI_TAPER;
    set_in_virtual(p, g, src, src_dtype);
    set_x_virtual(p, g, src, src_dtype);
    set_out_padded(p, g, pa);
    if (g.big_taps) {
        // (a kernel beyond the 25 x 25 record: the three blends through conv_big.hip's pass, weights from its own autocorrelations)
        int rcb = pb_launch_conv_big(ctx, p, g.big_taps, g.big_ksize);
        if (rcb) return rcb;
        set_in_padded(p, g, pa); set_x_padded(p, g, pa); set_out_padded(p, g, pb);
        rcb = pb_launch_conv_big(ctx, p, g.big_taps, g.big_ksize);
        if (rcb) return rcb;
        set_in_padded(p, g, pb); set_x_padded(p, g, pb); set_out_padded(p, g, pa);
        rcb = pb_launch_conv_big(ctx, p, g.big_taps, g.big_ksize);
        *result = pa;
        return rcb;
    }
    // (the spec the estimation built these records' spectra under: every kernel on the window form, one launch per blend)
    struct SpecScope { pb_ctx *c; ~SpecScope() { c->poly_want = no_poly(); } } scope{ctx};
    if (g.taper_windows) { ctx->poly_want = no_poly(); ctx->poly_want.always = 2; }
    int rc = pb_launch_conv(ctx, p);
    if (rc) return rc;
    // (window form for every image: the second and the third blend only touch the ring of window pairs on which alpha < 1
    // can reach them -- the rest of `pa` keeps the first blend's copy of the image, which is what three blends with alpha = 1
    // leave there; `pb` is only ever read inside the second blend's ring)
    const int ring2 = g.taper_windows && ctx->taper_ring ? 5 : 0, ring3 = g.taper_windows && ctx->taper_ring ? 4 : 0;
    set_in_padded(p, g, pa); set_x_padded(p, g, pa); set_out_padded(p, g, pb);
    p.ring = ring2;
    rc = pb_launch_conv(ctx, p);
    if (rc) return rc;
    set_in_padded(p, g, pb); set_x_padded(p, g, pb); set_out_padded(p, g, pa);
    p.ring = ring3;
    rc = pb_launch_conv(ctx, p);
    *result = pa;
    return rc;
}

// The three Horner steps of y = a3 K^3 x + a2 K^2 x + a1 K x + beta x (deblurring.py:122-138).
void make_steps(const Geometry &g, const void *xsrc, int x_dtype, const float *xpadded, const pb_blur_info *info, float alpha,
                float beta, int boundary, float *t1, float *t2, void *dst, int dst_dtype, int clamp01, ConvPass *steps) {
    const float a3 = alpha / 2 - beta + 2, a2 = 3 * beta - alpha - 6, a1 = 5 - 3 * beta + alpha / 2;
    ConvPass p = base_pass(g, info, boundary);
    auto set_x = [&](ConvPass &q) { if (xpadded) set_x_padded(q, g, xpadded); else set_x_virtual(q, g, xsrc, x_dtype); };
    const int tdt = g.t1h ? PB_F16 : PB_F32;
    void *T1 = g.t1h ? g.t1h : static_cast<void *>(t1), *T2 = g.t2h ? g.t2h : static_cast<void *>(t2);
    // t1 = K * (a3 x) + a2 x
    if (xpadded) set_in_padded(p, g, xpadded); else set_in_virtual(p, g, xsrc, x_dtype);
    set_x(p); set_out_padded(p, g, T1, tdt);
    p.scale = a3; p.coef = a2;
    steps[0] = p;
    // t2 = K * t1 + a1 x
    set_in_padded(p, g, T1, tdt); set_out_padded(p, g, T2, tdt);
    p.scale = 1.f; p.coef = a1;
    steps[1] = p;
    // y = K * t2 + beta x   (only the crop is needed)
    set_in_padded(p, g, T2, tdt); set_out_interior(p, g, dst, dst_dtype);
    p.coef = beta; p.clamp01 = clamp01;
    steps[2] = p;
}

// what the spectra of a polynomial with these steps should be those of (PolySpec; conv.hip: pb_poly_spec_mode)
// gaussians: the records are (or will be) point-symmetric Gaussians the estimation itself builds on an odd ker_size grid
PolySpec poly_spec(pb_ctx *ctx, const ConvPass *steps, float alpha, float beta, bool gaussians) {
    const int mode = pb_poly_spec_mode(ctx, steps);
    if (!mode) {
        // (the zero boundary on an image too small for the ring form, the estimation's own Gaussians: every kernel on three
        // window steps -- a fact of the call again, so the polynomial issues the wave body's three launches and nothing else)
        if (steps[0].boundary == PB_ZERO && gaussians && ctx->poly_always && pb_poly_three_steps_ok(ctx, steps)) {
            PolySpec ps = no_poly();
            ps.always = 2;
            return ps;
        }
        return no_poly();
    }
    // (128 x 128 windows need an image of some size: a workgroup takes ~38 us for its pair whatever the launch, and a 700 x 500
    // image yields 72 of them for 256 CUs -- 0.35 against 0.32 ms per call.  The rule looks at ONE image, not at the batch, so
    // that what an image gets does not depend on the batch it travels in; the threshold sits just below 1080p x 3 channels
    // (396 pairs at 90 x 90 tiles: alone 4 % slower through 128 x 128 windows, in a batch of 32 10 % faster))
    const long pairs128 = (long)((steps[2].W + 179) / 180) * ((steps[2].H + 89) / 90) * steps[2].C;
    const float cost128 = pairs128 >= ctx->poly_min_pairs128 ? ctx->poly_cost128 : 0.f;
    // (every composite of a 25-tap kernel fits a 128 x 128 window -- halo <= 36, tile >= 56 -- so where those windows are admitted
    // and every kernel is the estimation's own Gaussian, "every image takes one window pass" is a fact of the call's options and
    // sizes: PolySpec.always, and pb_launch_conv_poly issues the two window launches only)
    const int always = (mode == 3 && cost128 > 0.f && gaussians && ctx->poly_always) ? 1 : 0;
    // (under the zero boundary only that class takes one pass: interior by the window pass, frame by three ring steps)
    if (steps[0].boundary != PB_WRAP && !always) return no_poly();
    return PolySpec{mode, alpha / 2 - beta + 2, 3 * beta - alpha - 6, 5 - 3 * beta + alpha / 2, beta, ctx->poly_gain, ctx->poly_min_area, cost128, always};
}

// y = a3 K^3 x + a2 K^2 x + a1 K x + beta x by Horner, three stencil passes (deblurring.py:122-138).
// X is either the un-padded image (virtual pad) or a padded fp32 image (after edgetaper).
int run_polynomial(pb_ctx *ctx, const Geometry &g, const void *xsrc, int x_dtype, const float *xpadded,
                   const pb_blur_info *info, float alpha, float beta, int boundary, float *t1, float *t2, void *dst,
                   int dst_dtype, int clamp01) {
    const float a3 = alpha / 2 - beta + 2, a2 = 3 * beta - alpha - 6, a1 = 5 - 3 * beta + alpha / 2;
    if (g.sep1) {
        // K ~= K2 after K1 (estimate.hip: sep_records_kernel).  One launch per step keeps u = K1 * t in LDS (conv_xt.hip);
        // dtype combinations it is not built for (8-bit images) -- or PB_XT=2 -- take two launches through the sparse
        // phase lists of the general body: u = K1 * t, then t' = scale (K2 * u) + coef x
        const bool two_launch = ctx->xt_two_launch != 0;
        const float scale[3] = {a3, 1.f, 1.f}, coef[3] = {a2, a1, beta};
        float *tmp[2] = {t1, t2};
        ConvPass p1 = base_pass(g, g.sep1, boundary), p2 = base_pass(g, g.sep2, boundary);
        set_x_virtual(p1, g, xsrc, x_dtype); set_out_padded(p1, g, g.sep_u);          // coef = 0: x is not used
        set_x_virtual(p2, g, xsrc, x_dtype);
        for (int step = 0; step < 3; ++step) {
            if (step < 2) set_out_padded(p2, g, tmp[step]); else set_out_interior(p2, g, dst, dst_dtype);
            p2.scale = scale[step]; p2.coef = coef[step]; p2.clamp01 = step == 2 ? clamp01 : 0;
            int rc = PB_ERR_UNSUPPORTED;
            if (!two_launch) {
                if (step == 0) set_in_virtual(p2, g, xsrc, x_dtype); else set_in_padded(p2, g, tmp[step - 1]);
                rc = pb_launch_conv_xt(ctx, p2);
                if (rc != PB_OK && rc != PB_ERR_UNSUPPORTED) return rc;
                if (rc == PB_OK) {
                    // images whose exact kernel is rank-1 were skipped there: the exact separable body does them
                    ConvPass pe = p2;
                    pe.info = info; pe.skip_general = 1;
                    rc = pb_launch_conv(ctx, pe);
                    if (rc) return rc;
                }
            }
            if (rc == PB_ERR_UNSUPPORTED) {
                if (step == 0) set_in_virtual(p1, g, xsrc, x_dtype); else set_in_padded(p1, g, tmp[step - 1]);
                rc = pb_launch_conv(ctx, p1);
                if (rc) return rc;
                set_in_padded(p2, g, g.sep_u);
                rc = pb_launch_conv(ctx, p2);
                if (rc) return rc;
            }
        }
        return PB_OK;
    }
    ConvPass steps[3];
    make_steps(g, xsrc, x_dtype, xpadded, info, alpha, beta, boundary, t1, t2, dst, dst_dtype, clamp01, steps);
    if (g.big_taps) {
        for (int s = 0; s < 3; ++s) {
            const int rc = pb_launch_conv_big(ctx, steps[s], g.big_taps, g.big_ksize);
            if (rc) return rc;
        }
        return PB_OK;
    }
    // (under the wrap boundary the three steps are one filter, deblurring.py:139-169: images for which one window pass
    // with that filter's spectrum is the cheaper form take it, pb_fft_sel.poly; the spectra are then the polynomial's)
    // (records the estimation has just built under PolySpec.always keep that spec: their spectra are already those it asks for)
    const bool by_est = ctx->khat_by_estimate && ctx->khat_owner == info && ctx->khat_B == g.B && ctx->poly_built.always;
    ctx->poly_want = poly_spec(ctx, steps, alpha, beta, by_est || g.est_gaussians);
    const int rc = pb_launch_conv_poly(ctx, steps);
    ctx->poly_want = no_poly();
    return rc;
}

struct InverseScratch {
    float *t1, *t2, *y, *ox, *nM;
};

// src: what gets deconvolved (cur, or the smooth component).  dst: (B,C,H,W) of dst_dtype.
int inverse_filter(pb_ctx *ctx, const Geometry &g, const void *src, int src_dtype, void *dst, int dst_dtype,
                   const pb_blur_info *info, float alpha, float beta, int boundary, int edgetaping, int remove_halo,
                   const void *g0x, const void *g0y, const float *nM, int final_clamp,
                   const void *recomb_cur = nullptr, int recomb_cur_dtype = 0, const float *recomb_smooth = nullptr,
                   int g_dtype = PB_F32) {
    // recomb_cur (only with remove_halo): the halo kernel also adds back the detail layer cur - recomb_smooth
    float *t1 = nullptr, *t2 = nullptr;
    if (!g.t1h) {
        t1 = static_cast<float *>(pb_scratch(ctx, "inv.t1", sizeof(float) * g.P * g.pplane));
        t2 = static_cast<float *>(pb_scratch(ctx, "inv.t2", sizeof(float) * g.P * g.pplane));
        if (!t1 || !t2) return PB_ERR_NOMEM;
    }
    const float *xpadded = nullptr;
    if (edgetaping) {
        float *pa = static_cast<float *>(pb_scratch(ctx, "inv.pa", sizeof(float) * g.P * g.pplane));
        if (!pa) return PB_ERR_NOMEM;
        float *res = nullptr;
        int rc = run_edgetaper(ctx, g, src, src_dtype, info, boundary, pa, t1, &res);   // t1 is free until the polynomial
        if (rc) return rc;
        xpadded = res;
    }
    if (!remove_halo)
        return run_polynomial(ctx, g, src, src_dtype, xpadded, info, alpha, beta, boundary, t1, t2, dst, dst_dtype,
                              final_clamp);
    float *y = static_cast<float *>(pb_scratch(ctx, "inv.y", sizeof(float) * g.P * g.HW));
    if (!y) return PB_ERR_NOMEM;
    int rc = run_polynomial(ctx, g, src, src_dtype, xpadded, info, alpha, beta, boundary, t1, t2, y, PB_F32, 0);
    if (rc) return rc;
    void *ox = pb_scratch(ctx, "inv.ox", dsize(g_dtype) * g.P * g.HW);    // (in the type of grad_img's planes)
    if (!ox) return PB_ERR_NOMEM;
    rc = pb_fourier_gradients_typed(ctx, y, g.P, g.H, g.W, ox, nullptr, g_dtype);     // only gout_x is used (deblurring.py:174)
    if (rc) return rc;
    if (xpadded)
        return pb_halo_apply(ctx, xpadded + (long)g.pad * g.pp + g.pad, PB_F32, g.pp, g.pplane, y, g0x, g0y, ox, nM, dst,
                             dst_dtype, g.P, g.H, g.W, final_clamp, recomb_cur, recomb_cur_dtype, recomb_smooth, g_dtype);
    return pb_halo_apply(ctx, src, src_dtype, g.W, g.HW, y, g0x, g0y, ox, nM, dst, dst_dtype, g.P, g.H, g.W, final_clamp,
                         recomb_cur, recomb_cur_dtype, recomb_smooth, g_dtype);
}

int check_shape(pb_ctx *ctx, int dtype, int B, int C, int H, int W, int allow_u8 = 0) {
    if (!ctx) return PB_ERR_BADARG;
    if (dtype != PB_F32 && dtype != PB_F16 && !(allow_u8 && dtype == PB_U8))
        return pb_fail(ctx, PB_ERR_BADARG, allow_u8 ? "dtype must be PB_F32, PB_F16 or PB_U8" : "dtype must be PB_F32 or PB_F16");
    if (B < 1 || C < 1 || H < 2 || W < 2) return pb_fail(ctx, PB_ERR_BADARG, "bad shape (%d,%d,%d,%d)", B, C, H, W);
    return PB_OK;
}

}  // namespace

extern "C" {

int pb_estimate_blur(pb_ctx *ctx, const void *in, int dtype, int B, int C, int H, int W, const pb_options *opt,
                     pb_blur_info *dev_info) {
    int rc = check_shape(ctx, dtype, B, C, H, W, 1);
    if (rc) return rc;
    if (!in || !opt || !dev_info) return pb_fail(ctx, PB_ERR_BADARG, "null argument");
    PB_HIP(hipSetDevice(ctx->device));
    pb_forget_records(ctx, dev_info, B);
    return pb_estimate_impl(ctx, in, dtype, B, C, H, W, opt, dev_info);
}

int pb_make_kernels(pb_ctx *ctx, int B, const float *host_sigma, const float *host_rho, const float *host_theta_rad,
                    int support, pb_blur_info *dev_info) {
    if (!ctx || B < 1 || !host_sigma || !host_rho || !host_theta_rad || !dev_info) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    std::vector<pb_blur_info> h(B);
    memset(h.data(), 0, sizeof(pb_blur_info) * B);
    for (int i = 0; i < B; ++i) { h[i].sigma = host_sigma[i]; h[i].rho = host_rho[i]; h[i].theta = host_theta_rad[i]; }
    PB_HIP(hipMemcpyAsync(dev_info, h.data(), sizeof(pb_blur_info) * B, hipMemcpyHostToDevice, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    const int rc = pb_make_kernels_dev(ctx, B, dev_info, support, 0);
    return rc ? rc : pb_cache_records(ctx, dev_info, B);
}

int pb_make_separable_kernels(pb_ctx *ctx, int B, const pb_blur_info *dev_info, pb_blur_info *dev_sep, int support,
                              int ker_size) {
    if (!ctx || B < 1 || !dev_info || !dev_sep) return PB_ERR_BADARG;
    pb_options o;
    pb_default_options(&o);
    o.ker_size = ker_size;
    const int ksize = pb_kernel_size(&o);
    if (!ksize || ksize > PB_KSIZE || !(ksize & 1))
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "ker_size %d: the separable approximation is built for odd sizes from 3 to %d (an even size is the reference's off-centre grid, filters.py:78: not built)", ker_size, PB_KSIZE);
    PB_HIP(hipSetDevice(ctx->device));
    pb_forget_records(ctx, dev_sep, 2 * B);
    return pb_make_sep_records(ctx, B, dev_info, dev_sep, support, ksize);
}

int pb_set_kernels(pb_ctx *ctx, int B, const float *host_taps, int support, pb_blur_info *dev_info) {
    if (!ctx || B < 1 || !host_taps || !dev_info) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    std::vector<pb_blur_info> h(B);
    memset(h.data(), 0, sizeof(pb_blur_info) * B);
    for (int i = 0; i < B; ++i) memcpy(h[i].kernel, host_taps + (size_t)i * PB_KSIZE * PB_KSIZE, sizeof(float) * PB_KSIZE * PB_KSIZE);
    PB_HIP(hipMemcpyAsync(dev_info, h.data(), sizeof(pb_blur_info) * B, hipMemcpyHostToDevice, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    int rc = pb_make_kernels_dev(ctx, B, dev_info, support, 1);
    if (!rc) rc = pb_cache_records(ctx, dev_info, B);             // (forgets what was known about these records, flip set included)
    if (rc) return rc;
    // taps that are not point-symmetric: keep their reflection for the wrap boundary's passes (common.h: FlipSet)
    bool asym = false;
    for (int i = 0; i < B && !asym; ++i) {
        const float *k = host_taps + (size_t)i * PB_KSIZE * PB_KSIZE;
        for (int e = 0; e < PB_KSIZE * PB_KSIZE / 2 && !asym; ++e) asym = k[e] != k[PB_KSIZE * PB_KSIZE - 1 - e];
    }
    if (asym) {
        pb_ctx::FlipSet f;
        f.B = B; f.support = support;
        f.taps.resize((size_t)B * PB_KSIZE * PB_KSIZE);
        for (int i = 0; i < B; ++i)
            for (int e = 0; e < PB_KSIZE * PB_KSIZE; ++e)
                f.taps[(size_t)i * PB_KSIZE * PB_KSIZE + e] = host_taps[(size_t)i * PB_KSIZE * PB_KSIZE + (PB_KSIZE * PB_KSIZE - 1 - e)];
        ctx->flip_sets[dev_info] = std::move(f);
    }
    return PB_OK;
}

// The records a wrap-boundary pass runs with: the caller's, or -- caller-supplied taps that are not point-symmetric -- a copy built
// from the reflected taps (the reference's method='fft' convolves, filters.py:33-36; the records hold correlation taps).
static int wrap_records(pb_ctx *ctx, const pb_blur_info *dev_info, int B, int boundary, const pb_blur_info **use) {
    *use = dev_info;
    if (boundary != PB_WRAP) return PB_OK;
    const auto it = ctx->flip_sets.find(dev_info);
    if (it == ctx->flip_sets.end() || it->second.B < B) return PB_OK;
    pb_blur_info *flipped = static_cast<pb_blur_info *>(pb_scratch(ctx, "conv.flipinfo", sizeof(pb_blur_info) * (size_t)B));
    if (!flipped) return PB_ERR_NOMEM;
    std::vector<pb_blur_info> h(B);
    memset(h.data(), 0, sizeof(pb_blur_info) * B);
    for (int i = 0; i < B; ++i) memcpy(h[i].kernel, it->second.taps.data() + (size_t)i * PB_KSIZE * PB_KSIZE, sizeof(float) * PB_KSIZE * PB_KSIZE);
    const int support = it->second.support;
    pb_forget_records(ctx, flipped, B);
    PB_HIP(hipMemcpyAsync(flipped, h.data(), sizeof(pb_blur_info) * B, hipMemcpyHostToDevice, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    int rc = pb_make_kernels_dev(ctx, B, flipped, support, 1);
    if (!rc) rc = pb_cache_records(ctx, flipped, B);
    if (rc) return rc;
    *use = flipped;
    return PB_OK;
}

int pb_fourier_gradients(pb_ctx *ctx, const float *planes, int P, int H, int W, float *gx, float *gy) {
    if (!ctx || !planes) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    return pb_fourier_gradients_impl(ctx, planes, P, H, W, gx, gy);
}

int pb_inverse_filter(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                      const pb_blur_info *dev_info, float alpha, float beta, int boundary, int edgetaping,
                      int remove_halo, const float *grad0_x, const float *grad0_y) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!in || !out || !dev_info) return pb_fail(ctx, PB_ERR_BADARG, "null argument");
    if (remove_halo && (!grad0_x || !grad0_y)) return pb_fail(ctx, PB_ERR_BADARG, "remove_halo needs grad0_x / grad0_y");
    PB_HIP(hipSetDevice(ctx->device));
    const Geometry g = geometry(B, C, H, W);
    float *nM = nullptr;
    if (remove_halo) {
        nM = static_cast<float *>(pb_scratch(ctx, "inv.nM", sizeof(float) * g.P));
        if (!nM) return PB_ERR_NOMEM;
        rc = pb_grad_energy(ctx, grad0_x, grad0_y, nM, g.P, g.HW);
        if (rc) return rc;
    }
    const pb_blur_info *recs = nullptr;
    rc = wrap_records(ctx, dev_info, B, boundary, &recs);
    if (rc) return rc;
    return inverse_filter(ctx, g, in, dtype, out, dtype, recs, alpha, beta, boundary, edgetaping, remove_halo, grad0_x,
                          grad0_y, nM, 1);
}

int pb_convolve2d(pb_ctx *ctx, const float *in, float *out, int B, int C, int Hp, int Wp, const pb_blur_info *dev_info,
                  int boundary) {
    if (!ctx || !in || !out || !dev_info || Hp <= 2 * PB_KRAD || Wp <= 2 * PB_KRAD) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    // The given image IS the padded domain: address it as a padded source with pitch Wp.
    Geometry g = geometry(B, C, Hp - 2 * PB_KRAD, Wp - 2 * PB_KRAD);
    g.pp = Wp; g.pplane = (long)Hp * Wp;
    const pb_blur_info *recs = nullptr;
    const int rcw = wrap_records(ctx, dev_info, B, boundary, &recs);
    if (rcw) return rcw;
    ConvPass p = base_pass(g, recs, boundary);
    set_in_padded(p, g, in); set_x_padded(p, g, in); set_out_padded(p, g, out);
    p.scale = 1.f; p.coef = 0.f;
    return pb_launch_conv(ctx, p);
}

int pb_edgetaper(pb_ctx *ctx, const float *in, float *out, int B, int C, int Hp, int Wp, const pb_blur_info *dev_info,
                 int boundary) {
    if (!ctx || !in || !out || !dev_info || Hp <= 2 * PB_KRAD || Wp <= 2 * PB_KRAD) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    Geometry g = geometry(B, C, Hp - 2 * PB_KRAD, Wp - 2 * PB_KRAD);
    g.pp = Wp; g.pplane = (long)Hp * Wp;
    float *tmp = static_cast<float *>(pb_scratch(ctx, "taper.tmp", sizeof(float) * g.P * g.pplane));
    if (!tmp) return PB_ERR_NOMEM;
    const pb_blur_info *recs = nullptr;
    const int rcw = wrap_records(ctx, dev_info, B, boundary, &recs);
    if (rcw) return rcw;
    ConvPass p = base_pass(g, recs, boundary);
    p.epilogue = EPI_TAPER;
    set_in_padded(p, g, in); set_x_padded(p, g, in); set_out_padded(p, g, out);
    int rc = pb_launch_conv(ctx, p);
    if (rc) return rc;
    set_in_padded(p, g, out); set_x_padded(p, g, out); set_out_padded(p, g, tmp);
    rc = pb_launch_conv(ctx, p);
    if (rc) return rc;
    set_in_padded(p, g, tmp); set_x_padded(p, g, tmp); set_out_padded(p, g, out);
    return pb_launch_conv(ctx, p);
}

int pb_halo_mask(pb_ctx *ctx, const float *x, const float *y, const float *grad0_x, const float *grad0_y, float *out,
                 int B, int C, int H, int W) {
    if (!ctx || !x || !y || !grad0_x || !grad0_y || !out) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    const Geometry g = geometry(B, C, H, W);
    float *nM = static_cast<float *>(pb_scratch(ctx, "inv.nM", sizeof(float) * g.P));
    float *ox = static_cast<float *>(pb_scratch(ctx, "inv.ox", sizeof(float) * g.P * g.HW));
    if (!nM || !ox) return PB_ERR_NOMEM;
    int rc = pb_grad_energy(ctx, grad0_x, grad0_y, nM, g.P, g.HW);
    if (rc) return rc;
    rc = pb_fourier_gradients_impl(ctx, y, g.P, H, W, ox, nullptr);
    if (rc) return rc;
    return pb_halo_apply(ctx, x, PB_F32, W, g.HW, y, grad0_x, grad0_y, ox, nM, out, PB_F32, g.P, H, W, 0);
}

int pb_dt_recursive_filter(pb_ctx *ctx, const void *in, const void *joint, void *out, int dtype, int B, int C, int H,
                           int W, float sigma_s, float sigma_r, int num_iterations) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!in || !out || num_iterations < 1) return pb_fail(ctx, PB_ERR_BADARG, "bad argument");
    PB_HIP(hipSetDevice(ctx->device));
    const long n = (long)B * C * H * W;
    if (dtype == PB_F32) return pb_dt_filter_impl(ctx, in, joint, dtype, static_cast<float *>(out), B, C, H, W, sigma_s, sigma_r, num_iterations);
    float *tmp = static_cast<float *>(pb_scratch(ctx, "dt.out", sizeof(float) * n));
    if (!tmp) return PB_ERR_NOMEM;
    rc = pb_dt_filter_impl(ctx, in, joint, dtype, tmp, B, C, H, W, sigma_s, sigma_r, num_iterations);
    if (rc) return rc;
    return pb_convert_from_float(ctx, tmp, out, dtype, n);
}

int pb_dt_normalized_convolution(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                                 float sigma_s, float sigma_r, int num_iterations) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!in || !out || num_iterations < 1 || !(sigma_r > 0.f)) return pb_fail(ctx, PB_ERR_BADARG, "bad argument");
    PB_HIP(hipSetDevice(ctx->device));
    const long n = (long)B * C * H * W;
    if (dtype == PB_F32) return pb_nc_filter_impl(ctx, in, dtype, static_cast<float *>(out), B, C, H, W, sigma_s, sigma_r, num_iterations);
    float *tmp = static_cast<float *>(pb_scratch(ctx, "dt.out", sizeof(float) * n));
    if (!tmp) return PB_ERR_NOMEM;
    rc = pb_nc_filter_impl(ctx, in, dtype, tmp, B, C, H, W, sigma_s, sigma_r, num_iterations);
    if (rc) return rc;
    return pb_convert_from_float(ctx, tmp, out, dtype, n);
}

int pb_bilateral5(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!in || !out) return pb_fail(ctx, PB_ERR_BADARG, "null argument");
    PB_HIP(hipSetDevice(ctx->device));
    return pb_bilateral5_impl(ctx, in, dtype, out, dtype, B * C, H, W);
}

// ---------------------------------------------------------------------------------------------
// polyblur_deblurring (deblurring.py:23-96)
// ---------------------------------------------------------------------------------------------
int pb_polyblur_batch(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                      const pb_options *opt, pb_blur_info *host_info) {
    int rc = check_shape(ctx, dtype, B, C, H, W, 1);
    if (rc) return rc;
    if (!in || !out || !opt) return pb_fail(ctx, PB_ERR_BADARG, "null argument");
    if (in == out) return pb_fail(ctx, PB_ERR_BADARG, "out may not alias in");
    if (opt->n_iter < 0) return pb_fail(ctx, PB_ERR_BADARG, "n_iter < 0");
    if (opt->boundary != PB_WRAP && opt->boundary != PB_ZERO) return pb_fail(ctx, PB_ERR_BADARG, "bad boundary");
    const int ksize = pb_kernel_size(opt);
    if (!ksize) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "ker_size %d: sizes from 2 to %d are built", opt->ker_size, PB_KSIZE_MAX);
    if (opt->separable_approx && opt->edgetaping)
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "edgetaping is not defined for the separable approximation");
    // (an even ker_size is the reference's off-centre tap grid, blur_estimation.py:222 / filters.py:78: the 1-D kernels of the
    // separable approximation are built on the centred grid only; the edgetaper takes it since round 6 -- its weights are
    // autocorrelations of the projections, which do not care where the taps sit)
    if (!(ksize & 1) && opt->separable_approx)
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "ker_size %d: the separable approximation is built for odd sizes only", ksize);
    // (kernels beyond the 25 x 25 record take conv_big.hip's pass: no separable approximation there; the edgetaper since round 6)
    if (ksize > PB_KSIZE && opt->separable_approx)
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "ker_size %d: the separable approximation is built for sizes up to %d", ksize, PB_KSIZE);
    PB_HIP(hipSetDevice(ctx->device));
    // (every iteration's choices of body keep a slot of their own; whichever way the call ends, later passes on this context
    // read and write slot 0 again, and what one-pass spec the call asked for is forgotten)
    struct SlotGuard { pb_ctx *c; ~SlotGuard() { c->sel_slot = 0; c->poly_want = no_poly(); c->poly_want2 = no_poly(); } } slot_guard{ctx};
    ctx->sel_slot = 0;
    ctx->sel2_mask = 0;
    Geometry g = geometry(B, C, H, W, ksize / 2);
    const bool poly_eligible = (opt->boundary == PB_WRAP || ctx->zero_ring) && !opt->edgetaping && !opt->separable_approx && ksize <= PB_KSIZE;
    g.est_gaussians = (ksize & 1) && ksize <= PB_KSIZE && !opt->separable_approx;
    g.taper_windows = opt->edgetaping && g.est_gaussians && ctx->poly_always && ctx->poly_mode != 0 && ctx->fft_min_phases >= 0 && ctx->fft_wave;
    const long n = (long)g.P * g.HW;
    const int n_iter = opt->n_iter;
    if (n_iter == 0) {
        PB_HIP(hipMemcpyAsync(out, in, dsize(dtype) * n, hipMemcpyDeviceToDevice, ctx->stream));
        return PB_OK;
    }
    if (dtype == PB_U8 && (opt->remove_halo || opt->prefilter != PB_PREFILTER_NONE)) {
        // halo masking and the prefilters read the current image from several more kernels: those options run on
        // an fp32 copy, the conversions at either end being the same img_as_float32 / img_as_ubyte
        float *in32 = static_cast<float *>(pb_scratch(ctx, "pipe.u8in", sizeof(float) * n));
        float *out32 = static_cast<float *>(pb_scratch(ctx, "pipe.u8out", sizeof(float) * n));
        if (!in32 || !out32) return PB_ERR_NOMEM;
        rc = pb_convert_to_float(ctx, in, PB_U8, in32, n);
        if (rc) return rc;
        rc = pb_polyblur_batch(ctx, in32, out32, PB_F32, B, C, H, W, opt, host_info);
        if (rc) return rc;
        return pb_convert_from_float(ctx, out32, out, PB_U8, n);
    }
    pb_blur_info *infos = static_cast<pb_blur_info *>(pb_scratch(ctx, "pipe.info", sizeof(pb_blur_info) * (size_t)n_iter * B));
    if (!infos) return PB_ERR_NOMEM;
    // images between iterations are fp32 whatever the I/O type: fp16 / 8-bit I/O rounds once, in the last store
    // (fp32 I/O alternates between `out` and one scratch image; narrower I/O needs two fp32 scratch images)
    const int work = PB_F32;
    void *tmpimg = nullptr, *tmpimg2 = nullptr;
    if (n_iter > 1) {
        tmpimg = pb_scratch(ctx, "pipe.img", dsize(work) * n);
        if (!tmpimg) return PB_ERR_NOMEM;
    }
    if (n_iter > 2 && dtype != PB_F32) {
        tmpimg2 = pb_scratch(ctx, "pipe.img2", dsize(work) * n);
        if (!tmpimg2) return PB_ERR_NOMEM;
    }
    // gradients of the ORIGINAL image, used by halo masking in every iteration (deblurring.py:61,83)
    // (an fp16 call keeps them -- and every iteration's d/dx of the deblurred image -- as fp16 planes where the line transforms
    // can write such planes: halo_kernel's TG)
    void *g0x = nullptr, *g0y = nullptr;
    float *nM = nullptr;
    const int g_dtype = (dtype == PB_F16 && pb_gradient_planes_half(ctx, H, W)) ? PB_F16 : PB_F32;
    if (opt->remove_halo) {
        g0x = pb_scratch(ctx, "pipe.g0x", dsize(g_dtype) * n);
        g0y = pb_scratch(ctx, "pipe.g0y", dsize(g_dtype) * n);
        nM = static_cast<float *>(pb_scratch(ctx, "inv.nM", sizeof(float) * g.P));
        if (!g0x || !g0y || !nM) return PB_ERR_NOMEM;
        const float *in32 = static_cast<const float *>(in);
        if (dtype != PB_F32) {
            float *conv = static_cast<float *>(pb_scratch(ctx, "pipe.in32", sizeof(float) * n));
            if (!conv) return PB_ERR_NOMEM;
            rc = pb_convert_to_float(ctx, in, dtype, conv, n);
            if (rc) return rc;
            in32 = conv;
        }
        rc = pb_fourier_gradients_typed(ctx, in32, g.P, H, W, g0x, g0y, g_dtype);
        if (rc) return rc;
        rc = pb_grad_energy(ctx, g0x, g0y, nM, g.P, g.HW, g_dtype);
        if (rc) return rc;
    }
    float *smooth = nullptr, *ybuf = nullptr;
    if (opt->prefilter != PB_PREFILTER_NONE) {
        smooth = static_cast<float *>(pb_scratch(ctx, "pipe.smooth", sizeof(float) * n));
        ybuf = static_cast<float *>(pb_scratch(ctx, "pipe.y", sizeof(float) * n));
        if (!smooth || !ybuf) return PB_ERR_NOMEM;
    }
    if (opt->half_temporaries) {
        if (dtype != PB_F16 || opt->edgetaping || opt->separable_approx)
            return pb_fail(ctx, PB_ERR_UNSUPPORTED, "half_temporaries: fp16 images only, not with edgetaping / separable_approx");
        g.t1h = pb_scratch(ctx, "inv.t1h", sizeof(__half) * g.P * g.pplane);
        g.t2h = pb_scratch(ctx, "inv.t2h", sizeof(__half) * g.P * g.pplane);
        if (!g.t1h || !g.t2h) return PB_ERR_NOMEM;
    }
    pb_blur_info *sep = nullptr;
    if (opt->separable_approx && n_iter > 0) {
        sep = static_cast<pb_blur_info *>(pb_scratch(ctx, "pipe.sepinfo", sizeof(pb_blur_info) * 2 * (size_t)B));
        g.sep_u = static_cast<float *>(pb_scratch(ctx, "inv.u", sizeof(float) * g.P * g.pplane));
        if (!sep || !g.sep_u) return PB_ERR_NOMEM;
        g.sep1 = sep; g.sep2 = sep + B;
    }
    const void *cur = in;
    for (int it = 0; it < n_iter; ++it) {
        // the last iteration must land in `out`; alternate between out and tmpimg before that
        void *dst = ((n_iter - 1 - it) % 2 == 0) ? out : tmpimg;
        if (dtype != PB_F32 && it != n_iter - 1) dst = (it % 2 == 0) ? tmpimg : tmpimg2;
        const int cur_dtype = it == 0 ? dtype : work, dst_dtype = it == n_iter - 1 ? dtype : work;
        pb_blur_info *info = infos + (size_t)it * B;
        ctx->sel_slot = it;                       // (this iteration's choices of body keep a slot of their own: pb_body_selection)
        // The estimation ends with the kernels' spectra and the images' choice of body: it has to know whether this
        // iteration's polynomial may take the one-pass form (pb_fft_sel.poly) -- exactly what run_polynomial will ask for.
        if (poly_eligible) {
            const int src_dtype = opt->prefilter != PB_PREFILTER_NONE ? PB_F32 : cur_dtype;
            const int last_out = (opt->remove_halo || opt->prefilter != PB_PREFILTER_NONE) ? PB_F32 : dst_dtype;
            ConvPass steps[3];
            make_steps(g, cur, src_dtype, nullptr, info, opt->alpha, opt->beta, opt->boundary, nullptr, nullptr, dst, last_out, 1, steps);
            ctx->poly_want = poly_spec(ctx, steps, opt->alpha, opt->beta, (ksize & 1) != 0);
            // (the zero boundary in its window-pass + ring form: the ring steps run with the kernels' OWN spectra -- the estimation
            // builds them as its second set instead of a khat_kernel launch in front of the ring)
            if (opt->boundary == PB_ZERO && ctx->poly_want.always == 1) { ctx->poly_want2 = no_poly(); ctx->poly_want2.always = 2; }
        }
        else if (g.taper_windows) {
            ctx->poly_want = no_poly(); ctx->poly_want.always = 2;      // (the blends come first: the kernels' own spectra, window form for all)
            // (... and the polynomial behind them wants ITS spectra: the second set, under the spec run_polynomial will ask for --
            // its source is the padded, tapered plane)
            if (ctx->poly_padded) {
                const int src_dtype = opt->prefilter != PB_PREFILTER_NONE ? PB_F32 : cur_dtype;
                const int last_out = (opt->remove_halo || opt->prefilter != PB_PREFILTER_NONE) ? PB_F32 : dst_dtype;
                ConvPass steps[3];
                float *pa = static_cast<float *>(pb_scratch(ctx, "inv.pa", sizeof(float) * g.P * g.pplane));
                if (!pa) return PB_ERR_NOMEM;
                make_steps(g, cur, src_dtype, pa, info, opt->alpha, opt->beta, opt->boundary, nullptr, nullptr, dst, last_out, 1, steps);
                const PolySpec want = ctx->poly_want;
                ctx->poly_want2 = poly_spec(ctx, steps, opt->alpha, opt->beta, true);
                ctx->poly_want = want;
                if (opt->boundary == PB_ZERO) ctx->poly_want2 = no_poly();      // (the ring there wants a third set: left to its own launch)
            }
        }
        rc = pb_estimate_impl(ctx, cur, cur_dtype, B, C, H, W, opt, info);
        ctx->poly_want = no_poly(); ctx->poly_want2 = no_poly();
        if (rc) return rc;
        if (ksize > PB_KSIZE) {
            rc = pb_build_big_taps(ctx, info, B, ksize, (!(ksize & 1) && opt->boundary == PB_WRAP) ? 1 : 0, &g.big_taps);
            if (rc) return rc;
            g.big_ksize = ksize;
        }
        if (sep) {
            rc = pb_make_sep_records(ctx, B, info, sep, opt->support, ksize);
            if (rc) return rc;
        }
        if (opt->prefilter == PB_PREFILTER_NONE) {
            rc = inverse_filter(ctx, g, cur, cur_dtype, dst, dst_dtype, info, opt->alpha, opt->beta, opt->boundary,
                                opt->edgetaping, opt->remove_halo, g0x, g0y, nM, 1, nullptr, 0, nullptr, g_dtype);
            if (rc) return rc;
        } else {
            if (opt->prefilter == PB_PREFILTER_BILATERAL)
                rc = pb_bilateral5_impl(ctx, cur, cur_dtype, smooth, PB_F32, g.P, H, W);
            else if (opt->prefilter == PB_PREFILTER_NORMALIZED_CONVOLUTION)
                rc = pb_nc_filter_impl(ctx, cur, cur_dtype, smooth, B, C, H, W, opt->sigma_s, opt->sigma_r, 1);
            else
                rc = pb_dt_filter_impl(ctx, cur, nullptr, cur_dtype, smooth, B, C, H, W, opt->sigma_s, opt->sigma_r, 1);
            if (rc) return rc;
            if (opt->remove_halo) {                 // the halo kernel recombines as it stores: no ybuf round trip
                rc = inverse_filter(ctx, g, smooth, PB_F32, dst, dst_dtype, info, opt->alpha, opt->beta, opt->boundary,
                                    opt->edgetaping, 1, g0x, g0y, nM, 1, cur, cur_dtype, smooth, g_dtype);
                if (rc) return rc;
            } else {
                rc = inverse_filter(ctx, g, smooth, PB_F32, ybuf, PB_F32, info, opt->alpha, opt->beta, opt->boundary,
                                    opt->edgetaping, 0, g0x, g0y, nM, 1);
                if (rc) return rc;
                rc = pb_recombine(ctx, ybuf, cur, cur_dtype, smooth, dst, dst_dtype, n);
                if (rc) return rc;
            }
        }
        cur = dst;
    }
    ctx->sel_slot = 0;
    if (host_info) {
        PB_HIP(hipMemcpyAsync(host_info, infos, sizeof(pb_blur_info) * (size_t)n_iter * B, hipMemcpyDeviceToHost, ctx->stream));
        PB_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PB_OK;
}

int pb_u8_deinterleave(pb_ctx *ctx, const unsigned char *hwc, unsigned char *chw, int B, int C, int H, int W) {
    if (!ctx || !hwc || !chw || hwc == chw || B < 1 || C < 1 || H < 1 || W < 1) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    return pb_u8_layout(ctx, hwc, chw, B, C, H, W, 1);
}

int pb_u8_interleave(pb_ctx *ctx, const unsigned char *chw, unsigned char *hwc, int B, int C, int H, int W) {
    if (!ctx || !hwc || !chw || hwc == chw || B < 1 || C < 1 || H < 1 || W < 1) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    return pb_u8_layout(ctx, chw, hwc, B, C, H, W, 0);
}

int pb_extract_patches(pb_ctx *ctx, const void *img, void *patches, int dtype, int B, int C, int H, int W, int ph, int pw,
                       int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left, int first, int count) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!img || !patches || ph < 1 || pw < 1 || step_h < 1 || step_w < 1 || n_i < 1 || n_j < 1 || first < 0 || count < 1 ||
        first + count > n_i * n_j)
        return pb_fail(ctx, PB_ERR_BADARG, "extract_patches: bad argument");
    PB_HIP(hipSetDevice(ctx->device));
    return pb_extract_patches_impl(ctx, img, patches, dtype, B, C, H, W, ph, pw, step_h, step_w, n_j, pad_top, pad_left, first, count);
}

int pb_overlap_add(pb_ctx *ctx, const void *patches, void *out, int dtype, int B, int C, int H, int W, int ph, int pw,
                   int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left, const float *dev_win_y,
                   const float *dev_win_x) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!patches || !out || !dev_win_y || !dev_win_x || ph < 1 || pw < 1 || step_h < 1 || step_w < 1 || n_i < 1 || n_j < 1)
        return pb_fail(ctx, PB_ERR_BADARG, "overlap_add: bad argument");
    PB_HIP(hipSetDevice(ctx->device));
    return pb_overlap_add_impl(ctx, patches, out, dtype, B, C, H, W, ph, pw, step_h, step_w, n_i, n_j, pad_top, pad_left,
                               dev_win_y, dev_win_x);
}

int pb_body_selection(pb_ctx *ctx, int iteration, int *host, int B) {
    if (!ctx || !host || B < 1 || iteration < -1) return PB_ERR_BADARG;
    PB_HIP(hipSetDevice(ctx->device));
    const auto it = ctx->scratch.find("conv.fftsel");
    if (it == ctx->scratch.end() || ctx->sel_B != B || it->second.bytes < sizeof(pb_fft_sel) * (size_t)B * PB_SEL_SLOTS)
        return pb_fail(ctx, PB_ERR_BADARG, "pb_body_selection: no selection for %d images on this context", B);
    const int slot = (iteration < 0 ? ctx->sel_last : iteration) % PB_SEL_SLOTS;
    std::vector<pb_fft_sel> h((size_t)B);
    // (an iteration whose polynomial read the second set's selections -- behind an edgetaper -- is reported from there)
    const pb_fft_sel *src = static_cast<const pb_fft_sel *>(it->second.p);
    const auto it2 = ctx->scratch.find("conv.fftsel2");
    if ((ctx->sel2_mask >> slot) & 1u && it2 != ctx->scratch.end() && it2->second.bytes >= sizeof(pb_fft_sel) * (size_t)B * PB_SEL_SLOTS)
        src = static_cast<const pb_fft_sel *>(it2->second.p);
    PB_HIP(hipMemcpyAsync(h.data(), src + (size_t)slot * B, sizeof(pb_fft_sel) * (size_t)B,
                          hipMemcpyDeviceToHost, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) {
        host[6 * b] = h[b].use_fft; host[6 * b + 1] = h[b].pad_[1] == 1 ? -1 : h[b].rf; host[6 * b + 2] = h[b].strip;      // (pad_[1]: the short chain's record workgroup found asymmetric taps)
        host[6 * b + 3] = h[b].poly; host[6 * b + 4] = h[b].hx; host[6 * b + 5] = h[b].hy;
    }
    return PB_OK;
}

int pb_profile_begin(pb_ctx *ctx) {
    if (!ctx) return PB_ERR_BADARG;
    PB_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof) { ctx->evpool.push_back(r.a); ctx->evpool.push_back(r.b); }
    ctx->prof.clear();
    ctx->prof_on = true;
    return PB_OK;
}

int pb_profile_end(pb_ctx *ctx, float *host_ms, int *host_count) {
    if (!ctx || !host_ms || !host_count) return PB_ERR_BADARG;
    ctx->prof_on = false;
    PB_HIP(hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < PB_PROF_NTAGS; ++t) { host_ms[t] = 0.f; host_count[t] = 0; }
    for (auto &r : ctx->prof) {
        float ms = 0.f;
        PB_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        host_ms[r.tag] += ms;
        host_count[r.tag] += 1;
        ctx->evpool.push_back(r.a);
        ctx->evpool.push_back(r.b);
    }
    ctx->prof.clear();
    return PB_OK;
}

int pb_time_inner_loop(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                       const pb_blur_info *dev_info, float alpha, float beta, int boundary, int reps, float *host_ms) {
    int rc = check_shape(ctx, dtype, B, C, H, W);
    if (rc) return rc;
    if (!in || !out || !dev_info || reps < 1 || !host_ms) return pb_fail(ctx, PB_ERR_BADARG, "bad argument");
    PB_HIP(hipSetDevice(ctx->device));
    const Geometry g = geometry(B, C, H, W);
    float *t1 = static_cast<float *>(pb_scratch(ctx, "inv.t1", sizeof(float) * g.P * g.pplane));
    float *t2 = static_cast<float *>(pb_scratch(ctx, "inv.t2", sizeof(float) * g.P * g.pplane));
    if (!t1 || !t2) return PB_ERR_NOMEM;
    rc = run_polynomial(ctx, g, in, dtype, nullptr, dev_info, alpha, beta, boundary, t1, t2, out, dtype, 1);   // warm
    if (rc) return rc;
    PB_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        rc = run_polynomial(ctx, g, in, dtype, nullptr, dev_info, alpha, beta, boundary, t1, t2, out, dtype, 1);
        if (rc) return rc;
    }
    PB_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    PB_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PB_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *host_ms = ms / reps;
    return PB_OK;
}

}  // extern "C"
