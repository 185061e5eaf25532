/*
 * polyblur_hip.h -- C ABI of libpolyblur_hip.so, the MI355X (gfx950) Polyblur engine.
 *
 * Drop-in boundary for the hot path of teboli/polyblur (reference @ /root/reference):
 * the reference has no C ABI of its own -- its Python calls ATen ops, and its three
 * side-car pybind11 modules take torch::Tensor (RF.cpp:43-46,97-99,
 * separable_gaussian2d.cpp:186-191,252-255).  Each entry point below names the
 * reference function(s) (file:line) it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch / pybind types; every image pointer is a DEVICE pointer to a
 *     contiguous NCHW array (the reference's (B,C,H,W) tensors); `float *host_*`
 *     arguments are HOST pointers and make the call synchronise the stream;
 *   - all other calls are asynchronous on the context's stream;
 *   - the caller owns every buffer; inputs are never written; scratch memory lives in
 *     the context; a context runs on one stream at a time and is not thread-safe (one
 *     per host thread / GPU); pb_set_stream orders the new stream behind the work already
 *     queued on the old one, because the scratch buffers are shared;
 *   - return value: PB_OK (0) or a negative pb_status; pb_last_error_string() gives
 *     the text for the most recent failure on that context.
 */
#ifndef POLYBLUR_HIP_H
#define POLYBLUR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PB_VERSION 200           /* major*10000 + minor*100 + patch */
#define PB_KSIZE 25              /* kernel support of the reference (ker_size=25, deblurring.py:23) */
#define PB_KRAD 12
#define PB_MAX_ANGLES 13         /* n_angles + 1 <= 13 */
#define PB_MAX_INTERP 64         /* n_interpolated_angles <= 64 */
#define PB_MAX_PHASES 175        /* 25 kernel rows x 7 window chunks */

typedef struct pb_ctx pb_ctx;

typedef enum pb_status {
    PB_OK = 0,
    PB_ERR_BADARG = -1,
    PB_ERR_UNSUPPORTED = -2,
    PB_ERR_HIP = -3,
    PB_ERR_NOMEM = -4
} pb_status;

/* PB_U8: 8-bit images at the file edge of the reference's CLI flow (main.py:80-82,146): loaded as
 * v * float32(1/255) (skimage 0.19.2 img_as_float32) and stored as clip(rint(v*255), 0, 255)
 * (img_as_ubyte) inside the first / last kernels that touch them; everything in between is fp32.
 * Accepted by pb_polyblur_batch, pb_estimate_blur, pb_u8_interleave / pb_u8_deinterleave.     */
typedef enum pb_dtype { PB_F32 = 0, PB_F16 = 1, PB_U8 = 2 } pb_dtype;

/* Outer boundary of the three reblurring convolutions on the replicate-padded domain:
 * PB_WRAP == reference method='fft'  (circular, deblurring.py:141-169, filters.py:31-35)
 * PB_ZERO == reference method='direct' (zero 'same' padding, filters.py:45-49).        */
typedef enum pb_boundary { PB_WRAP = 0, PB_ZERO = 1 } pb_boundary;

typedef enum pb_prefilter { PB_PREFILTER_NONE = 0, PB_PREFILTER_BILATERAL = 1,
                            PB_PREFILTER_DOMAIN_TRANSFORM = 2,      /* recursive filter, N = 1  */
                            PB_PREFILTER_NORMALIZED_CONVOLUTION = 3 /* NC variant,       N = 1  */ } pb_prefilter;

/* Kernel-support policy: PB_SUPPORT_FULL evaluates every tap of the reference's 25x25 kernel
 * that is not exactly 0.0f (4-tap segments of a kernel row, and outer rows / columns, whose taps
 * all underflowed to zero are skipped, which is bit-identical to evaluating them);
 * PB_SUPPORT_ADAPTIVE also drops the rows / columns whose marginal mass is < 1e-8 and, inside
 * that box, the row segments whose taps are all < 1e-10 (the corners outside the Gaussian's
 * ellipse) -- results agree to fp32 rounding.  The staged halo is rounded up to 4, 6, 8, 10
 * or 12 samples (4, 8 or 12 for rank-1 kernels).                                          */
typedef enum pb_support { PB_SUPPORT_FULL = 0, PB_SUPPORT_ADAPTIVE = 1,
                          /* test/bench flag, OR-ed in: never take the rank-1 (separable) path */
                          PB_SUPPORT_FORCE_GENERAL = 16 } pb_support;

/* Keyword arguments of polyblur_deblurring() (deblurring.py:23-25). */
typedef struct pb_options {
    int32_t n_iter;
    float c, b;                 /* affine blur model (blur_estimation.py:171-185) */
    float alpha, beta;          /* polynomial parameters (deblurring.py:133-135) */
    float sigma_s, sigma_r;     /* domain-transform prefilter (deblurring.py:107) */
    float q;                    /* normalisation quantile in [0, 0.5) (blur_estimation.py:102-105); q > 0 is an exact
                                   torch.quantile by radix select on the device */
    int32_t n_angles;           /* 6 */
    int32_t n_interpolated_angles; /* 30 */
    int32_t remove_halo;
    int32_t edgetaping;
    int32_t prefilter;          /* pb_prefilter; reference prefiltering=True == BILATERAL */
    int32_t discard_saturation;
    int32_t boundary;           /* pb_boundary */
    int32_t support;            /* pb_support */
    float force_theta_deg;      /* < 0: estimate (default); >= 0: bench knob, overrides the
                                   estimated direction so that every kernel is rank-1 */
    int32_t separable_approx;   /* method='direct_separable': every reblurring K is replaced by the x-t separable
                                   APPROXIMATION of the same Gaussian (intent of separable_gaussian2d.cpp:91-183, whose own
                                   formulas do not run).  With q = A X^2 + 2 B X Y + C Y^2 the kernel's quadratic form
                                   (blur_estimation.py:204-207), q = A (X + (B/A) Y)^2 + (C - B^2/A) Y^2: a 1-D Gaussian of
                                   variance 1/A along X, then one of variance A/(AC - B^2) in Y taken along the line
                                   X = -(B/A) Y, each offset split between the two nearest samples by linear
                                   interpolation; X and Y swap roles when C > A (keeps |shear| <= 1).  Both 1-D kernels
                                   are sampled on the ker_size grid and normalised to sum 1.  Opt-in: deviates from the
                                   exact kernel by ~1e-2 (tests state the bound).  Zero boundary; not with edgetaping. */
    int32_t half_temporaries;   /* fp16 images only: store the two Horner temporaries t1, t2 as fp16 (accumulation and the
                                   x operand stay fp32; the images between iterations stay fp32).  Opt-in: the temporaries
                                   reach 9x the image range, where an fp16 ulp is 7.8e-3 -- measured 2e-3 .. 4.4e-3 from
                                   the fp32-temporary result; default 0 = fp32 temporaries             */
    int32_t ker_size;           /* support of the estimated Gaussian and, halved, the replicate pad (deblurring.py:23,
                                   blur_estimation.py:211-232, utils.py:48-53): 2 .. 49 (default 25; 0 means 25); even sizes are
                                   off-centre as in the reference and keep the stencil bodies.  Sizes above 25 do not fit
                                   pb_blur_info: their taps live in the context's scratch and every reblurring step is a plain
                                   LDS-tiled stencil (conv_big.hip: up to 2401 multiply-adds per sample -- 10 to 40 times the
                                   time of the default size); the record's `kernel` then holds the central 25 x 25 taps
                                   renormalised, theta / sigma / rho are exact.  Edgetaping and separable_approx are
                                   built for ODD sizes up to 25 only: with an even size or one above 25 pb_polyblur_batch and
                                   pb_make_separable_kernels return PB_ERR_UNSUPPORTED */
} pb_options;

/* Per-image, per-iteration estimation record (device or host copy). Mirrors the values the
 * reference computes in blur_estimation.py:59-73. */
typedef struct pb_blur_info {
    float gray_min, gray_max;           /* blur_estimation.py:107-108 */
    float mags[PB_MAX_ANGLES];          /* :122-134 */
    float interp[PB_MAX_INTERP];        /* :138-148 */
    int32_t i_min;                      /* :160 */
    float theta;                        /* radians, :167 */
    float sigma, rho;                   /* :171-185 */
    int32_t separable;                  /* 1 if the 25x25 kernel is rank-1 (theta % 90 == 0 or sigma == rho) */
    int32_t radius;                     /* support radius class actually evaluated: 4, 6, 8, 10 or 12 */
    float kernel[PB_KSIZE * PB_KSIZE];  /* :211-232, row-major [y][x] */
    float kx[PB_KSIZE], ky[PB_KSIZE];   /* marginals kx[j] = sum_i k[i][j], ky[i] = sum_j k[i][j];
                                           the exact rank-1 factors when `separable` */
    float acorr_y[PB_KSIZE], acorr_x[PB_KSIZE]; /* autocorrelation of ky / kx at lags 0..24: the closed
                                           form of edgetaper_alpha's 1-D FFTs (edgetaper.py:11-21) */
    float gtaps[(PB_KSIZE + 1) * 32];   /* kernel rows re-laid for the tile stencil: row y = {0,0,0, k[y][0..24], 0,0,0,0};
                                           row 25 is all zeros (the taps of the list's filler phases) */
    float gtaps_odd[(PB_KSIZE + 1) * 32]; /* the same rows shifted by one tap (gtaps_odd[y][n] = gtaps[y][n+1]): the second
                                           alignment of adjacent tap pairs for the packed-FMA stencil */
    /* x-t separable approximation (pb_options.separable_approx; filled in the SECOND record of a pair by
     * pb_make_separable_kernels): xt_first = 1 when the 1-D pass runs along x and the oblique pass walks rows, 0 for the
     * transposed arrangement; xt_g1 = the 1-D pass's taps; per offset i = -12..12 along the oblique pass's axis, the line
     * sits xt_m[i] + f samples across it and the two neighbours get the weights xt_wa[i] = g2 (1-f), xt_wb[i] = g2 f.   */
    int32_t xt_first;
    int32_t xt_exact_rank1;             /* the image's EXACT kernel is rank-1: it takes the exact separable body instead */
    float xt_g1[PB_KSIZE];
    int32_t xt_m[PB_KSIZE];
    float xt_wa[PB_KSIZE], xt_wb[PB_KSIZE];
    int32_t nphase[3];                  /* general (non rank-1) stencil: number of (kernel row, 4-tap segment) phases of
                                           kind 0 (inner chunk of a window row), 1 (first chunk), 2 (last chunk); each
                                           count is even (an all-zero filler phase pads an odd one)                  */
    int32_t phase[PB_MAX_PHASES + 9];   /* their descriptors, grouped by kind, row-major inside a kind: byte offset of the
                                           segment in the (64 + 2*radius)-wide staged tile | index of its first tap in
                                           gtaps << 16; three pad entries (read ahead, never evaluated) */
} pb_blur_info;

/* ---- context ------------------------------------------------------------------------- */
int pb_version(void);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the
 * device's default stream. */
int pb_create(pb_ctx **out, int device, void *stream);
int pb_destroy(pb_ctx *ctx);
int pb_set_stream(pb_ctx *ctx, void *stream);
int pb_synchronize(pb_ctx *ctx);
const char *pb_last_error_string(pb_ctx *ctx);
void pb_default_options(pb_options *opt);        /* the functional API's defaults, deblurring.py:23-25 */
/* How dense (non rank-1) kernels are evaluated by the reblurring pass.  PB_DENSE_STENCIL: always by the 2-D stencil
 * body (up to 625 multiply-adds per sample, fp32-vector-bound).  PB_DENSE_AUTO (default, min_phases = 16): images whose
 * stencil would run at least `min_phases` live (kernel row, 4-tap segment) phases (8 more when the support fits a
 * 4-sample halo; min_phases = 0: every dense point-symmetric kernel) are evaluated per 64 x 64 window in
 * the frequency domain inside LDS (overlap-save; same taps, same boundary models, results agree to fp32 rounding);
 * the others and rank-1 kernels keep the stencil bodies (fp32 planes: one wave per window pair, conv_wfft.hip; fp16 and
 * 8-bit planes, and passes too small to fill the chip: one workgroup per pair, conv_fft.hip).  Replaces nothing in the reference: both are
 * evaluations of filters.convolve2d (filters.py:14-49).  Environment default: PB_DENSE_EVAL=stencil | <min_phases>.
 * (PB_STRIP=1 -- rank-1 kernels of full support through the streaming strip body, conv_strip.hip, an experiment measured
 * slower than the tile body -- exists in `python -m polyblur_amd.build --experimental` builds only; the default library
 * ignores it.  INTEGRATION.md section 5 lists every environment knob.)
 * With the wrap boundary and no edgetaper the three Horner steps of an image's polynomial are ONE filter,
 * a3 K^3 + a2 K^2 + a1 K + b -- the reference's own 'fft' form, deblurring.py:139-169.  The engine measures that filter's
 * halo per axis and takes the polynomial as one window pass with the polynomial's spectrum wherever that costs less than
 * three passes with the kernel's halos (pb_polyblur_batch, pb_inverse_filter, pb_time_inner_loop; per image, decided on
 * the device): same taps, same results to fp32 rounding (fewer roundings), 2 words per sample through HBM instead of 8.
 * PB_POLY1=0 in the environment switches the form off, PB_POLY1=1 restricts it to kernels within the 4-sample halo class
 * (round 3's form); PB_POLY_GAIN / PB_POLY_MIN_AREA tune the cost model (csrc/khat.h).  pb_body_selection reports the
 * choice.  All of these variables are read when the context is created. */
typedef enum pb_dense_eval { PB_DENSE_STENCIL = 0, PB_DENSE_AUTO = 1 } pb_dense_eval;
int pb_set_dense_eval(pb_ctx *ctx, int mode, int min_phases);
/* Diagnostics: how the B images of iteration `iteration` of the most recent pb_polyblur_batch call on this context were
 * evaluated (-1: the most recent estimation / reblurring pass of any entry point) --
 * host[6 b + 0..5] = { tile-spectrum body (1) or a stencil body (0), halo class of the workgroup form (-1: a one-pass image
 * whose taps the device did NOT find point-symmetric although the call's class vouched for it -- NaN taps: the output is
 * then whatever the one-pass form makes of them), rank-1 strip flag,
 * whole polynomial in one window pass (1 / 2: on 64 x 64 / 128 x 128 windows) or three Horner steps (0), window halo along x,
 * along y }.  The choice is made
 * on the device from each record (no reference counterpart: filters.convolve2d, filters.py:14-49, has one evaluation);
 * bench.py labels its roofline line with it.  Synchronises the context's stream.                                   */
int pb_body_selection(pb_ctx *ctx, int iteration, int *host, int B);
/* bytes of scratch the context currently holds (for the HBM-footprint report) */
size_t pb_workspace_bytes(pb_ctx *ctx);

/* ---- plain device-memory helpers so a host without torch can drive the engine ------- */
int pb_malloc(pb_ctx *ctx, void **dptr, size_t bytes);
int pb_free(pb_ctx *ctx, void *dptr);
int pb_memcpy_h2d(pb_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous */
int pb_memcpy_d2h(pb_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous */

/* ---- whole pipeline: polyblur_deblurring (deblurring.py:23-96) ------------------------
 * in/out: (B,C,H,W) of `dtype`; out may not alias in.  host_info (optional) receives
 * n_iter*B records, iteration-major.                                                    */
int pb_polyblur_batch(pb_ctx *ctx, const void *in, void *out, int dtype,
                      int B, int C, int H, int W, const pb_options *opt,
                      pb_blur_info *host_info);

/* (H,W,C) interleaved bytes <-> (C,H,W) planar bytes for B images, the layouts either side of
 * utils.to_tensor / utils.to_array (utils.py:8-31) when the pixels stay uint8 on the device.  */
int pb_u8_deinterleave(pb_ctx *ctx, const unsigned char *hwc, unsigned char *chw, int B, int C, int H, int W);
int pb_u8_interleave(pb_ctx *ctx, const unsigned char *chw, unsigned char *hwc, int B, int C, int H, int W);

/* ---- stage entry points (each is also what the pipeline calls) ----------------------- */

/* gaussian_blur_estimation (blur_estimation.py:18-79): gray -> normalise -> spectral
 * gradients -> directional maxima -> direction -> (sigma, rho) -> 25x25 kernel.
 * Writes B records to dev_info (device memory).                                         */
int pb_estimate_blur(pb_ctx *ctx, const void *in, int dtype, int B, int C, int H, int W,
                     const pb_options *opt, pb_blur_info *dev_info);

/* create_gaussian_filter (blur_estimation.py:211-232) + support analysis, for caller-
 * supplied parameters: fills kernel/separable/radius of B device records from host arrays.
 * pb_make_kernels and pb_set_kernels synchronise, and the context remembers which bodies of the reblurring pass
 * these B records need (later passes on them skip the launch no image needs).  Records are to be rewritten through
 * this API only (pb_make_kernels, pb_set_kernels, pb_estimate_blur, pb_make_separable_kernels, pb_memcpy_h2d -- each
 * makes the context forget); after a raw copy into them call pb_set_dense_eval, which forgets everything.         */
int pb_make_kernels(pb_ctx *ctx, int B, const float *host_sigma, const float *host_rho,
                    const float *host_theta_rad, int support, pb_blur_info *dev_info);
/* Same, but the caller supplies arbitrary 25x25 taps (host, B*625 floats), in the orientation of the reference's kernel
 * tensors.  The reference applies such a kernel as a correlation under method='direct' (F.conv2d, filters.py:40-49) and as a
 * true circular convolution under method='fft' (K = p2o(kernel), filters.py:33-36) -- the same thing for point-symmetric
 * taps only.  The stage entry points follow it: with PB_ZERO the taps as given, with PB_WRAP -- where they are not
 * point-symmetric -- their point reflection (a second set of records built on the fly).                                 */
int pb_set_kernels(pb_ctx *ctx, int B, const float *host_taps, int support, pb_blur_info *dev_info);

/* method='direct_separable' (pb_options.separable_approx): from B records that hold (sigma, rho, theta) build the two
 * correlation kernels of the x-t separable approximation -- dev_sep[b] the 1-D pass, dev_sep[B + b] the oblique pass
 * with linear interpolation (intent of separable_gaussian2d.cpp:91-183; definition at pb_options.separable_approx).
 * ker_size: odd, 3 .. 25 (0 means 25); an even size -- the reference's off-centre grid, filters.py:78 -- or a larger one
 * returns PB_ERR_UNSUPPORTED.                                                                                          */
int pb_make_separable_kernels(pb_ctx *ctx, int B, const pb_blur_info *dev_info, pb_blur_info *dev_sep, int support,
                              int ker_size);

/* filters.fourier_gradients (filters.py:159-186) on P = B*C planes of H x W float32.
 * gx or gy may be NULL.  The transform keeps whole image lines in LDS when they fit -- up to 20480
 * samples when every prime factor is <= 7, up to 8192 otherwise: pb_fft_length_supported() == 1 --
 * and runs the same stages on a line buffer in device memory (context scratch, at most 256 MB; several
 * times slower per sample) for longer lines up to 65536 samples: == 2.  Beyond that: 0, and every
 * entry point that estimates blur or removes halos returns PB_ERR_UNSUPPORTED (the reference's
 * torch.fft takes any size). */
int pb_fft_length_supported(int n);                 /* 0, 1 or 2 (see above); needs no context */
int pb_fourier_gradients(pb_ctx *ctx, const float *planes, int P, int H, int W,
                         float *gx, float *gy);

/* inverse_filtering_rank3 (deblurring.py:211-239): replicate pad -> [edgetaper] ->
 * polynomial (three fused separable / general Gaussian stencil passes) -> crop ->
 * [halo masking with grad0 = gradients of the original image] -> clamp.
 * grad0_x/grad0_y: (B,C,H,W) float32, required iff remove_halo.                          */
int pb_inverse_filter(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                      const pb_blur_info *dev_info, float alpha, float beta, int boundary,
                      int edgetaping, int remove_halo, const float *grad0_x, const float *grad0_y);

/* filters.convolve2d on an already padded (B,C,Hp,Wp) float32 image: one reblurring pass
 * out = K * in with the given outer boundary (filters.py:14-37).                          */
int pb_convolve2d(pb_ctx *ctx, const float *in, float *out, int B, int C, int Hp, int Wp,
                  const pb_blur_info *dev_info, int boundary);

/* edgetaper.edgetaper (edgetaper.py:26-33) on a padded float32 image, per-image maximum. */
int pb_edgetaper(pb_ctx *ctx, const float *in, float *out, int B, int C, int Hp, int Wp,
                 const pb_blur_info *dev_info, int boundary);

/* halo_masking (deblurring.py:193-208), bug-compatible.  All (B,C,H,W) float32.           */
int pb_halo_mask(pb_ctx *ctx, const float *x, const float *y, const float *grad0_x,
                 const float *grad0_y, float *out, int B, int C, int H, int W);

/* domain_transform.recursive_filter (domain_transform.py:6-63; native twin RF.cpp:43-92).
 * joint may be NULL (filter guided by itself).                                           */
int pb_dt_recursive_filter(pb_ctx *ctx, const void *in, const void *joint, void *out, int dtype,
                           int B, int C, int H, int W, float sigma_s, float sigma_r, int num_iterations);

/* normalized_convolution (NC.cpp:143-204), the domain-transform variant built on box filters in the
 * transformed domain; single-image semantics of the reference applied per image, any C.  H, W <= 8190. */
int pb_dt_normalized_convolution(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                                 float sigma_s, float sigma_r, int num_iterations);

/* filters.bilateral_filter (filters.py:107-148), 5x5, sigma_spatial=5, sigma_color=0.1. */
int pb_bilateral5(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W);

/* ---- patch decomposition with windowed overlap-add (PolyblurDeblurring(patch_decomposition=True),
 * deblurring.py:269-340; "next" row 1 of SURVEY 8f).  The image is (virtually) replicate-padded by
 * (pad_top, pad_left) to a grid of n_i x n_j patches of ph x pw with strides (step_h, step_w).
 * pb_extract_patches gathers patches [first, first+count) of every image into
 * patches[(n-first)*B + b, c, y, x]; pb_overlap_add blends ALL n_i*n_j restored patches (laid out
 * [n*B + b, c, y, x]) with the separable window dev_win_y (x) dev_win_x, normalises by the summed
 * window (+1e-8), clamps to [0,1] and writes the H x W result.                               */
int pb_extract_patches(pb_ctx *ctx, const void *img, void *patches, int dtype, int B, int C, int H, int W,
                       int ph, int pw, int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left,
                       int first, int count);
int pb_overlap_add(pb_ctx *ctx, const void *patches, void *out, int dtype, int B, int C, int H, int W,
                   int ph, int pw, int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left,
                   const float *dev_win_y, const float *dev_win_x);

/* ---- timing hooks used by bench.py ------------------------------------------------------
 * Runs only the polynomial inner loop (three stencil passes, the SURVEY 8d "inner loop")
 * `reps` times on resident data and returns the average milliseconds per repetition
 * measured with hipEvents on the context's stream.                                       */
int pb_time_inner_loop(pb_ctx *ctx, const void *in, void *out, int dtype, int B, int C, int H, int W,
                       const pb_blur_info *dev_info, float alpha, float beta, int boundary,
                       int reps, float *host_ms);

/* Per-kernel-class device timing with hipEvents on the context's stream.  Between begin and
 * end every launch is bracketed by two events; end synchronises and returns, per tag, the
 * summed milliseconds and the launch count (arrays of PB_PROF_NTAGS).                        */
#define PB_PROF_NTAGS 10
typedef enum pb_prof_tag {
    PB_PROF_CONV = 0,        /* stencil pass (one Horner step / taper blend) */
    PB_PROF_GRAY = 1,        /* gray + min/max */
    PB_PROF_GRAD_ROWS = 2,   /* row spectral derivative */
    PB_PROF_GRAD_COLS = 3,   /* column spectral derivative (+ directional maxima) */
    PB_PROF_PARAMS = 4,      /* parameter / kernel generation */
    PB_PROF_HALO = 5,        /* halo masking + its reductions */
    PB_PROF_PREFILTER = 6,   /* bilateral / domain transform / recombination */
    PB_PROF_OTHER = 7,
    PB_PROF_CONV_FUSED = 8,  /* reserved (an experiment of round 2 used it) */
    PB_PROF_CONV_FFT = 9     /* the same pass for dense kernels: tile-spectrum body (conv_fft.hip) */
} pb_prof_tag;
int pb_profile_begin(pb_ctx *ctx);
int pb_profile_end(pb_ctx *ctx, float *host_ms, int *host_count);

/* ---- batch scatter / gather over RCCL (one process per GPU) ----------------------------------
 * The reference has no communication (its only batching is the sequential patch-group loop of
 * deblurring.py:310-336).  Images are independent, so the one exchange the engine needs is: the batch
 * lives on a root rank, every rank deblurs a contiguous shard, the results return to the root.  These
 * entry points give that to a host without torch.distributed (the Python layer uses torch's process
 * group, polyblur_amd/distributed.py): grouped ncclSend / ncclRecv on the context's stream, xGMI
 * inside a node.  RCCL (librccl.so) is loaded when the first id / communicator is made.
 *
 *   rank 0:      pb_comm_unique_id(id);  ship the 128 bytes to the other ranks (file, socket, MPI ...)
 *   every rank:  pb_comm_init(&comm, ctx, rank, world, id);        (world == 1: id may be NULL)
 *                pb_comm_shard(B, world, rank, &first, &count);    contiguous shards, the first B % world get one more
 *                pb_comm_scatter(comm, root_batch, shard, dtype, B, C, H, W, root);
 *                pb_polyblur_batch(ctx, shard, shard_out, dtype, count, C, H, W, &opt, NULL);
 *                pb_comm_gather(comm, shard_out, root_batch_out, dtype, B, C, H, W, root);
 * root_batch is read / written on the root only (NULL elsewhere); a rank whose shard is empty may pass
 * NULL for its shard.  Everything is asynchronous on the context's stream.                              */
#define PB_COMM_ID_BYTES 128
typedef struct pb_comm pb_comm;
int pb_comm_shard(int B, int world, int rank, int *first, int *count);
int pb_comm_unique_id(unsigned char *id);
int pb_comm_init(pb_comm **comm, pb_ctx *ctx, int rank, int world, const unsigned char *id);
int pb_comm_destroy(pb_comm *comm);
int pb_comm_scatter(pb_comm *comm, const void *root_batch, void *shard, int dtype, int B, int C, int H, int W, int root);
int pb_comm_gather(pb_comm *comm, const void *shard, void *root_batch, int dtype, int B, int C, int H, int W, int root);
/* The overlapped form of scatter -> deblur -> gather (what polyblur_amd/distributed.py:deblur_from_root does over
 * torch.distributed; the reference's analogue is the sequential patch-group loop of deblurring.py:310-336): the batch
 * lives on `root` and travels in CHUNKS of k consecutive images -- exchange step t is one grouped ncclSend / ncclRecv
 * operation on a stream of its own in which the root sends chunk t of every peer's shard and receives result chunk t - 2
 * from every peer --, every rank deblurs chunk t - 1 meanwhile (ONE pb_polyblur_batch call of k images: a lone 1080p image
 * costs twice what it costs inside a batch), and the results land in root_out on the root.  Called by every rank with the
 * same B, C, H, W, dtype, options, root and chunk; root_batch / root_out are read / written on the root only (NULL
 * elsewhere).  Returns with the context's stream behind every transfer.  k: pb_comm_set_chunk -- 0 (the default) = image by
 * image (k = 1: the chunked exchange stays opt-in until it has run on two GPUs), k >= 1 = that many images per step,
 * PB_COMM_CHUNK_AUTO = pb_comm_default_chunk = about sqrt(largest peer shard / 2): 4 for 32 images per GPU, 1 for one.
 *   On an error between two steps (a failed pb_polyblur_batch, a HIP error) every remaining step is still posted -- moving
 * buffers whose content no longer matters -- so that no peer is left waiting for a matching send / recv; the first error
 * is returned once the exchange stream has been joined.
 *   EXPERIMENTAL for world > 1: the exchange has run over gloo (2 and 3 ranks, bit-exact) and in a world of one on an
 * MI355X; no grouped ncclSend / ncclRecv of it has executed on two GPUs yet (tests/test_gpu_rccl.py runs it on first
 * contact with a node that shows more than one).
 * pb_comm_plan_steps(_chunked) / pb_comm_plan(_chunked) state the order of a step's operations -- chunked: ops[4 i + 0..3] =
 * { 1 send | 0 recv, peer, first image, images }; image by image (k = 1): ops[3 i + 0..2] = { 1 send | 0 recv, peer, image
 * index } -- at most 2 (world - 1) of them, the order both sides enumerate them in; host-only, no GPU needed.   */
int pb_comm_deblur_from_root(pb_comm *comm, const void *root_batch, void *root_out, int dtype, int B, int C, int H, int W,
                             const pb_options *opt, int root);
#define PB_COMM_CHUNK_AUTO (-1)
int pb_comm_set_chunk(pb_comm *comm, int chunk);
int pb_comm_default_chunk(int B, int world, int root);
int pb_comm_plan_steps(int B, int world, int root);
int pb_comm_plan(int B, int world, int root, int rank, int step, int *ops, int *n_ops);
int pb_comm_plan_steps_chunked(int B, int world, int root, int chunk);
int pb_comm_plan_chunked(int B, int world, int root, int rank, int step, int chunk, int *ops, int *n_ops);

#ifdef __cplusplus
}
#endif
#endif /* POLYBLUR_HIP_H */
