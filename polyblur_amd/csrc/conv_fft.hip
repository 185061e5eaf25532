// Dense (non rank-1) reblurring kernels evaluated per tile in the frequency domain.
//
// Same pass, same operands and same result as the general body of conv.hip -- one Horner step  t <- K*t + coef*x  of the
// polynomial deconvolution (reference deblurring.py:122-138 / :141-169), or one edgetaper blend (edgetaper.py:30-32), on
// the replicate-padded domain with the reference's boundary models (filters.py:14-49) -- but the exact 2-D stencil of a
// workgroup's tile is evaluated as a CIRCULAR correlation of a 64 x 64 window held in LDS: forward 2-D DFT, product with
// the kernel's 64 x 64 spectrum, inverse DFT, of which only the samples at least R from the window's edge (those no tap
// reaches around the wrap) are kept -- overlap-save.  Nothing leaves LDS between the two transforms; HBM sees what the
// stencil pass sees (a window read, the x operand, one store).  The 625 multiply-adds per sample of the dense stencil
// (fp32-vector-bound: 0.30 ms per 4K launch) become ~110 flops per sample: 0.09 ms per 4K launch.
//
// * Two real tiles ride one complex transform: z = A + iB for two horizontally adjacent windows.  The kernel is real,
//   so K (*) z = K (*) A + i K (*) B -- no real-to-complex packing or unpacking step exists.
// * 64 = 8 x 8: every 1-D pass is two radix-8 stages (pbfft::dft_small<8>, complex values in aligned register pairs,
//   packed adds).  Decimation in frequency forward -- position 8a + b of a transformed axis holds frequency a + 8b --
//   and the mirrored stages backward, so no reordering pass exists either; the spectrum of the kernel is laid out in
//   the same permuted order by khat_kernel.
// * Stage order: columns (stage 1 straight from global memory, stage 2), rows (stage 1, stage 2 x spectrum x inverse
//   stage 2 in registers), rows^-1 stage 1, columns^-1 (stage 2, stage 1 straight into the epilogue and global memory).
//   Six LDS round trips per window pair.  The first and last stages touch global memory with 64 consecutive window
//   columns per wave (256-byte segments).
// * LDS rows are 65 complex values long: every row-direction access (stride 8 or contiguous 8 per lane) and every
//   column-direction access is bank-conflict-free for ds_read_b64 / ds_write_b64.
// * 512 threads per window pair, one radix-8 butterfly per thread and stage; all threads of a wave share n2 (stage 1)
//   or k1 (stage 2) -- the wave number -- so the seven inter-stage twiddles W64^(n2 k) are wave-uniform and live in
//   scalar registers.  One workgroup per window pair, four workgroups (32 waves) per CU.
// * The kernels are point-symmetric (every Gaussian the estimator builds is, bit for bit), so their spectrum is real:
//   16 KB per image, 8 values per thread, requested -- like the x operand of the epilogue -- before the first stage.
//   Caller-supplied taps that are not point-symmetric keep the stencil body (khat_kernel decides per image).
// * The interior path addresses planes through buffer descriptors (32-bit offsets; halo rows and columns get an
//   out-of-range offset instead of a branch); LDS reads are single ds_read_b64 from inline assembly.
//
// The window keeps R = 4, 8 or 12 samples of halo (the record's radius class rounded up to a multiple of 4: 16-byte
// aligned windows), i.e. 56, 48 or 40 outputs per side; every image picks its own on the device, like its body.
// No MFMA, no library FFT; the transforms exist only inside a workgroup's LDS.

#include "conv_fft_common.h"
#include "khat.h"

namespace {

// ---------------------------------------------------------------------------------------------
// spectrum of every image's kernel (eight workgroups per image)
// ---------------------------------------------------------------------------------------------
// khat[py][px] = (1/4096) sum_{u,v} k[u][v] cos(2 pi (fy (u-12) + fx (v-12)) / 64),  f = (p >> 3) + 8 (p & 7):
// the spectrum of a point-symmetric real kernel -- every Gaussian the estimator builds is one, bit for bit
// (k[12+u][12+v] == k[12-u][12-v]) -- is real, so correlation and convolution coincide and a thread's 16 spectrum
// values fit 16 registers.  Laid out in the transforms' permuted order, both transforms' normalisation folded in.  Only
// the taps inside the record's support box count, exactly as in the stencil body.  Accumulated in double with a
// double cosine table (the spectrum is then the correctly rounded fp32 one: with fp32 table values the pass loses 1e-6 of
// agreement with the stencil).  Caller-supplied taps that are not point-symmetric keep the stencil body.
static_assert(KH_NT == KH_THREADS && FT_N == KH_FT_N, "khat.h is written for these");
__global__ __launch_bounds__(KH_NT) void khat_kernel(const pb_blur_info *infos, float *khat, pb_fft_sel *sel, int min_phases, const PolySpec ps) {
    khat_body(infos + blockIdx.x, khat + (long)blockIdx.x * PB_KHAT_STRIDE, sel + blockIdx.x, min_phases, (int)blockIdx.y, ps);
}

// ---------------------------------------------------------------------------------------------
// the pass
// ---------------------------------------------------------------------------------------------
// one sample of the source at padded coordinates (py, px) under the pass's boundary model
template <typename T>
__device__ __forceinline__ float load_elem(const ConvPass &a, const T *plane, int py, int px) {
    const int iy = map_axis(py, a.H, a.in_kind, a.boundary, a.pad), ix = map_axis(px, a.W, a.in_kind, a.boundary, a.pad);
    if ((iy | ix) < 0) return 0.f;
    return pb_ld(plane + (long)iy * a.in_pitch + ix);
}
// the x operand of the output at padded (py, px): rows and columns of a virtual source clamp
template <typename TX>
__device__ __forceinline__ float load_x1(const ConvPass &a, const TX *xpl, int py, int px) {
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int xr = virt ? min(max(py - a.pad, 0), a.H - 1) : py;
    const int xc = virt ? min(max(px - a.pad, 0), a.W - 1) : px;
    return pb_ld(xpl + (long)xr * a.x_pitch + xc);
}
// epilogue + store of one output (the caller has checked that (py, px) lies in the output region)
template <typename TOut>
__device__ __forceinline__ void finish1(const ConvPass &a, const pb_blur_info *info, TOut *opl, int py, int px, float acc, float xv) {
    float v;
    if (a.epilogue == EPI_TAPER) {
        const float al = taper_weight(info->acorr_y, py, a.H + 2 * a.pad) * taper_weight(info->acorr_x, px, a.W + 2 * a.pad);
        v = al * xv + (1.f - al) * acc;
    } else {
        v = a.scale * acc + a.coef * xv;
    }
    if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    pb_st(opl + (long)(py - oo) * a.out_pitch + (px - oo), v);
}


// Thread mapping (512 threads, wave w = 0..7, lane = 0..63).  Every 1-D pass of 64 points is two radix-8 stages over the
// index split  n = 8 n1 + n2 -> k = k1 + 8 k2  (position 8 k1 + k2 of a transformed axis holds frequency k1 + 8 k2):
//   stage 1:  the thread of (line = lane, n2 = w) takes elements 8 n1 + n2 of its line, transforms over n1 and multiplies
//             by the inter-stage twiddles W64^(n2 k1) -- wave-uniform, so they sit in scalar registers;
//   stage 2:  the thread of (line = lane, k1 = w) takes elements 8 k1 .. 8 k1 + 7 and transforms over n2.
// In the column passes `line` is the window column (64 consecutive columns per wave: 256-byte global segments in the first
// and last stage, consecutive LDS addresses in between), in the row passes the window row (row pitch 65: rows 0..31 of a
// half wave fall into 32 different 8-byte bank slots).
struct Twiddles { cf w[8]; };
__device__ __forceinline__ Twiddles load_twiddles(int n2u) {
    Twiddles t;
    const PB_CONSTANT float *tab = as_constant(reinterpret_cast<const float *>(kW64));       // scalar loads: n2u is uniform
    t.w[0] = (cf){1.f, 0.f};
#pragma unroll
    for (int k = 1; k < 8; ++k) { const int m = (n2u * k) & 63; t.w[k] = (cf){tab[2 * m], tab[2 * m + 1]}; }
    return t;
}

// Border windows: rows and columns are mapped through the boundary model once per window row / column of the thread
// (10 maps for its 16 samples), then the samples are fetched one by one.
template <typename TIn>
__device__ __forceinline__ void stage1_mapped(const ConvPass &a, const TIn *ipl, int wy0, int wxA, int wxB, bool hasB, float2 *Z,
                                              const Twiddles &tw, int n2, int x) {
    const int ixa = map_axis(wxA + x, a.W, a.in_kind, a.boundary, a.pad);
    const int ixb = hasB ? map_axis(wxB + x, a.W, a.in_kind, a.boundary, a.pad) : -1;
    cf v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int iy = map_axis(wy0 + 8 * n1 + n2, a.H, a.in_kind, a.boundary, a.pad);
        const TIn *row = ipl + (long)max(iy, 0) * a.in_pitch;
        v[n1] = (cf){(iy | ixa) >= 0 ? pb_ld(row + ixa) : 0.f, (iy | ixb) >= 0 ? pb_ld(row + ixb) : 0.f};
    }
    pbfft::dft_small<8>(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) Z[(8 * k + n2) * FT_P + x] = pbfft::to_f2(k ? cmul_s(v[k], tw.w[k]) : v[0]);
}
// Border / taper epilogues, one output at a time.
template <typename TX, typename TOut>
__device__ __forceinline__ void epilogue_mapped(const ConvPass &a, const pb_blur_info *info, const TX *xpl, TOut *opl, int wy0,
                                                int wxA, int R, bool hasB, const float2 *Z, const Twiddles &tw, int n2, int x) {
    const OutRegion rg = out_region(a);
    const int wxB = wxA + FT_N - 2 * R;
    const bool colA = x >= R && x < FT_N - R && wxA + x < rg.x_hi;
    const bool colB = colA && hasB && wxB + x < rg.x_hi;
    cf v[8];
    lds_read8<8 * FT_P * 8>(v, Z + n2 * FT_P + x);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul_conj_s(v[k], tw.w[k]);
    idft8(v);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 8 * n1 + n2, py = wy0 + i;
        const bool rowok = i >= R && i < FT_N - R && py < rg.y_hi;
        if (rowok && colA) finish1<TOut>(a, info, opl, py, wxA + x, v[n1].x, load_x1<TX>(a, xpl, py, wxA + x));
        if (rowok && colB) finish1<TOut>(a, info, opl, py, wxB + x, v[n1].y, load_x1<TX>(a, xpl, py, wxB + x));
    }
}

// One window pair.  Z: the workgroup's LDS tile; kp: the image's spectrum, [x position][y position].
template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ void window_pair(const ConvPass &a, const pb_blur_info *info, int plane, int ty, int pxi, int R, float2 *Z,
                                            const float *kp) {
    const int T = FT_N - 2 * R;
    const OutRegion rg = out_region(a);
    const int wy0 = rg.y_lo + ty * T - R;                       // window origin, padded coordinates
    const int wxA = rg.x_lo + 2 * pxi * T - R, wxB = wxA + T;
    const bool hasB = wxB + R < rg.x_hi;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // n2 in the stage-1 passes, k1 in the stage-2 passes
    const Twiddles tw = load_twiddles(w);
    // the 8 spectrum values this thread multiplies in the centre stage: window row `lane`, columns 8 w .. 8 w + 7
    float kh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kh[j] = kp[(8 * w + j) * FT_N + lane];
    // Fast epilogue: both tiles complete, inside the output region, their x operand addressed without clamping, plain Horner
    // epilogue -- plane descriptors + 32-bit offsets; the halo rows (wave-uniform) and columns get an out-of-range offset.
    // Its x operand is requested here, at the very start: it arrives while the transforms run.
    const int xo = a.x_kind == SRC_VIRTUAL ? a.pad : 0, oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const int oy0 = wy0 + R, oxA = wxA + R;
    const bool fast = a.epilogue == EPI_HORNER && hasB && oy0 + T <= rg.y_hi && oxA + 2 * T <= rg.x_hi && oy0 - xo >= 0 &&
                      oxA - xo >= 0 && oy0 + T - xo <= (a.x_kind == SRC_VIRTUAL ? a.H : a.H + 2 * a.pad) &&
                      oxA + 2 * T - xo <= (a.x_kind == SRC_VIRTUAL ? a.W : a.W + 2 * a.pad);
    const bool colok = lane >= R && lane < FT_N - R;
    float xa[8], xb[8];
    if (fast) {
        const brsrc rx = plane_rsrc(xpl, a.x_plane);
        const unsigned xstep = 8u * (unsigned)a.x_pitch * (unsigned)sizeof(TX);
        const int txb = T * (int)sizeof(TX);
        const unsigned xoff = colok ? ((unsigned)(wy0 + w - xo) * (unsigned)a.x_pitch + (unsigned)(wxA + lane - xo)) * (unsigned)sizeof(TX) : kNoAccess;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const bool rowok = 8 * n1 + w >= R && 8 * n1 + w < FT_N - R;
            const unsigned o = rowok ? xoff + n1 * xstep : kNoAccess;
            xa[n1] = BufIO<TX>::ld(rx, o, 0); xb[n1] = BufIO<TX>::ld(rx, o, txb);
        }
    }

    // ---- columns, stage 1, straight from global memory: window rows 8 n1 + w of column `lane` ----
    {
        const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
        const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
        const bool inside = wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && hasB;
        if (inside) {
            // interior pair: plane descriptor + 32-bit offsets
            const brsrc rin = plane_rsrc(ipl, a.in_plane);
            const unsigned step = 8u * (unsigned)a.in_pitch * (unsigned)sizeof(TIn);
            const int tb = T * (int)sizeof(TIn);
            const unsigned off = ((unsigned)(wy0 - lo + w) * (unsigned)a.in_pitch + (unsigned)(wxA - lo + lane)) * (unsigned)sizeof(TIn);
            cf v[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1] = (cf){BufIO<TIn>::ld(rin, off + n1 * step, 0), BufIO<TIn>::ld(rin, off + n1 * step, tb)};
            pbfft::dft_small<8>(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) Z[(8 * k + w) * FT_P + lane] = pbfft::to_f2(k ? cmul_s(v[k], tw.w[k]) : v[0]);
        } else {
            stage1_mapped<TIn>(a, ipl, wy0, wxA, wxB, hasB, Z, tw, w, lane);
        }
    }
    __syncthreads();
    // ---- columns, stage 2: rows 8 w .. 8 w + 7 of column `lane` ----
    {
        float2 *p = Z + (8 * w) * FT_P + lane;
        cf v[8];
        lds_read8<FT_P * 8>(v, p);
        pbfft::dft_small<8>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j * FT_P] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- rows, stage 1: columns 8 n1 + w of row `lane` ----
    {
        float2 *p = Z + lane * FT_P + w;
        cf v[8];
        lds_read8<64>(v, p);
        pbfft::dft_small<8>(v);
#pragma unroll
        for (int k = 0; k < 8; ++k) p[8 * k] = pbfft::to_f2(k ? cmul_s(v[k], tw.w[k]) : v[0]);
    }
    __syncthreads();
    // ---- rows, stage 2 -> x spectrum -> inverse stage 2: columns 8 w .. 8 w + 7 of row `lane` ----
    {
        float2 *p = Z + lane * FT_P + 8 * w;
        cf v[8];
        lds_read8<8>(v, p);
        pbfft::dft_small<8>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * kh[j];
        idft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- rows, inverse stage 1 ----
    {
        float2 *p = Z + lane * FT_P + w;
        cf v[8];
        lds_read8<64>(v, p);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul_conj_s(v[k], tw.w[k]);
        idft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[8 * j] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- columns, inverse stage 2 ----
    {
        float2 *p = Z + (8 * w) * FT_P + lane;
        cf v[8];
        lds_read8<FT_P * 8>(v, p);
        idft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j * FT_P] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- columns, inverse stage 1, into the epilogue: window rows 8 n1 + w of column `lane` ----
    if (!fast) {
        epilogue_mapped<TX, TOut>(a, info, xpl, opl, wy0, wxA, R, hasB, Z, tw, w, lane);
        return;
    }
    const float sc = a.scale, cfx = a.coef;
    const bool cl = a.clamp01 != 0;
    const brsrc ro = plane_rsrc(opl, a.out_plane);
    const unsigned ostep = 8u * (unsigned)a.out_pitch * (unsigned)sizeof(TOut);
    const int tob = T * (int)sizeof(TOut);
    const unsigned ooff = colok ? ((unsigned)(wy0 + w - oo) * (unsigned)a.out_pitch + (unsigned)(wxA + lane - oo)) * (unsigned)sizeof(TOut) : kNoAccess;
    cf v[8];
    lds_read8<8 * FT_P * 8>(v, Z + w * FT_P + lane);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul_conj_s(v[k], tw.w[k]);
    idft8(v);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const bool rowok = 8 * n1 + w >= R && 8 * n1 + w < FT_N - R;
        const unsigned o = rowok ? ooff + n1 * ostep : kNoAccess;
        float ra = fmaf(sc, v[n1].x, cfx * xa[n1]), rb = fmaf(sc, v[n1].y, cfx * xb[n1]);
        if (cl) { ra = fminf(fmaxf(ra, 0.f), 1.f); rb = fminf(fmaxf(rb, 0.f), 1.f); }
        BufIO<TOut>::st(ro, o, 0, ra); BufIO<TOut>::st(ro, o, tob, rb);
    }
}

// One workgroup per window pair.  Workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only): the window
// pairs of every plane are split into eight contiguous runs, one per XCD, so that neighbouring windows share their halos
// in that XCD's L2.  g.slots = pairs per plane and XCD when the tiles are the smallest (40 x 40) -- the grid is sized for
// that; an image with larger tiles leaves the surplus workgroups idle.
template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(FT_NT, 4) void conv_fft_kernel(const ConvPass a, const FftGeom g) {
    extern __shared__ __attribute__((aligned(16))) float2 Z[];
    const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
    const int plane = __builtin_amdgcn_readfirstlane(div_small(s, g.inv_slots)), i = s - plane * g.slots;
    const int img = __builtin_amdgcn_readfirstlane(plane / a.C);
    const PB_CONSTANT pb_fft_sel *sel = as_constant(a.fsel + img);
    if (!sel->use_fft) return;                                  // a stencil body of conv_tile_kernel does this image
    if (!poly_match(a.poly, sel->poly)) return;                 // (one-pass polynomial: see ConvPass.poly)
    const int R = sel->rf, c = (R >> 2) - 1;
    if (R == 0) return;                                         // (a one-pass image with halos of the wave form only: never routed here)
    // This XCD's run of the plane has per <= slots pairs.  They are dealt to the slots evenly -- slot i takes pair
    // floor(i per / slots) when that differs from its successor's -- so that the idle workgroups of an image with larger
    // tiles are sprinkled between the working ones (a run of idle workgroups in front of the next plane's would drain
    // the chip while the dispatcher works through it) and a plane's pairs still run in order (halo rows of the window
    // row above are still in the XCD's L2).  (double: exact for any plane the engine accepts)
    const int per = g.per[c];
    const int j0 = (int)(((double)i * (double)per) / (double)g.slots), j1 = (int)(((double)(i + 1) * (double)per) / (double)g.slots);
    const int local = __builtin_amdgcn_readfirstlane(xcd * per + j0);
    if (__builtin_amdgcn_readfirstlane(j1) == local - xcd * per || local >= g.njobs[c]) return;
    const int ty = __builtin_amdgcn_readfirstlane(div_small(local, g.inv_pairs_x[c])), pxi = local - ty * g.pairs_x[c];
    const ConvPass af = fold_pass(a, a.poly == 2 && sel->poly != 0);
    window_pair<TIn, TX, TOut>(af, a.info + img, plane, ty, pxi, R, Z, a.khat + (long)img * PB_KHAT_STRIDE);
}


template <typename TIn, typename TX, typename TOut>
int launch_fft_typed(pb_ctx *ctx, const ConvPass &p) {
    FftGeom g;
    if (!fft_geometry(p, g)) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: too many windows for the tile-spectrum body");
    const long total = (long)g.slots * p.P;
    hipLaunchKernelGGL((conv_fft_kernel<TIn, TX, TOut>), dim3((unsigned)(8 * total)), dim3(FT_NT), kFftLds, ctx->stream, p, g);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// Spectra + per-image body selection for the B images of `info` (B = P / C).
int pb_khat_buffers(pb_ctx *ctx, int B, float **khat, pb_fft_sel **sel) {
    float *k = static_cast<float *>(pb_scratch(ctx, "conv.khat", sizeof(float) * PB_KHAT_STRIDE * (size_t)B));
    // (one slot of selections per iteration of the call in progress, so that pb_body_selection can report every iteration's)
    pb_fft_sel *s = static_cast<pb_fft_sel *>(pb_scratch(ctx, "conv.fftsel", sizeof(pb_fft_sel) * (size_t)B * PB_SEL_SLOTS));
    if (!k || !s) return PB_ERR_NOMEM;
    s += (size_t)(ctx->sel_slot % PB_SEL_SLOTS) * B;
    ctx->sel_B = B; ctx->sel_last = ctx->sel_slot;
    if (k != ctx->khat_buf) { ctx->khat_buf = k; ctx->khat_owner = nullptr; ctx->khat_B = 0; ctx->khat_by_estimate = false; }   // (the scratch buffer was reallocated)
    *khat = k; *sel = s;
    return PB_OK;
}

int pb_khat2_buffers(pb_ctx *ctx, int B, float **khat, pb_fft_sel **sel) {
    float *k = static_cast<float *>(pb_scratch(ctx, "conv.khat2", sizeof(float) * PB_KHAT_STRIDE * (size_t)B));
    pb_fft_sel *s = static_cast<pb_fft_sel *>(pb_scratch(ctx, "conv.fftsel2", sizeof(pb_fft_sel) * (size_t)B * PB_SEL_SLOTS));
    if (!k || !s) return PB_ERR_NOMEM;
    s += (size_t)(ctx->sel_slot % PB_SEL_SLOTS) * B;          // (one slot per iteration, as pb_khat_buffers)
    if (k != ctx->khat2_buf) { ctx->khat2_buf = k; ctx->khat2_owner = nullptr; ctx->khat2_B = 0; }      // (the scratch buffer was reallocated)
    *khat = k; *sel = s;
    return PB_OK;
}
// whether the second set holds the spectra of these records under this spec (the estimation built them: estimate.hip)
static bool khat2_holds(pb_ctx *ctx, const pb_blur_info *info, int B, const PolySpec &spec) {
    return ctx->khat2_owner && ctx->khat2_owner == info && ctx->khat2_B == B && same_spec(ctx->khat2_spec, spec) &&
           (spec.on != 0 || ctx->khat2_spec.always == spec.always);
}
int pb_build_khat(pb_ctx *ctx, const pb_blur_info *info, int B, float **khat, pb_fft_sel **sel, bool launch) {
    float *k = nullptr; pb_fft_sel *s = nullptr;
    const int rcb = pb_khat_buffers(ctx, B, &k, &s);
    if (rcb) return rcb;
    // (spectra of other records, of fewer records than this pass covers, or of the kernel where the pass wants the polynomial's)
    // (... or selections written to another slot than the one this pass reads)
    // (the estimation may have built what this pass wants into the SECOND set: the polynomial behind an edgetaper)
    if ((!ctx->khat_owner || ctx->khat_owner != info || ctx->khat_B != B || !same_spec(ctx->poly_built, ctx->poly_want)) &&
        ctx->poly_want.on != 0 && khat2_holds(ctx, info, B, ctx->poly_want)) {
        float *k2 = nullptr; pb_fft_sel *s2 = nullptr;
        const int rc2 = pb_khat2_buffers(ctx, B, &k2, &s2);
        if (rc2) return rc2;
        if (khat2_holds(ctx, info, B, ctx->poly_want)) {
            ctx->sel2_mask |= 1u << (ctx->sel_slot % PB_SEL_SLOTS);
            *khat = k2; *sel = s2;
            return PB_OK;
        }
    }
    if (!ctx->khat_owner || ctx->khat_owner != info || ctx->khat_B != B || !same_spec(ctx->poly_built, ctx->poly_want) ||
        ctx->khat_slot != ctx->sel_slot % PB_SEL_SLOTS) launch = true;
    if (launch) {
        ctx->khat_owner = info; ctx->khat_B = B; ctx->khat_by_estimate = false; ctx->poly_built = ctx->poly_want;
        ctx->khat_slot = ctx->sel_slot % PB_SEL_SLOTS;
        ProfScope prof(ctx, PB_PROF_PARAMS);
        hipLaunchKernelGGL(khat_kernel, dim3((unsigned)B, KH_SLICES), dim3(KH_NT), 0, ctx->stream, info, k, s, ctx->fft_min_phases,
                           ctx->poly_want);
        PB_LAUNCH_CHECK();
    }
    *khat = k; *sel = s;
    return PB_OK;
}

// The kernels' own spectra (not the polynomial's) and halos in a SECOND scratch set, every point-symmetric kernel on the
// three-step window form: what the border ring of a zero-boundary polynomial runs its three Horner steps with while the
// first set holds the polynomial's spectra of the interior's one window pass (pb_launch_conv_poly).  One launch, not cached.
int pb_build_khat_ring(pb_ctx *ctx, const pb_blur_info *info, int B, float **khat, pb_fft_sel **sel) {
    float *k = nullptr; pb_fft_sel *s = nullptr;
    const int rcb = pb_khat2_buffers(ctx, B, &k, &s);
    if (rcb) return rcb;
    PolySpec ps = no_poly();
    ps.always = 2;
    if (!khat2_holds(ctx, info, B, ps)) {
        ProfScope prof(ctx, PB_PROF_PARAMS);
        hipLaunchKernelGGL(khat_kernel, dim3((unsigned)B, KH_SLICES), dim3(KH_NT), 0, ctx->stream, info, k, s, ctx->fft_min_phases, ps);
        PB_LAUNCH_CHECK();
        ctx->khat2_owner = info; ctx->khat2_B = B; ctx->khat2_spec = ps;
    }
    *khat = k; *sel = s;
    return PB_OK;
}

bool pb_conv_fft_feasible(const ConvPass &p) { FftGeom g; return fft_geometry(p, g); }

bool pb_conv_fft_types(const ConvPass &p) {
    switch (p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype) {
        case 0: case 1: case 3: case 4: case 9: case 10: case 12: case 13: case 24: case 6: case 8: case 2: return true;
        default: return false;
    }
}

int pb_launch_conv_fft(pb_ctx *ctx, const ConvPass &p) {
    ProfScope prof(ctx, PB_PROF_CONV_FFT);
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    switch (key) {
        case 0: return launch_fft_typed<float, float, float>(ctx, p);
        case 1: return launch_fft_typed<float, float, __half>(ctx, p);
        case 3: return launch_fft_typed<float, __half, float>(ctx, p);
        case 4: return launch_fft_typed<float, __half, __half>(ctx, p);
        case 9: return launch_fft_typed<__half, float, float>(ctx, p);
        case 10: return launch_fft_typed<__half, float, __half>(ctx, p);
        case 12: return launch_fft_typed<__half, __half, float>(ctx, p);
        case 13: return launch_fft_typed<__half, __half, __half>(ctx, p);
        // 8-bit images (see pb_launch_conv)
        case 24: return launch_fft_typed<unsigned char, unsigned char, float>(ctx, p);
        case 6: return launch_fft_typed<float, unsigned char, float>(ctx, p);
        case 8: return launch_fft_typed<float, unsigned char, unsigned char>(ctx, p);
        case 2: return launch_fft_typed<float, float, unsigned char>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "tile-spectrum pass: unsupported dtype combination %d", key);
    }
}
