// The reblurring stencil pass: out = epilogue(K * in, x) on the replicate-padded domain.
//
// One launch evaluates one Horner step  t <- K*t + coef*x  of the polynomial deconvolution
// (reference deblurring.py:122-138 / :141-169) or one edgetaper blend (edgetaper.py:30-32)
// for every plane of the batch.  Replaces filters.convolve2d / conv2d_ (filters.py:14-49),
// utils.pad_with_kernel / crop_with_kernel (utils.py:48-61, folded into index arithmetic)
// and the intent of separable_gaussian2d.cpp:47-88 (x-then-y 1-D Gaussian passes).
//
// Work decomposition (CDNA4): a 256-thread workgroup (4 waves) owns a TH x TW output tile of
// one plane.  The (TH+2R) x (TW+2R) input tile is staged once through LDS with coalesced
// row reads; rank-1 kernels run a row pass LDS->LDS and a column pass LDS->registers
// (both 1-D filters fused in one launch: 3 words of HBM traffic per sample, not 5);
// general (rotated anisotropic) kernels run the exact 2-D stencil from the same LDS tile.
// Every thread produces a 4-wide x PR-high register block so that LDS is read with
// ds_read_b128 and each loaded value feeds up to 4*(2R+1) FMAs.  No MFMA: the pass is
// bound by HBM (separable) or by fp32 VALU (general), never by a dense contraction.
#include "common.h"

namespace {

constexpr int TW = 64;
constexpr int NT = 256;

__device__ __forceinline__ int wrap_idx(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// Load LH x LW floats starting at padded coordinates (py0, px0) into LDS (row pitch LW).
template <typename T, int LH, int LW>
__device__ __forceinline__ void load_tile(float *s, const T *plane, int kind, int pitch, int H, int W,
                                          int py0, int px0, int boundary) {
    const int Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + LH <= Hp && px0 + LW <= Wp;
    if (kind == SRC_VIRTUAL)
        inside = inside && py0 >= PB_PAD && px0 >= PB_PAD && py0 + LH <= PB_PAD + H && px0 + LW <= PB_PAD + W;
    const int tid = threadIdx.x;
    if (inside) {
        const T *base = (kind == SRC_VIRTUAL) ? plane + (long)(py0 - PB_PAD) * pitch + (px0 - PB_PAD)
                                              : plane + (long)py0 * pitch + px0;
#pragma unroll 4
        for (int e = tid; e < LH * LW; e += NT) {
            const int r = e / LW, c = e - r * LW;
            s[e] = pb_ld(base + (long)r * pitch + c);
        }
    } else {
        for (int e = tid; e < LH * LW; e += NT) {
            const int r = e / LW, c = e - r * LW;
            int py = py0 + r, px = px0 + c;
            float v = 0.f;
            bool ok = true;
            if (boundary == PB_WRAP) {
                py = wrap_idx(py, Hp);
                px = wrap_idx(px, Wp);
            } else {
                ok = py >= 0 && py < Hp && px >= 0 && px < Wp;
            }
            if (ok) {
                if (kind == SRC_VIRTUAL) {
                    const int iy = min(max(py - PB_PAD, 0), H - 1), ix = min(max(px - PB_PAD, 0), W - 1);
                    v = pb_ld(plane + (long)iy * pitch + ix);
                } else {
                    v = pb_ld(plane + (long)py * pitch + px);
                }
            }
            s[e] = v;
        }
    }
}

__device__ __forceinline__ float taper_weight(const float *ac, int p, int n) {
    // v[p] = 1 - z[p]/z[0], z = circular autocorrelation with period n-1, z[n-1] := z[0]
    // (edgetaper.py:11-15): non-zero only within 24 samples of either end.
    int m = min(p, n - 1 - p);
    float z = (m < PB_KSIZE) ? ac[m] : 0.f;
    return 1.f - z / ac[0];
}

// Epilogue + store of one output row segment of 4 samples.
template <typename TX, typename TOut>
__device__ __forceinline__ void finish4(const ConvPass &a, const pb_blur_info *info, const TX *xpl, TOut *opl,
                                        int py, int px, float4 acc) {
    const int H = a.H, W = a.W;
    const int Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    int y_lo, y_hi, x_lo, x_hi;
    if (a.out_kind == OUT_INTERIOR) { y_lo = PB_PAD; y_hi = PB_PAD + H; x_lo = PB_PAD; x_hi = PB_PAD + W; }
    else { y_lo = 0; y_hi = Hp; x_lo = 0; x_hi = Wp; }
    if (py < y_lo || py >= y_hi) return;
    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
    float ty = 1.f;
    if (a.epilogue == EPI_TAPER) ty = taper_weight(info->acorr_y, py, Hp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qx = px + i;
        if (qx < x_lo || qx >= x_hi) continue;
        float xv;
        if (a.x_kind == SRC_VIRTUAL) {
            const int iy = min(max(py - PB_PAD, 0), H - 1), ix = min(max(qx - PB_PAD, 0), W - 1);
            xv = pb_ld(xpl + (long)iy * a.x_pitch + ix);
        } else {
            xv = pb_ld(xpl + (long)py * a.x_pitch + qx);
        }
        float v;
        if (a.epilogue == EPI_TAPER) {
            const float al = ty * taper_weight(info->acorr_x, qx, Wp);
            v = al * xv + (1.f - al) * av[i];
        } else {
            v = a.scale * av[i] + a.coef * xv;
        }
        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        if (a.out_kind == OUT_INTERIOR) pb_st(opl + (long)(py - PB_PAD) * a.out_pitch + (qx - PB_PAD), v);
        else pb_st(opl + (long)py * a.out_pitch + qx, v);
    }
}

// ---- rank-1 kernels: row pass into LDS, column pass into registers -----------------------
template <typename TIn, typename TX, typename TOut, int TH, int R>
__device__ __forceinline__ void body_separable(const ConvPass &a, const pb_blur_info *info, const TIn *ipl,
                                               const TX *xpl, TOut *opl, int oy0, int ox0, float *smem) {
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R, NTAP = 2 * R + 1;
    constexpr int G = TW / 4;                 // 4-wide column groups per tile row
    constexpr int PR = TH * G / NT;           // output rows per thread
    static_assert(TH * G % NT == 0 && PR >= 1, "tile shape");
    float *s_in = smem;
    float *s_row = smem + LH * LW;
    load_tile<TIn, LH, LW>(s_in, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    float kx[NTAP], ky[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        kx[t] = info->kx[PB_KRAD - R + t];
        ky[t] = info->ky[PB_KRAD - R + t];
    }
    __syncthreads();
    const int tid = threadIdx.x;
    // row pass: LH rows x G groups
    for (int item = tid; item < LH * G; item += NT) {
        const int rr = item / G, g = item - rr * G;
        const float4 *src = reinterpret_cast<const float4 *>(s_in + rr * LW + 4 * g);
        float seg[4 + 2 * R];
#pragma unroll
        for (int q = 0; q < 1 + R / 2; ++q) {
            const float4 v = src[q];
            seg[4 * q] = v.x; seg[4 * q + 1] = v.y; seg[4 * q + 2] = v.z; seg[4 * q + 3] = v.w;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            o.x = fmaf(kx[t], seg[t], o.x);
            o.y = fmaf(kx[t], seg[t + 1], o.y);
            o.z = fmaf(kx[t], seg[t + 2], o.z);
            o.w = fmaf(kx[t], seg[t + 3], o.w);
        }
        *reinterpret_cast<float4 *>(s_row + rr * TW + 4 * g) = o;
    }
    __syncthreads();
    // column pass
    const int g = tid % G, rg = tid / G;
    float4 acc[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < PR + 2 * R; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(s_row + (rg * PR + i) * TW + 4 * g);
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int t = i - r;
            if (t >= 0 && t < NTAP) {
                acc[r].x = fmaf(ky[t], v.x, acc[r].x);
                acc[r].y = fmaf(ky[t], v.y, acc[r].y);
                acc[r].z = fmaf(ky[t], v.z, acc[r].z);
                acc[r].w = fmaf(ky[t], v.w, acc[r].w);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) finish4<TX, TOut>(a, info, xpl, opl, oy0 + rg * PR + r, ox0 + 4 * g, acc[r]);
}

// ---- general kernels: exact 2-D stencil from the LDS tile --------------------------------
template <typename TIn, typename TX, typename TOut, int TH, int R>
__device__ __forceinline__ void body_general(const ConvPass &a, const pb_blur_info *info, const TIn *ipl,
                                             const TX *xpl, TOut *opl, int oy0, int ox0, float *smem) {
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R, NTAP = 2 * R + 1;
    constexpr int G = TW / 4;
    constexpr int PR = TH * G / NT;
    float *s_in = smem;
    load_tile<TIn, LH, LW>(s_in, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    __syncthreads();
    const int tid = threadIdx.x;
    const int g = tid % G, rg = tid / G;
    float4 acc[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *taps = info->kernel + (PB_KRAD - R) * PB_KSIZE + (PB_KRAD - R);
#pragma unroll 1
    for (int dy = 0; dy < NTAP; ++dy) {
        float k[NTAP];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) k[t] = taps[dy * PB_KSIZE + t];
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const float4 *src = reinterpret_cast<const float4 *>(s_in + (rg * PR + r + dy) * LW + 4 * g);
            float seg[4 + 2 * R];
#pragma unroll
            for (int q = 0; q < 1 + R / 2; ++q) {
                const float4 v = src[q];
                seg[4 * q] = v.x; seg[4 * q + 1] = v.y; seg[4 * q + 2] = v.z; seg[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                acc[r].x = fmaf(k[t], seg[t], acc[r].x);
                acc[r].y = fmaf(k[t], seg[t + 1], acc[r].y);
                acc[r].z = fmaf(k[t], seg[t + 2], acc[r].z);
                acc[r].w = fmaf(k[t], seg[t + 3], acc[r].w);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) finish4<TX, TOut>(a, info, xpl, opl, oy0 + rg * PR + r, ox0 + 4 * g, acc[r]);
}

template <typename TIn, typename TX, typename TOut, int TH>
__global__ __launch_bounds__(NT) void conv_pass_kernel(const ConvPass a, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles = tiles_x * tiles_y;
    const int plane = blockIdx.x / tiles;
    const int t = blockIdx.x - plane * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int org = (a.out_kind == OUT_INTERIOR) ? PB_PAD : 0;
    const int oy0 = org + ty * TH, ox0 = org + tx * TW;
    const pb_blur_info *info = a.info + plane / a.C;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = a.force_full ? PB_KRAD : info->radius;
    if (info->separable) {
        if (R <= 4) body_separable<TIn, TX, TOut, TH, 4>(a, info, ipl, xpl, opl, oy0, ox0, smem);
        else if (R <= 8) body_separable<TIn, TX, TOut, TH, 8>(a, info, ipl, xpl, opl, oy0, ox0, smem);
        else body_separable<TIn, TX, TOut, TH, 12>(a, info, ipl, xpl, opl, oy0, ox0, smem);
    } else {
        if (R <= 4) body_general<TIn, TX, TOut, TH, 4>(a, info, ipl, xpl, opl, oy0, ox0, smem);
        else if (R <= 8) body_general<TIn, TX, TOut, TH, 8>(a, info, ipl, xpl, opl, oy0, ox0, smem);
        else body_general<TIn, TX, TOut, TH, 12>(a, info, ipl, xpl, opl, oy0, ox0, smem);
    }
}

constexpr int TILE_H = 32;

template <typename TIn, typename TX, typename TOut>
int launch_typed(pb_ctx *ctx, const ConvPass &p) {
    constexpr int TH = TILE_H;
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + TW - 1) / TW, tiles_y = (oh + TH - 1) / TH;
    const long blocks = (long)tiles_x * tiles_y * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    constexpr int R = PB_KRAD;
    const size_t lds = sizeof(float) * ((TH + 2 * R) * (TW + 2 * R) + (TH + 2 * R) * TW);
    ProfScope prof(ctx, PB_PROF_CONV);
    hipLaunchKernelGGL((conv_pass_kernel<TIn, TX, TOut, TH>), dim3((unsigned)blocks), dim3(NT), lds, ctx->stream, p,
                       tiles_x, tiles_y);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

int pb_launch_conv(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 4 + p.x_dtype * 2 + p.out_dtype;
    switch (key) {
        case 0: return launch_typed<float, float, float>(ctx, p);
        case 1: return launch_typed<float, float, __half>(ctx, p);
        case 2: return launch_typed<float, __half, float>(ctx, p);
        case 3: return launch_typed<float, __half, __half>(ctx, p);
        case 6: return launch_typed<__half, __half, float>(ctx, p);
        case 7: return launch_typed<__half, __half, __half>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: unsupported dtype combination %d", key);
    }
}
