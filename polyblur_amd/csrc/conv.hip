// The reblurring stencil pass: out = epilogue(K * in, x) on the replicate-padded domain.
//
// One launch evaluates one Horner step  t <- K*t + coef*x  of the polynomial deconvolution
// (reference deblurring.py:122-138 / :141-169) or one edgetaper blend (edgetaper.py:30-32)
// for every plane of the batch.  Replaces filters.convolve2d / conv2d_ (filters.py:14-49),
// utils.pad_with_kernel / crop_with_kernel (utils.py:48-61, folded into index arithmetic)
// and the intent of separable_gaussian2d.cpp:47-88 (x-then-y 1-D Gaussian passes).
//
// Two bodies live in the one kernel; every image picks its own at run time from its
// pb_blur_info record (so a batch may mix them and the host never synchronises):
//
// * rank-1 kernels (theta % 90 == 0 or sigma == rho) -- "streaming" body.  Each WAVE owns a
//   232-column strip segment and walks down it one row at a time: the row is loaded with
//   16-byte coalesced reads (prefetched one row ahead), staged through a 1.1 KB wave-private
//   LDS line, filtered along x (ds_read_b128 + 100 FMA per lane), and scattered along y into
//   25 rotating accumulator rows held in registers; each step emits one finished output row.
//   No workgroup barrier, no second LDS buffer, no vertical re-filtering: per sample the pass
//   moves its 3 algorithmic words of HBM traffic and ~7 words of LDS traffic.
// * general (rotated anisotropic) kernels -- "tile" body: a 256-thread workgroup stages a
//   (64+2R)^2 tile in LDS once and evaluates the exact 2-D stencil from it, 4x4 outputs per
//   thread, the 25-tap rows of the kernel streamed through SGPRs.  fp32 VALU-bound by design.
//
// No MFMA anywhere: the pass is bound by HBM (rank-1) or by the fp32 vector rate (general).
#include "common.h"

namespace {

constexpr int NT = 256;
#define PB_CONSTANT __attribute__((address_space(4)))

// loads through the constant address space become scalar (s_load) when the address is uniform
template <typename T> __device__ __forceinline__ const PB_CONSTANT T *as_constant(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const PB_CONSTANT T *)p;
#pragma clang diagnostic pop
}

// compiler-only ordering of LDS traffic inside one wavefront (no instruction is emitted)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wrap_idx(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// padded coordinate -> source index along one axis, or -1 for "reads as zero"
__device__ __forceinline__ int map_axis(int p, int n_unpadded, int kind, int boundary) {
    const int np = n_unpadded + 2 * PB_PAD;
    if (boundary == PB_WRAP) p = wrap_idx(p, np);
    else if (p < 0 || p >= np) return -1;
    if (kind == SRC_VIRTUAL) return min(max(p - PB_PAD, 0), n_unpadded - 1);
    return p;
}

__device__ __forceinline__ float taper_weight(const float *ac, int p, int n) {
    // v[p] = 1 - z[p]/z[0], z = circular autocorrelation with period n-1, z[n-1] := z[0]
    // (edgetaper.py:11-15): non-zero only within 24 samples of either end.
    const int m = min(p, n - 1 - p);
    const float z = (m < PB_KSIZE) ? ac[m] : 0.f;
    return 1.f - z / ac[0];
}

template <typename T> __device__ __forceinline__ float4 ld4(const T *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<__half>(const __half *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
template <typename T> __device__ __forceinline__ void st4(T *p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<__half>(__half *p, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned *>(&a);
    u.y = *reinterpret_cast<const unsigned *>(&b);
    *reinterpret_cast<uint2 *>(p) = u;
}

// Where the outputs of a pass live, in padded coordinates.
struct OutRegion { int y_lo, y_hi, x_lo, x_hi; };
__device__ __forceinline__ OutRegion out_region(const ConvPass &a) {
    if (a.out_kind == OUT_INTERIOR) return OutRegion{PB_PAD, PB_PAD + a.H, PB_PAD, PB_PAD + a.W};
    return OutRegion{0, a.H + 2 * PB_PAD, 0, a.W + 2 * PB_PAD};
}

// Epilogue + store of 4 horizontally adjacent outputs at padded (py, px..px+3).
template <typename TX, typename TOut>
__device__ __forceinline__ void finish4(const ConvPass &a, const pb_blur_info *info, const TX *xpl, TOut *opl,
                                        const OutRegion &rg, int py, int px, float4 acc) {
    if (py < rg.y_lo || py >= rg.y_hi || px >= rg.x_hi) return;
    const int H = a.H, W = a.W;
    const int Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    float av[4] = {acc.x, acc.y, acc.z, acc.w};
    float xv[4];
    const bool full = px >= rg.x_lo && px + 3 < rg.x_hi;
    // x operand: rows clamp uniformly; a 16-byte load when the four columns are contiguous in the source
    const int xr = (a.x_kind == SRC_VIRTUAL) ? min(max(py - PB_PAD, 0), H - 1) : py;
    const int xc0 = (a.x_kind == SRC_VIRTUAL) ? px - PB_PAD : px;
    const int xcmax = (a.x_kind == SRC_VIRTUAL) ? W : Wp;
    const TX *xrow = xpl + (long)xr * a.x_pitch;
    if (full && xc0 >= 0 && xc0 + 3 < xcmax && ((a.x_pitch | xc0) & 3) == 0) {
        const float4 t = ld4<TX>(xrow + xc0);
        xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = pb_ld(xrow + min(max(xc0 + i, 0), xcmax - 1));
    }
    float ty = 1.f;
    if (a.epilogue == EPI_TAPER) ty = taper_weight(info->acorr_y, py, Hp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if (a.epilogue == EPI_TAPER) {
            const float al = ty * taper_weight(info->acorr_x, px + i, Wp);
            v = al * xv[i] + (1.f - al) * av[i];
        } else {
            v = a.scale * av[i] + a.coef * xv[i];
        }
        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        av[i] = v;
    }
    const int orow = (a.out_kind == OUT_INTERIOR) ? py - PB_PAD : py;
    const int oc0 = (a.out_kind == OUT_INTERIOR) ? px - PB_PAD : px;
    TOut *orow_p = opl + (long)orow * a.out_pitch;
    if (full && ((a.out_pitch | oc0) & 3) == 0) {
        st4<TOut>(orow_p + oc0, make_float4(av[0], av[1], av[2], av[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px + i >= rg.x_lo && px + i < rg.x_hi) pb_st(orow_p + oc0 + i, av[i]);
    }
}

// =============================================================================================
// rank-1 kernels: wave-private streaming body
// =============================================================================================
constexpr int SR = PB_KRAD;                   // the streaming body always evaluates all 25 taps
constexpr int SNT = 2 * SR + 1;
constexpr int SOUT = 256 - 2 * SR;            // 232 output columns per strip: 58 lanes x 4 ...
constexpr int SLANES = SOUT / 4;              // ... so that the 256-sample staged line is ONE float4 per lane
constexpr int SU = 5;                         // rows per unrolled group == prefetch depth
constexpr int SACC = SNT + SU - 1;            // accumulator rows alive inside a group
constexpr int SLINE = 72 * 4;                 // LDS floats per wave (64 float4 + read-ahead slack)

// one source row -> this lane's float4 of the staged line (columns c[0..3]; -1 reads as zero)
template <typename T>
__device__ __forceinline__ float4 load_line(const T *plane, int pitch, int iy, bool fast, const int (&c)[4]) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy < 0) return v;
    const T *row = plane + (long)iy * pitch;
    if (fast) return ld4<T>(row + c[0]);
    if (c[0] >= 0) v.x = pb_ld(row + c[0]);
    if (c[1] >= 0) v.y = pb_ld(row + c[1]);
    if (c[2] >= 0) v.z = pb_ld(row + c[2]);
    if (c[3] >= 0) v.w = pb_ld(row + c[3]);
    return v;
}

// x operand of one output row for this lane (same addressing rules as finish4)
template <typename TX>
__device__ __forceinline__ float4 load_x4(const ConvPass &a, const TX *xpl, const OutRegion &rg, int py, int px) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (py < rg.y_lo || py >= rg.y_hi || px >= rg.x_hi) return r;
    const int H = a.H, W = a.W, Wp = W + 2 * PB_PAD;
    const int xr = (a.x_kind == SRC_VIRTUAL) ? min(max(py - PB_PAD, 0), H - 1) : py;
    const int xc0 = (a.x_kind == SRC_VIRTUAL) ? px - PB_PAD : px;
    const int xcmax = (a.x_kind == SRC_VIRTUAL) ? W : Wp;
    const TX *xrow = xpl + (long)xr * a.x_pitch;
    if (xc0 >= 0 && xc0 + 3 < xcmax && ((a.x_pitch | xc0) & 3) == 0) return ld4<TX>(xrow + xc0);
    r.x = pb_ld(xrow + min(max(xc0, 0), xcmax - 1));
    r.y = pb_ld(xrow + min(max(xc0 + 1, 0), xcmax - 1));
    r.z = pb_ld(xrow + min(max(xc0 + 2, 0), xcmax - 1));
    r.w = pb_ld(xrow + min(max(xc0 + 3, 0), xcmax - 1));
    return r;
}

// epilogue + store with the x operand already in registers
template <typename TOut>
__device__ __forceinline__ void finish4x(const ConvPass &a, const pb_blur_info *info, TOut *opl, const OutRegion &rg,
                                         int py, int px, float4 acc, float4 x) {
    if (py < rg.y_lo || py >= rg.y_hi || px >= rg.x_hi) return;
    const int Hp = a.H + 2 * PB_PAD, Wp = a.W + 2 * PB_PAD;
    float av[4] = {acc.x, acc.y, acc.z, acc.w};
    const float xv[4] = {x.x, x.y, x.z, x.w};
    float ty = 1.f;
    if (a.epilogue == EPI_TAPER) ty = taper_weight(info->acorr_y, py, Hp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if (a.epilogue == EPI_TAPER) {
            const float al = ty * taper_weight(info->acorr_x, px + i, Wp);
            v = al * xv[i] + (1.f - al) * av[i];
        } else {
            v = a.scale * av[i] + a.coef * xv[i];
        }
        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        av[i] = v;
    }
    const int orow = (a.out_kind == OUT_INTERIOR) ? py - PB_PAD : py;
    const int oc0 = (a.out_kind == OUT_INTERIOR) ? px - PB_PAD : px;
    TOut *orow_p = opl + (long)orow * a.out_pitch;
    if (px + 3 < rg.x_hi && ((a.out_pitch | oc0) & 3) == 0) {
        st4<TOut>(orow_p + oc0, make_float4(av[0], av[1], av[2], av[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px + i < rg.x_hi) pb_st(orow_p + oc0 + i, av[i]);
    }
}

template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ void body_stream(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                            TOut *opl, int task, int nsx, int seg_h, float *line) {
    const OutRegion rg = out_region(a);
    const int lane = threadIdx.x & 63;
    const int sy = task / nsx, sx = task - sy * nsx;
    const int x0 = rg.x_lo + sx * SOUT;
    const int y0 = rg.y_lo + sy * seg_h;
    const int rows_out = min(seg_h, rg.y_hi - y0);
    if (rows_out <= 0) return;
    const int H = a.H, W = a.W;
    // taps -> SGPRs.  Rank-1 records carry symmetric marginals (kx[t] == kx[24-t], enforced when the
    // record is built), so 13 + 13 scalars are enough and the whole walk keeps them resident.
    const PB_CONSTANT float *ckx = as_constant(info->kx), *cky = as_constant(info->ky);
    float hx[SR + 1], hy[SR + 1];
#pragma unroll
    for (int t = 0; t <= SR; ++t) { hx[t] = ckx[t]; hy[t] = cky[t]; }
#define KX(t) hx[(t) <= SR ? (t) : 2 * SR - (t)]
#define KY(t) hy[(t) <= SR ? (t) : 2 * SR - (t)]
    // source columns of this lane's float4 of the staged line (fixed for the whole walk)
    int cs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] = map_axis(x0 - SR + 4 * lane + e, W, a.in_kind, a.boundary);
    const bool fast = (a.in_pitch & 3) == 0 && cs[0] >= 0 && (cs[0] & 3) == 0 && cs[1] == cs[0] + 1 &&
                      cs[2] == cs[0] + 2 && cs[3] == cs[0] + 3;
    const int px = (lane < SLANES) ? x0 + 4 * lane : rg.x_hi;      // lanes 58..63 only help loading
    float4 *line4 = reinterpret_cast<float4 *>(line);
    float4 acc[SACC];
#pragma unroll
    for (int t = 0; t < SACC; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_in = rows_out + 2 * SR;
    // software pipeline: SU source rows and SU x rows in flight
    float4 pf[SU], xq[SU];
#pragma unroll
    for (int k = 0; k < SU; ++k) {
        pf[k] = load_line<TIn>(ipl, a.in_pitch, k < n_in ? map_axis(y0 - SR + k, H, a.in_kind, a.boundary) : -1, fast, cs);
        xq[k] = load_x4<TX>(a, xpl, rg, y0 + k - 2 * SR, px);
    }
    for (int i0 = 0; i0 < n_in; i0 += SU) {
#pragma unroll
        for (int s = 0; s < SU; ++s) {
            const int i = i0 + s;
            if (i < n_in) {                                           // uniform
                const float4 cur = pf[s];
                const float4 xcur = xq[s];
                const int inext = i + SU;
                pf[s] = load_line<TIn>(ipl, a.in_pitch, inext < n_in ? map_axis(y0 - SR + inext, H, a.in_kind, a.boundary) : -1,
                                       fast, cs);
                xq[s] = load_x4<TX>(a, xpl, rg, y0 + inext - 2 * SR, px);
                // The staged line is exchanged between LANES of this wave: the LDS unit executes a wave's
                // accesses in order, but the compiler must be told not to move them across each other.
                wave_lds_fence();
                line4[lane] = cur;
                wave_lds_fence();
                float seg[4 + 2 * SR];
#pragma unroll
                for (int q = 0; q < 1 + SR / 2; ++q) {
                    const float4 t = line4[lane + q];
                    seg[4 * q] = t.x; seg[4 * q + 1] = t.y; seg[4 * q + 2] = t.z; seg[4 * q + 3] = t.w;
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < SNT; ++t) {
                    v.x = fmaf(KX(t), seg[t], v.x);
                    v.y = fmaf(KX(t), seg[t + 1], v.y);
                    v.z = fmaf(KX(t), seg[t + 2], v.z);
                    v.w = fmaf(KX(t), seg[t + 3], v.w);
                }
                // input row i feeds output rows i-t with tap ky[t]; inside the group output row o
                // lives in slot o - (i0 - 2R), so row i touches slots s .. s+2R
#pragma unroll
                for (int t = 0; t < SNT; ++t) {
                    const int slot = s + 2 * SR - t;
                    acc[slot].x = fmaf(KY(t), v.x, acc[slot].x);
                    acc[slot].y = fmaf(KY(t), v.y, acc[slot].y);
                    acc[slot].z = fmaf(KY(t), v.z, acc[slot].z);
                    acc[slot].w = fmaf(KY(t), v.w, acc[slot].w);
                }
                const int o = i - 2 * SR;                              // complete now: slot s
                if (o >= 0) finish4x<TOut>(a, info, opl, rg, y0 + o, px, acc[s], xcur);
            }
        }
        // slide the accumulator window down by SU rows
#pragma unroll
        for (int j = 0; j < SACC; ++j) acc[j] = (j + SU < SACC) ? acc[j + SU] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#undef KX
#undef KY
}

// =============================================================================================
// general kernels: workgroup tile body
// =============================================================================================
constexpr int GT = 64;   // 64 x 64 outputs per workgroup, 4 x 4 per thread

template <typename T, int LH, int LW, int LP>
__device__ __forceinline__ void load_tile(float *s, const T *plane, int kind, int pitch, int H, int W, int py0, int px0,
                                          int boundary) {
    const int Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + LH <= Hp && px0 + LW <= Wp;
    int sy0 = py0, sx0 = px0;
    if (kind == SRC_VIRTUAL) {
        inside = inside && py0 >= PB_PAD && px0 >= PB_PAD && py0 + LH <= PB_PAD + H && px0 + LW <= PB_PAD + W;
        sy0 -= PB_PAD; sx0 -= PB_PAD;
    }
    const int tid = threadIdx.x;
    if (inside && ((pitch | sx0) & 3) == 0) {
        const T *base = plane + (long)sy0 * pitch + sx0;
        constexpr int C4 = LW / 4;
        for (int e = tid; e < LH * C4; e += NT) {
            const int r = e / C4, c = e - r * C4;
            *reinterpret_cast<float4 *>(s + r * LP + 4 * c) = ld4<T>(base + (long)r * pitch + 4 * c);
        }
    } else {
        for (int e = tid; e < LH * LW; e += NT) {
            const int r = e / LW, c = e - r * LW;
            const int iy = map_axis(py0 + r, H, kind, boundary), ix = map_axis(px0 + c, W, kind, boundary);
            s[r * LP + c] = (iy >= 0 && ix >= 0) ? pb_ld(plane + (long)iy * pitch + ix) : 0.f;
        }
    }
}

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void body_tile(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                          TOut *opl, int tile, int tiles_x, float *smem) {
    constexpr int LW = GT + 2 * R, LH = GT + 2 * R, NTAP = 2 * R + 1;
    constexpr int LP = LW + 4;                 // row pitch in LDS (floats), keeps 16-byte alignment
    constexpr int WCH = 1 + R / 2;             // float4 chunks per window row
    constexpr int NTP = 4 * WCH + 3;           // taps that can touch a window (3 zeros either side at R = 12)
    constexpr int PR = 4;
    const OutRegion rg = out_region(a);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    load_tile<TIn, LH, LW, LP>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    __syncthreads();
    const int tid = threadIdx.x;
    const int g = tid & 15, rgp = tid >> 4;    // 16 column groups x 16 row groups
    float4 acc[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    // gtaps[dy] = {0,0,0, k[dy][0..24], 0,0,0,0}: window element m feeds output i with tap t[m - i + 3]
    const PB_CONSTANT float *taps = as_constant(info->gtaps) + (PB_KRAD - R) * 32 + (PB_KRAD - R);
    const float *base = smem + (rgp * PR) * LP + 4 * g;
#pragma unroll 1
    for (int dy = 0; dy < NTAP; ++dy) {
        float t[NTP];
#pragma unroll
        for (int n = 0; n < NTP; ++n) t[n] = taps[dy * 32 + n];
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const float4 *src = reinterpret_cast<const float4 *>(base + (r + dy) * LP);
            float seg[4 * WCH];
#pragma unroll
            for (int q = 0; q < WCH; ++q) {
                const float4 v = src[q];
                seg[4 * q] = v.x; seg[4 * q + 1] = v.y; seg[4 * q + 2] = v.z; seg[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int m = 0; m < 4 * WCH; ++m) {
                acc[r].x = fmaf(t[m + 3], seg[m], acc[r].x);
                acc[r].y = fmaf(t[m + 2], seg[m], acc[r].y);
                acc[r].z = fmaf(t[m + 1], seg[m], acc[r].z);
                acc[r].w = fmaf(t[m], seg[m], acc[r].w);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) finish4<TX, TOut>(a, info, xpl, opl, rg, oy0 + rgp * PR + r, ox0 + 4 * g, acc[r]);
}

constexpr size_t kTileLds = sizeof(float) * (GT + 2 * PB_KRAD) * (GT + 2 * PB_KRAD + 4);

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT) void conv_pass_kernel(const ConvPass a, int blocks_per_plane, int tiles_x, int nsx,
                                                       int nsy, int seg_h) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x / blocks_per_plane;
    const int local = blockIdx.x - plane * blocks_per_plane;
    const pb_blur_info *info = a.info + plane / a.C;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const PB_CONSTANT pb_blur_info *cinfo = as_constant(info);
    if (cinfo->separable) {
        const int task = local * (NT / 64) + (threadIdx.x >> 6);
        if (task < nsx * nsy) body_stream<TIn, TX, TOut>(a, info, ipl, xpl, opl, task, nsx, seg_h, smem + (threadIdx.x >> 6) * SLINE);
    } else {
        const int R = a.force_full ? PB_KRAD : cinfo->radius;
        if (R <= 4) body_tile<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else if (R <= 8) body_tile<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else body_tile<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    }
}

template <typename TIn, typename TX, typename TOut>
int launch_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    // streaming decomposition: strips of 256 columns, cut vertically until the launch has enough
    // waves to fill 256 CUs x 12 waves (short segments re-read 24 halo rows each)
    const int nsx = (ow + SOUT - 1) / SOUT;
    const long want_waves = 256L * 14;
    long nsy = (want_waves + (long)p.P * nsx - 1) / ((long)p.P * nsx);
    const long nsy_max = (oh + 31) / 32;
    if (nsy > nsy_max) nsy = nsy_max;
    if (nsy < 1) nsy = 1;
    const int seg_h = (int)((oh + nsy - 1) / nsy);
    const int nsy_i = (oh + seg_h - 1) / seg_h;
    const long stream_blocks = ((long)nsx * nsy_i + NT / 64 - 1) / (NT / 64);
    const long tile_blocks = (long)tiles_x * tiles_y;
    const long bpp = stream_blocks > tile_blocks ? stream_blocks : tile_blocks;
    const long blocks = bpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    ProfScope prof(ctx, PB_PROF_CONV);
    hipLaunchKernelGGL((conv_pass_kernel<TIn, TX, TOut>), dim3((unsigned)blocks), dim3(NT), kTileLds, ctx->stream, p,
                       (int)bpp, tiles_x, nsx, nsy_i, seg_h);
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
