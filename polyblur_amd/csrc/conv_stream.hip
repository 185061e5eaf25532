// Stencil pass, rank-1 kernels: the wave-private "streaming" body (see conv.hip for the overview).
//
// One 64-lane workgroup (= one wave, no barriers) owns a 232-column strip segment of one plane and
// walks down it.  Per source row:  one coalesced 16-byte load per lane (1 KiB per wave, prefetched
// SU rows ahead)  ->  wave-private LDS line  ->  x pass (7 ds_read_b128 + 100 FMA per lane)  ->
// y pass scattered into 25+ rotating accumulator rows in registers  ->  one finished output row,
// Horner/taper epilogue, one 16-byte store per lane.  HBM sees each source word once per strip
// (+10 % column halo, + 24 rows per segment, both L2 hits), LDS moves ~8 words per sample.
//
// The kernel is specialised (MODE) on where the operands live for the three Horner steps so that
// the hot loop carries few scalars; MODE 3 reads the layout at run time (edgetaper passes).
#include <cstdlib>

#include "common.h"
#include "conv_common.h"

namespace {

template <typename TX>
__device__ __forceinline__ float4 stream_x4(const ConvPass &a, const TX *xpl, const OutRegion &rg, int py, int px) {
    if (py < rg.y_lo || py >= rg.y_hi || px >= rg.x_hi) return make_float4(0.f, 0.f, 0.f, 0.f);
    return load_x4<TX>(a, xpl, py, px);
}


constexpr int SR = PB_KRAD;                   // always evaluates all 25 taps per axis
constexpr int SNT = 2 * SR + 1;
constexpr int SOUT = 256 - 2 * SR;            // 232 outputs per strip: 58 lanes x 4 ...
constexpr int SLANES = SOUT / 4;              // ... so the 256-sample staged line is ONE float4 per lane
constexpr int SU = 4;                         // rows per unrolled group == source prefetch depth
constexpr int SACC = SNT + SU - 1;            // accumulator rows alive inside a group

// x pass of one staged row.  d[m] = (s[2m], s[2m+1]), m = 0..13: this lane's 28-sample window.
// Output pair (x,y) += (k[j], k[j-1]) * s[j], j = 0..25;  (z,w) += (k[j-2], k[j-3]) * s[j], j = 2..27.
// With symmetric taps k[p] = h[min(p, 24-p)], the pair T[p] = (k[p], k[p-1]) is TP[p] for p <= 12
// and the swapped TP[25-p] for p >= 13 (TP[p] = (h[p], h[p-1]), h[-1] = 0).
template <int P> struct XTap {
    static __device__ __forceinline__ void xy(f2 &acc, const f2 (&TP)[SR + 1], const f2 (&d)[14]) {
        constexpr int j = P;
        if (P <= SR) pk_bcast_data<0, j & 1>(acc, TP[P <= SR ? P : 0], d[j >> 1]);
        else pk_bcast_data<1, j & 1>(acc, TP[P > SR ? 2 * SR + 1 - P : 0], d[j >> 1]);
    }
    static __device__ __forceinline__ void zw(f2 &acc, const f2 (&TP)[SR + 1], const f2 (&d)[14]) {
        constexpr int j = P + 2;
        if (P <= SR) pk_bcast_data<0, j & 1>(acc, TP[P <= SR ? P : 0], d[j >> 1]);
        else pk_bcast_data<1, j & 1>(acc, TP[P > SR ? 2 * SR + 1 - P : 0], d[j >> 1]);
    }
};
template <int P> struct XPassUnroll {
    static __device__ __forceinline__ void run(f2 &vxy, f2 &vzw, const f2 (&TP)[SR + 1], const f2 (&d)[14]) {
        XTap<P>::xy(vxy, TP, d);
        XTap<P>::zw(vzw, TP, d);
        XPassUnroll<P + 1>::run(vxy, vzw, TP, d);
    }
};
template <> struct XPassUnroll<2 * SR + 2> {
    static __device__ __forceinline__ void run(f2 &, f2 &, const f2 (&)[SR + 1], const f2 (&)[14]) {}
};
// y scatter of one x-filtered row into slots S .. S+2R: slot S+2R-t += ky[t] * v,
// ky[t] = hy[min(t, 24-t)] taken from the pairs HY[m] = (hy[2m], hy[2m+1]).
template <int S, int T> struct YScatter {
    static __device__ __forceinline__ void run(f2 (&axy)[SNT + 3], f2 (&azw)[SNT + 3], const f2 (&HY)[7], f2 vxy, f2 vzw) {
        constexpr int q = T <= SR ? T : 2 * SR - T;
        constexpr int slot = S + 2 * SR - T;
        pk_bcast_tap<q & 1>(axy[slot], HY[q >> 1], vxy);
        pk_bcast_tap<q & 1>(azw[slot], HY[q >> 1], vzw);
        YScatter<S, T + 1>::run(axy, azw, HY, vxy, vzw);
    }
};
template <int S> struct YScatter<S, 2 * SR + 1> {
    static __device__ __forceinline__ void run(f2 (&)[SNT + 3], f2 (&)[SNT + 3], const f2 (&)[7], f2, f2) {}
};

template <int MODE> struct ModeKinds;
template <> struct ModeKinds<0> { static constexpr int in = SRC_VIRTUAL, x = SRC_VIRTUAL, out = OUT_PADDED; };
template <> struct ModeKinds<1> { static constexpr int in = SRC_PADDED, x = SRC_VIRTUAL, out = OUT_PADDED; };
template <> struct ModeKinds<2> { static constexpr int in = SRC_PADDED, x = SRC_VIRTUAL, out = OUT_INTERIOR; };

template <typename TIn, typename TX, typename TOut, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stream_kernel(const ConvPass a, int nsx, int nsy, int seg_h) {
    __shared__ __attribute__((aligned(16))) float4 line4[72];
    const int tasks = nsx * nsy;
    const int plane = blockIdx.x / tasks;
    const int task = blockIdx.x - plane * tasks;
    const pb_blur_info *info = a.info + plane / a.C;
    const PB_CONSTANT pb_blur_info *cinfo = as_constant(info);
    if (!cinfo->separable) return;                                 // this image takes the tile body
    const int in_kind = MODE < 3 ? ModeKinds<MODE < 3 ? MODE : 0>::in : a.in_kind;
    const int x_kind = MODE < 3 ? ModeKinds<MODE < 3 ? MODE : 0>::x : a.x_kind;
    const int out_kind = MODE < 3 ? ModeKinds<MODE < 3 ? MODE : 0>::out : a.out_kind;
    const int epilogue = MODE < 3 ? EPI_HORNER : a.epilogue;
    const int H = a.H, W = a.W, Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    const int y_lo = out_kind == OUT_INTERIOR ? PB_PAD : 0, y_hi = out_kind == OUT_INTERIOR ? PB_PAD + H : Hp;
    const int x_lo = out_kind == OUT_INTERIOR ? PB_PAD : 0, x_hi = out_kind == OUT_INTERIOR ? PB_PAD + W : Wp;
    const int lane = threadIdx.x;
    const int sy = task / nsx, sx = task - sy * nsx;
    const int x0 = x_lo + sx * SOUT, y0 = y_lo + sy * seg_h;
    const int rows_out = min(seg_h, y_hi - y0);
    if (rows_out <= 0) return;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    // taps -> SGPRs.  Rank-1 records carry symmetric marginals (kx[t] == kx[24-t], enforced when the
    // record is built), so 13 + 13 scalars are enough and stay resident for the whole walk.
    const PB_CONSTANT float *ckx = as_constant(info->kx), *cky = as_constant(info->ky);
    static_assert(SU == 4 && SACC == SNT + 3, "accumulator window");
    f2 TP[SR + 1], HY[7];
#pragma unroll
    for (int t = 0; t <= SR; ++t) TP[t] = (f2){ckx[t], t ? ckx[t - 1] : 0.f};
#pragma unroll
    for (int m = 0; m < 7; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= SR ? cky[2 * m + 1] : 0.f};

    const int n_in = rows_out + 2 * SR;
    const int px = x0 + 4 * lane;                                  // this lane's first output column (lanes < 58)
    const bool out_lane = lane < SLANES && px < x_hi;
    // ---- is the whole task free of wrap / zero / clamp handling and 16-byte aligned? (uniform) ----
    const int in_off = in_kind == SRC_VIRTUAL ? PB_PAD : 0;        // padded coordinate -> source index
    const int in_rows = in_kind == SRC_VIRTUAL ? H : Hp, in_cols = in_kind == SRC_VIRTUAL ? W : Wp;
    const int x_off = x_kind == SRC_VIRTUAL ? PB_PAD : 0;
    const int x_rows = x_kind == SRC_VIRTUAL ? H : Hp, x_cols = x_kind == SRC_VIRTUAL ? W : Wp;
    const int o_off = out_kind == OUT_INTERIOR ? PB_PAD : 0;
    const bool interior =
        x0 - SR - in_off >= 0 && x0 - SR - in_off + 256 <= in_cols && y0 - SR - in_off >= 0 &&
        y0 - SR - in_off + n_in <= in_rows && x0 - x_off >= 0 && x0 - x_off + SOUT <= x_cols && y0 - x_off >= 0 &&
        y0 - x_off + rows_out <= x_rows && x0 + SOUT <= x_hi && ((a.in_pitch | a.x_pitch | a.out_pitch) & 3) == 0 &&
        epilogue == EPI_HORNER;

    f2 axy[SACC], azw[SACC];
#pragma unroll
    for (int t = 0; t < SACC; ++t) { axy[t] = (f2){0.f, 0.f}; azw[t] = (f2){0.f, 0.f}; }

    // Software pipeline of one row step (both loops):
    //   PB_STAGE     stage source row i in the LDS line and start reading this lane's 28-sample window;
    //   PB_SCATTER   meanwhile scatter the PREVIOUS row's x-filtered value v into accumulator slots
    //                S .. S+2R (tap t -> slot S+2R-t): covers the LDS round trip with 100 packed FMAs;
    //   PB_XPASS     x pass of row i -> v, consumed by the next step.
    // Inside a group of SU steps output row o lives in slot o - (i0 - 1 - 2R); after the scatter at
    // step S slot S is complete (it is output row i - 1 - 2R).
#define PB_STAGE(CUR)                                                                                         \
    wave_lds_fence();                                                                                         \
    line4[lane] = (CUR);                                                                                      \
    wave_lds_fence();                                                                                         \
    f2 dwin[14];                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 1 + SR / 2; ++q) {                                                  \
        const float4 t4 = line4[lane + q];                                                                    \
        dwin[2 * q] = (f2){t4.x, t4.y}; dwin[2 * q + 1] = (f2){t4.z, t4.w};                                   \
    }
#define PB_SCATTER(S) YScatter<S, 0>::run(axy, azw, HY, vxy, vzw);
#define PB_XPASS()                                                                                            \
    vxy = (f2){0.f, 0.f}; vzw = (f2){0.f, 0.f};                                                               \
    XPassUnroll<0>::run(vxy, vzw, TP, dwin);

    f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};    // x-filtered previous row (zero before the first)
    if (interior) {
        // ---------------- fast loop: plain pointer walks, 16-byte accesses only ----------------------
        const TIn *ip = ipl + (long)(y0 - SR - in_off) * a.in_pitch + (x0 - SR - in_off) + 4 * lane;
        const TX *xp = xpl + (long)(y0 - x_off) * a.x_pitch + (px - x_off);
        TOut *op = opl + (long)(y0 - o_off) * a.out_pitch + (px - o_off);
        const float scale = a.scale, coef = a.coef;
        const bool clamp01 = a.clamp01 != 0;
        // SU source rows and SU x rows in flight (loads retire in order: keep both streams equally deep)
        float4 pf[SU], xq[SU];
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            pf[k] = ld4<TIn>(ip + (long)k * a.in_pitch);                 // n_in >= 25 > SU
            xq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ip += (long)SU * a.in_pitch;
        // x row of output o is consumed at step i = o + 2R + 1; steps 2R+1-SU .. start its prefetch
#define PB_FAST_STEP(S)                                                                                       \
    {                                                                                                         \
        const int i = i0 + (S);                                                                               \
        if (i <= n_in) { /* uniform; step n_in only flushes */                                                \
            const float4 cur = pf[S];                                                                         \
            const float4 xcur = xq[S];                                                                        \
            if (i + SU < n_in) pf[S] = ld4<TIn>(ip);                                                          \
            ip += a.in_pitch;                                                                                 \
            const int o_next = i + SU - 2 * SR - 1; /* output row whose x operand is fetched now */          \
            if (o_next >= 0 && o_next < rows_out && out_lane) xq[S] = ld4<TX>(xp + (long)o_next * a.x_pitch); \
            PB_STAGE(cur)                                                                                     \
            PB_SCATTER(S)                                                                                     \
            const int o = i - 1 - 2 * SR;                                                                     \
            if (o >= 0 && out_lane) {                                                                         \
                float4 r;                                                                                     \
                r.x = scale * axy[S].x + coef * xcur.x; r.y = scale * axy[S].y + coef * xcur.y;               \
                r.z = scale * azw[S].x + coef * xcur.z; r.w = scale * azw[S].y + coef * xcur.w;               \
                if (clamp01) {                                                                                \
                    r.x = fminf(fmaxf(r.x, 0.f), 1.f); r.y = fminf(fmaxf(r.y, 0.f), 1.f);                     \
                    r.z = fminf(fmaxf(r.z, 0.f), 1.f); r.w = fminf(fmaxf(r.w, 0.f), 1.f);                     \
                }                                                                                             \
                st4<TOut>(op + (long)o * a.out_pitch, r);                                                     \
            }                                                                                                 \
            PB_XPASS()                                                                                        \
        }                                                                                                     \
    }
        for (int i0 = 0; i0 <= n_in; i0 += SU) {
            PB_FAST_STEP(0) PB_FAST_STEP(1) PB_FAST_STEP(2) PB_FAST_STEP(3)
#pragma unroll
            for (int j = 0; j < SACC; ++j) {
                axy[j] = (j + SU < SACC) ? axy[j + SU] : (f2){0.f, 0.f};
                azw[j] = (j + SU < SACC) ? azw[j + SU] : (f2){0.f, 0.f};
            }
        }
    } else {
        // ---------------- boundary loop: every index mapped (wrap / zero / replicate clamp) --------
        ConvPass b = a;
        b.in_kind = in_kind; b.x_kind = x_kind; b.out_kind = out_kind; b.epilogue = epilogue;
        const OutRegion rg{y_lo, y_hi, x_lo, x_hi};
        const int pxo = out_lane ? px : x_hi;
#define PB_SLOW_STEP(S)                                                                                       \
    {                                                                                                         \
        const int i = i0 + (S);                                                                               \
        if (i <= n_in) {                                                                                      \
            const int iy = i < n_in ? map_axis(y0 - SR + i, H, in_kind, a.boundary) : -1;                     \
            float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
            if (iy >= 0) {                                                                                    \
                const TIn *row = ipl + (long)iy * a.in_pitch;                                                 \
                const int c0 = x0 - SR + 4 * lane;                                                            \
                const int j0 = map_axis(c0, W, in_kind, a.boundary), j1 = map_axis(c0 + 1, W, in_kind, a.boundary);     \
                const int j2 = map_axis(c0 + 2, W, in_kind, a.boundary), j3 = map_axis(c0 + 3, W, in_kind, a.boundary); \
                if (j0 >= 0) cur.x = pb_ld(row + j0);                                                         \
                if (j1 >= 0) cur.y = pb_ld(row + j1);                                                         \
                if (j2 >= 0) cur.z = pb_ld(row + j2);                                                         \
                if (j3 >= 0) cur.w = pb_ld(row + j3);                                                         \
            }                                                                                                 \
            PB_STAGE(cur)                                                                                     \
            PB_SCATTER(S)                                                                                     \
            if (i - 1 >= 2 * SR)                                                                              \
                finish4<TOut>(b, info, opl, rg, y0 + i - 1 - 2 * SR, pxo,                                     \
                              make_float4(axy[S].x, axy[S].y, azw[S].x, azw[S].y),                            \
                              stream_x4<TX>(b, xpl, rg, y0 + i - 1 - 2 * SR, pxo));                           \
            PB_XPASS()                                                                                        \
        }                                                                                                     \
    }
        for (int i0 = 0; i0 <= n_in; i0 += SU) {
            PB_SLOW_STEP(0) PB_SLOW_STEP(1) PB_SLOW_STEP(2) PB_SLOW_STEP(3)
#pragma unroll
            for (int j = 0; j < SACC; ++j) {
                axy[j] = (j + SU < SACC) ? axy[j + SU] : (f2){0.f, 0.f};
                azw[j] = (j + SU < SACC) ? azw[j + SU] : (f2){0.f, 0.f};
            }
        }
    }
#undef PB_FAST_STEP
#undef PB_SLOW_STEP
#undef PB_STAGE
#undef PB_SCATTER
#undef PB_XPASS
}

template <typename TIn, typename TX, typename TOut>
int launch_stream_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    // Strips of 232 columns, cut vertically into segments.  The kernel holds 2 waves per SIMD
    // (218 VGPRs), i.e. 2048 resident waves on 256 CUs: aim for ONE full round of waves -- a second,
    // partly filled round costs a whole segment time -- and never cut below 32 rows (every segment
    // re-filters 24 halo rows).
    static long slots = 0;
    if (!slots) {
        const char *e = getenv("PB_STREAM_WAVES");
        slots = e ? atol(e) : 2048;
        if (slots < 1) slots = 2048;
    }
    const int nsx = (ow + SOUT - 1) / SOUT;
    const long cols = (long)p.P * nsx;
    long nsy = slots / cols;                       // floor: stay within one round
    const long nsy_max = (oh + 31) / 32;
    if (nsy > nsy_max) nsy = nsy_max;
    if (nsy < 1) nsy = 1;
    const int seg_h = (int)((oh + nsy - 1) / nsy);
    const int nsy_i = (oh + seg_h - 1) / seg_h;
    const long blocks = (long)nsx * nsy_i * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "stream pass: bad grid");
    int mode = 3;
    if (p.epilogue == EPI_HORNER && p.x_kind == SRC_VIRTUAL) {
        if (p.in_kind == SRC_VIRTUAL && p.out_kind == OUT_PADDED) mode = 0;
        else if (p.in_kind == SRC_PADDED && p.out_kind == OUT_PADDED) mode = 1;
        else if (p.in_kind == SRC_PADDED && p.out_kind == OUT_INTERIOR) mode = 2;
    }
#define PB_STREAM(M) hipLaunchKernelGGL((conv_stream_kernel<TIn, TX, TOut, M>), dim3((unsigned)blocks), dim3(64), 0, \
                                        ctx->stream, p, nsx, nsy_i, seg_h)
    switch (mode) {
        case 0: PB_STREAM(0); break;
        case 1: PB_STREAM(1); break;
        case 2: PB_STREAM(2); break;
        default: PB_STREAM(3); break;
    }
#undef PB_STREAM
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

int pb_launch_conv_stream(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 4 + p.x_dtype * 2 + p.out_dtype;
    switch (key) {
        case 0: return launch_stream_typed<float, float, float>(ctx, p);
        case 1: return launch_stream_typed<float, float, __half>(ctx, p);
        case 2: return launch_stream_typed<float, __half, float>(ctx, p);
        case 3: return launch_stream_typed<float, __half, __half>(ctx, p);
        case 6: return launch_stream_typed<__half, __half, float>(ctx, p);
        case 7: return launch_stream_typed<__half, __half, __half>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "stream pass: unsupported dtype combination %d", key);
    }
}
