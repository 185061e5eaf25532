// Stencil pass, rank-1 kernels: a workgroup walks DOWN a 64-column strip segment, one 64x64 output tile per step.
//
// The one-shot tile body (conv.hip) stages (64+2R) rows per tile and x-filters all of them, although 2R of them
// were already staged and x-filtered by the tile above.  Here the x-filtered bottom 2R rows of a step are copied
// to the top of the LDS tile and only 64 new rows are loaded and x-filtered per step: 27 % fewer loads, LDS writes
// and x-pass FMAs at R = 12, and the wave -> row assignment becomes regular (16 rows per wave per step).
// Geometry and arithmetic are otherwise conv.hip's in-LDS rank-1 body (in-place x pass, y pass into 4x4 register
// blocks, packed FMAs).  The kernel takes the "simple" tiles of rank-1 images (conv_tile_common.h:
// pb_tile_is_simple); conv_tile_kernel takes the rest of the same pass.
#include <cstdlib>

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

template <int R> struct SGeom {
    static constexpr int LW = GT + 2 * R, LH = GT + 2 * R, LP = LW, C4 = LW / 4;
    static constexpr int XROT = (16 - ((LP / 4) % 16)) % 16, YROT = (16 - (LP % 16)) % 16;
};

// wave-private staging of rows [row_lo + wave*RPW, +RPW) of the LDS tile from source rows starting at `base`
// (which points at the sample that belongs in LDS row row_lo, column 0)
template <typename TIn, int R, int NROWS>
__device__ __forceinline__ void stage_rows(float *smem, const TIn *base, int pitch, int row_lo) {
    using G = SGeom<R>;
    constexpr int RPW = (NROWS + 3) / 4;
    constexpr int NLD = (RPW * G::C4 + 63) / 64;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));            // recompute the lane's offsets every step instead of keeping them alive
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = wave * RPW;
    const int nrows = min(RPW, NROWS - r0);
    float4 buf[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = lane + k * 64;
        const int r = e / G::C4, c = e - r * G::C4;
        if (r < nrows) buf[k] = ld4<TIn>(base + (long)(r0 + r) * pitch + 4 * c);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = lane + k * 64;
        const int r = e / G::C4, c = e - r * G::C4;
        if (r < nrows) *reinterpret_cast<float4 *>(smem + (row_lo + r0 + r) * G::LP + 4 * c) = buf[k];
    }
}

// in-place x pass over LDS rows [row_lo + wave*RPW, +RPW): each wave filters the rows it staged itself
template <int R, int NROWS>
__device__ __forceinline__ void xpass_rows(float *smem, const f2 (&TP)[R + 1], int row_lo) {
    using G = SGeom<R>;
    constexpr int RPW = (NROWS + 3) / 4;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = tid >> 6, lane = tid & 63;
    const int rsub = lane >> 4, g = ((lane & 15) + G::XROT * (rsub & 1)) & 15;
    for (int it = 0; it < (RPW + 3) / 4; ++it) {
        const int rr = wave * RPW + it * 4 + rsub;
        const bool ok = (it * 4 + rsub) < RPW && rr < NROWS;
        float *row = smem + (row_lo + (ok ? rr : 0)) * G::LP;
        f2 d[R + 2];
#pragma unroll
        for (int p = 0; p < 1 + R / 2; ++p) {
            const float4 t4 = *reinterpret_cast<const float4 *>(row + 4 * (g + p));
            d[2 * p] = (f2){t4.x, t4.y};
            d[2 * p + 1] = (f2){t4.z, t4.w};
        }
        f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};
        XPassR<R, 0>::run(vxy, vzw, TP, d);
        wave_lds_fence();
        if (ok) *reinterpret_cast<float4 *>(row + 4 * g) = make_float4(vxy.x, vxy.y, vzw.x, vzw.y);
        wave_lds_fence();
    }
}

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void run_segment(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl, TOut *opl,
                                            int tx, int ty0, int ty1, float *smem) {
    using G = SGeom<R>;
    const OutRegion rg = out_region(a);
    const int in_off = a.in_kind == SRC_VIRTUAL ? PB_PAD : 0, x_off = a.x_kind == SRC_VIRTUAL ? PB_PAD : 0;
    const int out_off = a.out_kind == OUT_INTERIOR ? PB_PAD : 0;
    const int tid = threadIdx.x;
    const int rgp = tid >> 4, gy = ((tid & 15) + G::YROT * (rgp & 1)) & 15;
    const int ox0 = rg.x_lo + tx * GT;
    const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
    const float sc = a.scale, cf = a.coef;
    const bool cl = a.clamp01 != 0;
    // the retained rows: 2R x 16 float4, spread over the workgroup (copy slot j = tid + 256 * i)
    constexpr int NCOPY = (2 * R * 16 + NT - 1) / NT;
    for (int ty = ty0; ty < ty1; ++ty) {
        const int oy0 = rg.y_lo + ty * GT;
        const TX *xp = xpl + (long)(oy0 + rgp * 4 - x_off) * a.x_pitch + (ox0 + 4 * gy - x_off);
        TOut *op = opl + (long)(oy0 + rgp * 4 - out_off) * a.out_pitch + (ox0 + 4 * gy - out_off);
        float4 xr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = ld4<TX>(xp + (long)r * a.x_pitch);
        {
            f2 TP[R + 1];
#pragma unroll
            for (int p = 0; p <= R; ++p) TP[p] = (f2){ckx[p], p ? ckx[p - 1] : 0.f};
            if (ty == ty0) {
                // first step of the segment: the whole (64+2R)-row window
                const TIn *base = ipl + (long)(oy0 - R - in_off) * a.in_pitch + (ox0 - R - in_off);
                stage_rows<TIn, R, G::LH>(smem, base, a.in_pitch, 0);
                wave_lds_fence();
                xpass_rows<R, G::LH>(smem, TP, 0);
            } else {
                // later steps: rows 0 .. 2R-1 were copied from the previous step; 64 new rows below them
                const TIn *base = ipl + (long)(oy0 + R - in_off) * a.in_pitch + (ox0 - R - in_off);
                stage_rows<TIn, R, GT>(smem, base, a.in_pitch, 2 * R);
                wave_lds_fence();
                xpass_rows<R, GT>(smem, TP, 2 * R);
            }
        }
        __syncthreads();
        // ---- y pass: 4 x 4 outputs per thread; then pick up this thread's share of the rows to retain ----
        f2 HY[(R + 2) / 2];
#pragma unroll
        for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
        f2 axy[4], azw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
        YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * G::LP + 4 * gy, G::LP);
        float4 keep[NCOPY];
        const bool more = ty + 1 < ty1;
        if (more) {
#pragma unroll
            for (int i = 0; i < NCOPY; ++i) {
                const int j = tid + i * NT;
                if (j < 2 * R * 16) keep[i] = *reinterpret_cast<const float4 *>(smem + (GT + (j >> 4)) * G::LP + 4 * (j & 15));
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 v;
            v.x = fmaf(sc, axy[r].x, cf * xr[r].x); v.y = fmaf(sc, axy[r].y, cf * xr[r].y);
            v.z = fmaf(sc, azw[r].x, cf * xr[r].z); v.w = fmaf(sc, azw[r].y, cf * xr[r].w);
            if (cl) {
                v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
            }
            st4<TOut>(op + (long)r * a.out_pitch, v);
        }
        __syncthreads();                                    // everybody has finished reading the tile
        if (more) {
#pragma unroll
            for (int i = 0; i < NCOPY; ++i) {
                const int j = tid + i * NT;
                if (j < 2 * R * 16) *reinterpret_cast<float4 *>(smem + (j >> 4) * G::LP + 4 * (j & 15)) = keep[i];
            }
            // the copied rows are read only after the next step's barrier; the rows staged next are disjoint from them
        }
    }
}

constexpr size_t kSlideLds = sizeof(float) * SGeom<PB_KRAD>::LH * SGeom<PB_KRAD>::LP;      // 30 976 B

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT, 5) void conv_slide_kernel(const ConvPass a, int tiles_x, int tiles_y, int nseg, int seg_tiles,
                                                           int total_jobs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware order (speed only): every XCD gets one contiguous run of jobs; a job = (plane, segment, strip)
    const int chunk = gridDim.x >> 3;
    const int job = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (job >= total_jobs) return;
    const int per_plane = tiles_x * nseg;
    const int plane = job / per_plane;
    const int local = job - plane * per_plane;
    const int seg = local / tiles_x, tx = local - seg * tiles_x;       // neighbouring strips run side by side
    const pb_blur_info *info = a.info + plane / a.C;
    const PB_CONSTANT pb_blur_info *ci = as_constant(info);
    if (ci->separable == 0) return;
    const int R = ci->radius <= 4 ? 4 : (ci->radius <= 8 ? 8 : PB_KRAD);
    // the simple tiles of a strip are one contiguous run of tile rows
    int ty_lo = 0;
    while (ty_lo < tiles_y && !pb_tile_is_simple(a, R, ty_lo, tx)) ++ty_lo;
    int ty_hi = ty_lo;
    while (ty_hi < tiles_y && pb_tile_is_simple(a, R, ty_hi, tx)) ++ty_hi;
    const int ty0 = ty_lo + seg * seg_tiles, ty1 = min(ty_hi, ty0 + seg_tiles);
    if (ty0 >= ty1) return;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    if (R == 4) run_segment<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, tx, ty0, ty1, smem);
    else if (R == 8) run_segment<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, tx, ty0, ty1, smem);
    else run_segment<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, tx, ty0, ty1, smem);
}

template <typename TIn, typename TX, typename TOut>
int launch_slide_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    const long total_tiles = (long)tiles_x * tiles_y * p.P;
    // tiles per segment: long enough to amortise the 2R-row start-up, short enough to fill 256 CUs x 5 workgroups
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("PB_SLIDE_TILES"); forced = e ? atoi(e) : 0; }
    int seg_tiles = forced > 0 ? forced : (int)(total_tiles / 1280);
    if (seg_tiles < 1) seg_tiles = 1;
    if (seg_tiles > 8 && forced <= 0) seg_tiles = 8;
    const int nseg = (tiles_y + seg_tiles - 1) / seg_tiles;
    const long jobs = (long)tiles_x * nseg * p.P;
    if (jobs <= 0 || jobs > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    const long grid = (jobs + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_slide_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(NT), kSlideLds, ctx->stream, p, tiles_x,
                       tiles_y, nseg, seg_tiles, (int)jobs);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// the simple tiles of rank-1 images of float / half Horner passes; everything else is skipped on the device
int pb_launch_conv_slide(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    switch (key) {
        case 0: return launch_slide_typed<float, float, float>(ctx, p);
        case 1: return launch_slide_typed<float, float, __half>(ctx, p);
        case 3: return launch_slide_typed<float, __half, float>(ctx, p);
        case 4: return launch_slide_typed<float, __half, __half>(ctx, p);
        case 12: return launch_slide_typed<__half, __half, float>(ctx, p);
        case 13: return launch_slide_typed<__half, __half, __half>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "sliding stencil: unsupported dtype combination %d", key);
    }
}
