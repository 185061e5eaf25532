// Stencil pass, rank-1 kernels: a workgroup walks DOWN a 64-column strip segment, one 64x64 output tile per step.
//
// Every byte a workgroup requests goes through the L2 whether it hits or not, and the CU <-> L2 fabric moves about
// 6.5 TB/s on this chip (tools/ubench4.hip: 406 MB per 4K launch with 64x64 tiles and their halos take 62 us, 300 MB
// without halos take 50 us).  The one-shot tile body (conv.hip) requests (64+2R)^2 input samples per 64^2 outputs and
// x-filters all (64+2R) rows, although the top 2R of them were requested and x-filtered by the tile above.  Here the
// x-filtered bottom 2R rows of a step are copied to the top of the LDS tile (LDS -> LDS, by the wave that is about to
// overwrite them) and only 64 new rows are requested and x-filtered per step: at R = 12, 13 % fewer bytes through L2,
// 27 % fewer x-pass FMAs and LDS writes.  Borders need nothing special: the rows a step retains are the same PADDED
// rows the next step would have loaded, and new rows go through the same mapped loader as in conv.hip.
// Geometry and arithmetic are otherwise conv.hip's in-LDS rank-1 body.  Takes the rank-1 images of a pass;
// conv_tile_kernel (sep_in_tile = 0) takes the others.
#include <cstdlib>

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

template <int R> struct SGeom {
    static constexpr int LW = GT + 2 * R, LH = GT + 2 * R, LP = LW;
    static constexpr int XROT = (16 - ((LP / 4) % 16)) % 16, YROT = (16 - (LP % 16)) % 16;
};

// in-place x pass over LDS rows [row_lo + wave*RPW, +RPW), RPW = ceil(NROWS/4): each wave filters the rows it staged
template <int R, int NROWS>
__device__ __forceinline__ void xpass_rows(float *smem, const f2 (&TP)[R + 1], int row_lo, int tid) {
    using G = SGeom<R>;
    constexpr int RPW = (NROWS + 3) / 4;
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
    const int ox0 = rg.x_lo + tx * GT;
    const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
    for (int ty = ty0; ty < ty1; ++ty) {
        const int oy0 = rg.y_lo + ty * GT;
        if (oy0 >= rg.y_hi) break;
        // lane-derived offsets are recomputed every step (kept opaque) instead of living in registers across the loop
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6, lane = tid & 63;
        const int rgp = tid >> 4, gy = ((tid & 15) + G::YROT * (rgp & 1)) & 15;
        Block4x4Epilogue<TX, TOut> epi;
        epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
        {
            f2 TP[R + 1];
#pragma unroll
            for (int p = 0; p <= R; ++p) TP[p] = (f2){ckx[p], p ? ckx[p - 1] : 0.f};
            if (ty == ty0) {
                // first step of the segment: the whole (64+2R)-row window
                load_rows_wave<TIn, G::LH, G::LW, G::LP, (G::LH + 3) / 4>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R,
                                                                          ox0 - R, a.boundary, tid);
                wave_lds_fence();
                xpass_rows<R, G::LH>(smem, TP, 0, tid);
            } else {
                // later steps: x-filtered rows 64 .. 64+2R-1 of the previous step become rows 0 .. 2R-1; every wave moves
                // those of them that lie in the range it is about to overwrite (this wave stages rows 2R+16w .. +15)
                const int lo = max(GT, 2 * R + 16 * wave), hi = min(GT + 2 * R, 2 * R + 16 * wave + 16);
                for (int j = lane; j < (hi - lo) * 16; j += 64) {
                    const int r = lo + (j >> 4), c = j & 15;
                    const float4 v = *reinterpret_cast<const float4 *>(smem + r * G::LP + 4 * c);
                    *reinterpret_cast<float4 *>(smem + (r - GT) * G::LP + 4 * c) = v;
                }
                wave_lds_fence();
                load_rows_wave<TIn, GT, G::LW, G::LP, GT / 4>(smem + 2 * R * G::LP, ipl, a.in_kind, a.in_pitch, a.H, a.W,
                                                               oy0 + R, ox0 - R, a.boundary, tid);
                wave_lds_fence();
                xpass_rows<R, GT>(smem, TP, 2 * R, tid);
            }
        }
        __syncthreads();
        // ---- y pass: 4 x 4 outputs per thread ----
        f2 HY[(R + 2) / 2];
#pragma unroll
        for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
        f2 axy[4], azw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
        YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * G::LP + 4 * gy, G::LP);
        float4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
        epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, acc);
        __syncthreads();                                    // everybody has finished reading the tile
    }
}

constexpr size_t kSlideLds = sizeof(float) * SGeom<PB_KRAD>::LH * SGeom<PB_KRAD>::LP;      // 30 976 B

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT, 5) void conv_slide_kernel(const ConvPass a, int tiles_x, int tiles_y, int nseg, int seg_tiles,
                                                           int total_jobs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware order (speed only): every XCD gets one contiguous run of jobs; a job = (plane, segment, strip) and
    // neighbouring strips of one segment run side by side (they share halo columns in that XCD's L2)
    const int chunk = gridDim.x >> 3;
    const int job = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (job >= total_jobs) return;
    const int per_plane = tiles_x * nseg;
    const int plane = __builtin_amdgcn_readfirstlane(job / per_plane);
    const int local = job - plane * per_plane;
    const int seg = __builtin_amdgcn_readfirstlane(local / tiles_x), tx = local - seg * tiles_x;
    const pb_blur_info *info = a.info + __builtin_amdgcn_readfirstlane(plane / a.C);
    const PB_CONSTANT pb_blur_info *ci = as_constant(info);
    if (ci->separable == 0) return;                                  // dense taps: conv_tile_kernel does this image
    const int ty0 = seg * seg_tiles, ty1 = min(tiles_y, ty0 + seg_tiles);
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = ci->radius;
    if (R <= 4) run_segment<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, tx, ty0, ty1, smem);
    else if (R <= 8) run_segment<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, tx, ty0, ty1, smem);
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
    if (seg_tiles > tiles_y) seg_tiles = tiles_y;
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

// the rank-1 images of a pass (any epilogue, any dtype combination the tile kernel knows); others are skipped on the device
int pb_launch_conv_slide(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    typedef unsigned char u8;
    switch (key) {
        case 0: return launch_slide_typed<float, float, float>(ctx, p);
        case 1: return launch_slide_typed<float, float, __half>(ctx, p);
        case 3: return launch_slide_typed<float, __half, float>(ctx, p);
        case 4: return launch_slide_typed<float, __half, __half>(ctx, p);
        case 12: return launch_slide_typed<__half, __half, float>(ctx, p);
        case 13: return launch_slide_typed<__half, __half, __half>(ctx, p);
        case 24: return launch_slide_typed<u8, u8, float>(ctx, p);
        case 6: return launch_slide_typed<float, u8, float>(ctx, p);
        case 8: return launch_slide_typed<float, u8, u8>(ctx, p);
        case 2: return launch_slide_typed<float, float, u8>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: unsupported dtype combination %d", key);
    }
}
