// Stencil pass, rank-1 kernels: 128 x 64 output tiles, 512-thread workgroups.
//
// The memory side of a rank-1 Horner step -- stage a (64+2R)-row window in LDS, read the x operand, store the
// centre -- was measured on its own (tools/ubench4.hip, 4K, arithmetic removed): 61-64 us per launch with 64x64
// tiles and 256 threads, 52-55 us with 128x64 tiles and 512 threads (longer contiguous row segments, a quarter less
// halo per output, twice the loads in flight per workgroup), against 52 us with no halo at all.  This kernel is
// conv.hip's in-LDS rank-1 body in that geometry: eight waves stage and x-filter 11 rows each in place (a wave
// instruction covers 2 rows x 32 column groups, so no lane rotation is needed to stay bank-conflict-free), one
// barrier, then every thread accumulates a 4x4 output block along y.  LDS: 88 x 152 floats = 53 504 B, three
// workgroups (24 waves) per CU at <= 80 VGPRs.
//
// It takes the rank-1 images of a pass; conv_tile_kernel (launched with sep_in_tile = 0) takes the others.
#include <cstdlib>

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

constexpr int WNT = 512;            // threads per workgroup
constexpr int WTW = 128, WTH = 64;  // outputs per tile

// wave-private staging, 8 waves: as load_rows_wave (conv_tile_common.h) but every wave owns RPW = ceil(LH/8) rows
template <typename T, int LH, int LW, int LP, int RPW>
__device__ __forceinline__ void stage_rows8(float *s, const T *plane, int kind, int pitch, int H, int W, int py0, int px0,
                                            int boundary) {
    const int Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + LH <= Hp && px0 + LW <= Wp;
    int sy0 = py0, sx0 = px0;
    if (kind == SRC_VIRTUAL) {
        inside = inside && py0 >= PB_PAD && px0 >= PB_PAD && py0 + LH <= PB_PAD + H && px0 + LW <= PB_PAD + W;
        sy0 -= PB_PAD; sx0 -= PB_PAD;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = wave * RPW;
    const int nrows = min(RPW, LH - r0);
    constexpr int C4 = LW / 4;
    constexpr int NLD = (RPW * C4 + 63) / 64;
    const bool aligned = ((pitch | sx0) & 3) == 0;
    float4 buf[NLD];
    if (inside && aligned) {
        const T *base = plane + (long)(sy0 + r0) * pitch + sx0;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows) buf[k] = ld4<T>(base + (long)r * pitch + 4 * c);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows)
                buf[k] = load_chunk_mapped<T>(plane, pitch, map_axis(py0 + r0 + r, H, kind, boundary), px0 + 4 * c, W, kind,
                                              boundary, aligned);
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = lane + k * 64;
        const int r = e / C4, c = e - r * C4;
        if (r < nrows) *reinterpret_cast<float4 *>(s + (r0 + r) * LP + 4 * c) = buf[k];
    }
}

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void body_wide(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl, TOut *opl,
                                          int ty, int tx, float *smem) {
    constexpr int LW = WTW + 2 * R, LH = WTH + 2 * R, LP = LW;
    constexpr int RPW = (LH + 7) / 8;                      // rows staged and x-filtered by each of the 8 waves
    const OutRegion rg = out_region(a);
    const int oy0 = rg.y_lo + ty * WTH, ox0 = rg.x_lo + tx * WTW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rgp = tid >> 5, gy = tid & 31;               // y pass / output mapping: 16 row groups x 32 column groups
    Block4x4Epilogue<TX, TOut> epi;
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
    stage_rows8<TIn, LH, LW, LP, RPW>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
    {
        f2 TP[R + 1];
#pragma unroll
        for (int t = 0; t <= R; ++t) TP[t] = (f2){ckx[t], t ? ckx[t - 1] : 0.f};
        wave_lds_fence();                                   // a wave reads back only rows it staged itself
        // ---- x pass, in place: a wave instruction covers 2 rows x 32 column groups ----
        const int rsub = lane >> 5, g = lane & 31;
        for (int it = 0; it < (RPW + 1) / 2; ++it) {
            const int rr = wave * RPW + it * 2 + rsub;
            const bool ok = (it * 2 + rsub) < RPW && rr < LH;
            float *row = smem + (ok ? rr : 0) * LP;
            f2 d[R + 2];
#pragma unroll
            for (int q = 0; q < 1 + R / 2; ++q) {
                const float4 t4 = *reinterpret_cast<const float4 *>(row + 4 * (g + q));
                d[2 * q] = (f2){t4.x, t4.y};
                d[2 * q + 1] = (f2){t4.z, t4.w};
            }
            f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};
            XPassR<R, 0>::run(vxy, vzw, TP, d);
            wave_lds_fence();          // every lane of the wave has read its window before the row is overwritten
            if (ok) *reinterpret_cast<float4 *>(row + 4 * g) = make_float4(vxy.x, vxy.y, vzw.x, vzw.y);
            wave_lds_fence();
        }
    }
    __syncthreads();
    // ---- y pass: 4 x 4 outputs per thread from the x-filtered tile ----
    f2 HY[(R + 2) / 2];
#pragma unroll
    for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
    f2 axy[4], azw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
    YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * LP + 4 * gy, LP);
    float4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, acc);
}

constexpr size_t kWideLds = sizeof(float) * (WTW + 2 * PB_KRAD) * (WTH + 2 * PB_KRAD);   // 152 x 88 floats

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(WNT, 6) void conv_wide_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware order (speed only): every XCD gets one contiguous run of tiles (the grid is padded to a multiple of 8)
    const int chunk = gridDim.x >> 3;
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int plane = __builtin_amdgcn_readfirstlane(tile_id / tiles_per_plane);
    const int local = tile_id - plane * tiles_per_plane;
    const pb_blur_info *info = a.info + __builtin_amdgcn_readfirstlane(plane / a.C);
    const PB_CONSTANT pb_blur_info *cinfo = as_constant(info);
    if (cinfo->separable == 0) return;                              // dense taps: conv_tile_kernel does this image
    const int ty = __builtin_amdgcn_readfirstlane(local / tiles_x), tx = local - ty * tiles_x;
    const OutRegion rg = out_region(a);
    if (rg.y_lo + ty * WTH >= rg.y_hi) return;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = cinfo->radius;
    if (R <= 4) body_wide<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, ty, tx, smem);
    else if (R <= 8) body_wide<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, ty, tx, smem);
    else body_wide<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, ty, tx, smem);
}

template <typename TIn, typename TX, typename TOut>
int launch_wide_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + WTW - 1) / WTW, tiles_y = (oh + WTH - 1) / WTH;
    const long tpp = (long)tiles_x * tiles_y;
    const long blocks = tpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    const long grid = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_wide_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(WNT), kWideLds, ctx->stream, p, (int)tpp,
                       tiles_x, (int)blocks);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// rank-1 images of a pass (any epilogue, any dtype combination the tile kernel knows); others are skipped on the device
int pb_launch_conv_wide(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    typedef unsigned char u8;
    switch (key) {
        case 0: return launch_wide_typed<float, float, float>(ctx, p);
        case 1: return launch_wide_typed<float, float, __half>(ctx, p);
        case 3: return launch_wide_typed<float, __half, float>(ctx, p);
        case 4: return launch_wide_typed<float, __half, __half>(ctx, p);
        case 12: return launch_wide_typed<__half, __half, float>(ctx, p);
        case 13: return launch_wide_typed<__half, __half, __half>(ctx, p);
        case 24: return launch_wide_typed<u8, u8, float>(ctx, p);
        case 6: return launch_wide_typed<float, u8, float>(ctx, p);
        case 8: return launch_wide_typed<float, u8, u8>(ctx, p);
        case 2: return launch_wide_typed<float, float, u8>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: unsupported dtype combination %d", key);
    }
}
