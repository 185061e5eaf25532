// Horner steps 2 and 3 of the polynomial in ONE launch, for rank-1 (separable) kernels:
//
//     t2 = K * t1 + a1 x          (deblurring.py:122-138, second reblurring)
//     y  = clamp(K * t2 + beta x) (third reblurring + crop + clamp, deblurring.py:234-239)
//
// The separable stencil pass is bound by the 128-byte lines a CU can request from L2 per unit time, not by HBM
// bytes or arithmetic (profiles/r02_inner_rank1_counters.txt): two separate passes move t2 out to memory and back
// in through that path, halo re-reads included.  Here a workgroup computes, for its 64x64 output tile, the
// (64+2R)^2 patch of t2 it needs from a (64+4R)^2 window of t1 and keeps it in LDS: t2 never leaves the CU.
//
//   window of t1 (DMA to LDS) -> x pass in place -> y pass into registers (+ a1 x) -> barrier -> t2 patch written
//   over the window -> x pass in place -> y pass (+ beta x, clamp) -> store.
//
// 1.7x the multiply-adds of the two separate passes, 0.64x their line requests, half their HBM bytes.  The same
// building blocks as the one-step rank-1 body (conv_tile_common.h): packed FMAs with SGPR tap pairs, waves that
// x-filter the rows they staged themselves, a register ring in the y pass.  Results are bit-identical to the
// two-launch path on interior blocks (same operations in the same order).
//
// Only valid outputs matter: an output at padded row py reads t2 rows py-R .. py+R, all inside the padded domain
// (the outputs are the interior crop), so t2 needs no boundary rule of its own; the boundary (wrap = 'fft',
// zero = 'direct') acts where it does in the two-launch path, on the window of t1.
#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

template <int R, int LP>
__device__ __forceinline__ void xpass_rows16(float *smem, int row0, int nrows, int lane, const f2 (&TP)[R + 1]) {
    // output groups 0..15 of rows [row0, row0 + nrows): a wave instruction covers 4 rows x 16 groups (the two rows
    // that share a 32-lane half are read conflict-free when the odd one starts XROT groups further along the row)
    constexpr int XROT = (16 - ((LP / 4) % 16)) % 16;
    const int rsub = lane >> 4, g = ((lane & 15) + XROT * (rsub & 1)) & 15;
    for (int it = 0; it * 4 < nrows; ++it) {
        const bool ok = it * 4 + rsub < nrows;
        float *row = smem + (ok ? row0 + it * 4 + rsub : row0) * LP;
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

// output groups 16 .. 16+E-1 of the same rows (the t2 patch is 64 + 2R = 4 (16 + E) samples wide, E = R/2): 64 / E rows
// per wave instruction.  Must run after xpass_rows16 of the same rows (that pass still reads input chunks 16 .. 21,
// which this one overwrites; it wrote chunks 0 .. 15, which this one does not read).
template <int R, int LP>
__device__ __forceinline__ void xpass_rows_extra(float *smem, int row0, int nrows, int lane, const f2 (&TP)[R + 1]) {
    constexpr int E = R / 2, RB = 64 / E;
    const int lr = lane / E, g = 16 + (lane - lr * E);
    for (int it = 0; it * RB < nrows; ++it) {
        const bool ok = lr < RB && it * RB + lr < nrows;
        float *row = smem + (ok ? row0 + it * RB + lr : row0) * LP;
        f2 d[R + 2];
#pragma unroll
        for (int q = 0; q < 1 + R / 2; ++q) {
            const float4 t4 = *reinterpret_cast<const float4 *>(row + 4 * (g + q));
            d[2 * q] = (f2){t4.x, t4.y};
            d[2 * q + 1] = (f2){t4.z, t4.w};
        }
        f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};
        XPassR<R, 0>::run(vxy, vzw, TP, d);
        wave_lds_fence();
        if (ok) *reinterpret_cast<float4 *>(row + 4 * g) = make_float4(vxy.x, vxy.y, vzw.x, vzw.y);
        wave_lds_fence();
    }
}

template <typename TX, typename TOut, int R>
__device__ __forceinline__ void body_fused(const ConvPass &a, float coef_mid, const pb_blur_info *info, const float *t1pl,
                                           const TX *xpl, TOut *opl, int tile, int tiles_x, float *smem) {
    constexpr int E = R / 2;
    constexpr int WA = GT + 4 * R, LPA = WA;          // window of t1
    constexpr int NB = 16 + E;                        // 4x4 blocks per side of the t2 patch
    constexpr int LB = GT + 2 * R, LPB = LB;          // the t2 patch
    constexpr int YROT = (16 - (LPB % 16)) % 16;
    const OutRegion rg = out_region(a);               // interior crop
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- stage the window of t1: every wave the rows it x-filters itself ----
    constexpr int RPWA = WA / 4;
    load_rows_wave<float, WA, WA, LPA, RPWA>(smem, t1pl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - 2 * R, ox0 - 2 * R, a.boundary, a.pad);
    const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
    f2 TP[R + 1], HY[(R + 2) / 2];
#pragma unroll
    for (int t = 0; t <= R; ++t) TP[t] = (f2){ckx[t], t ? ckx[t - 1] : 0.f};
#pragma unroll
    for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
    wave_lds_fence();
    xpass_rows16<R, LPA>(smem, wave * RPWA, RPWA, lane, TP);
    xpass_rows_extra<R, LPA>(smem, wave * RPWA, RPWA, lane, TP);
    // ---- t2 patch: two rounds of 4x4 blocks per thread (16x16 blocks, then the L-shaped rest), kept in registers ----
    int by[2], bx[2];
    bool have[2];
    by[0] = tid >> 4; bx[0] = tid & 15; have[0] = true;
    constexpr int NRIGHT = NB * E;                    // blocks in columns 16 .. NB-1 (all NB block rows)
    if (tid < NRIGHT) { by[1] = tid / E; bx[1] = 16 + (tid - by[1] * E); have[1] = true; }
    else { const int k = tid - NRIGHT; by[1] = 16 + (k >> 4); bx[1] = k & 15; have[1] = k < E * 16; }
    if (!have[1]) { by[1] = 0; bx[1] = 0; }
    float4 xq[2][4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xq[rd][r] = make_float4(0.f, 0.f, 0.f, 0.f);
            // (rows beyond the padded domain belong to outputs that are never stored: any finite value will do)
            if (have[rd]) xq[rd][r] = load_x4<TX>(a, xpl, min(max(oy0 - R + by[rd] * 4 + r, 0), a.H + 2 * a.pad - 1), ox0 - R + bx[rd] * 4);
        }
    __syncthreads();
    float4 t2[2][4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        f2 axy[4], azw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
        YPassR<R, 0>::run(axy, azw, HY, smem + (by[rd] * 4) * LPA + 4 * bx[rd], LPA);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t2[rd][r].x = fmaf(1.f, axy[r].x, coef_mid * xq[rd][r].x); t2[rd][r].y = fmaf(1.f, axy[r].y, coef_mid * xq[rd][r].y);
            t2[rd][r].z = fmaf(1.f, azw[r].x, coef_mid * xq[rd][r].z); t2[rd][r].w = fmaf(1.f, azw[r].y, coef_mid * xq[rd][r].w);
        }
    }
    __syncthreads();                                  // every thread has finished reading the filtered window
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
        if (have[rd])
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4 *>(smem + (by[rd] * 4 + r) * LPB + 4 * bx[rd]) = t2[rd][r];
    // ---- third step on the patch: exactly the one-step rank-1 body from here on ----
    Block4x4Epilogue<TX, TOut> epi;
    const int rgp = tid >> 4, gy = ((tid & 15) + YROT * (rgp & 1)) & 15;
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
    __syncthreads();
    constexpr int RPWB = (LB + 3) / 4;
    xpass_rows16<R, LPB>(smem, wave * RPWB, min(RPWB, LB - wave * RPWB), lane, TP);
    __syncthreads();
    f2 axy[4], azw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
    YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * LPB + 4 * gy, LPB);
    float4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, acc);
}

constexpr size_t kFusedLds = sizeof(float) * (GT + 4 * PB_KRAD) * (GT + 4 * PB_KRAD);   // 112 x 112 floats: 3 workgroups per CU

template <typename TX, typename TOut>
__global__ __launch_bounds__(NT, 3) void conv_fused_kernel(const ConvPass a, float coef_mid, int tiles_per_plane, int tiles_x,
                                                          int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int chunk = gridDim.x >> 3;                                 // XCD-aware order, as conv_tile_kernel
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int plane = __builtin_amdgcn_readfirstlane(tile_id / tiles_per_plane);
    const int local = tile_id - plane * tiles_per_plane;
    const pb_blur_info *info = a.info + __builtin_amdgcn_readfirstlane(plane / a.C);
    const PB_CONSTANT pb_blur_info *cinfo = as_constant(info);
    if (cinfo->separable == 0) return;                                // dense kernels take the two one-step launches
    const float *t1pl = static_cast<const float *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = cinfo->radius;
    if (R <= 4) body_fused<TX, TOut, 4>(a, coef_mid, info, t1pl, xpl, opl, local, tiles_x, smem);
    else if (R <= 8) body_fused<TX, TOut, 8>(a, coef_mid, info, t1pl, xpl, opl, local, tiles_x, smem);
    else body_fused<TX, TOut, 12>(a, coef_mid, info, t1pl, xpl, opl, local, tiles_x, smem);
}

template <typename TX, typename TOut>
int launch_fused(pb_ctx *ctx, const ConvPass &p, float coef_mid) {
    const int tiles_x = (p.W + GT - 1) / GT, tiles_y = (p.H + GT - 1) / GT;
    const long tpp = (long)tiles_x * tiles_y;
    const long blocks = tpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "fused pass: bad grid");
    static bool attr_set = false;                                     // > 48 KB of dynamic LDS needs the opt-in
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_fused_kernel<TX, TOut>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds));
        attr_set = true;
    }
    const long grid = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_fused_kernel<TX, TOut>), dim3((unsigned)grid), dim3(NT), kFusedLds, ctx->stream, p, coef_mid,
                       (int)tpp, tiles_x, (int)blocks);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// p describes the THIRD step (in = t1 padded fp32, x, out = interior crop, coef = beta, clamp); coef_mid = a1.
// Returns PB_ERR_UNSUPPORTED (without launching) for layouts the fused pass does not take.
int pb_launch_conv_fused(pb_ctx *ctx, const ConvPass &p, float coef_mid) {
    if (p.in_dtype != PB_F32 || p.in_kind != SRC_PADDED || p.out_kind != OUT_INTERIOR || p.epilogue != EPI_HORNER ||
        p.scale != 1.f)
        return PB_ERR_UNSUPPORTED;
    ProfScope prof(ctx, PB_PROF_CONV_FUSED);
    const int key = p.x_dtype * 3 + p.out_dtype;
    switch (key) {
        case 0: return launch_fused<float, float>(ctx, p, coef_mid);
        case 1: return launch_fused<float, __half>(ctx, p, coef_mid);
        case 3: return launch_fused<__half, float>(ctx, p, coef_mid);
        case 4: return launch_fused<__half, __half>(ctx, p, coef_mid);
        default: return PB_ERR_UNSUPPORTED;
    }
}
