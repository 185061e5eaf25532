// The reblurring stencil pass: out = epilogue(K * in, x) on the replicate-padded domain.
//
// One launch evaluates one Horner step  t <- K*t + coef*x  of the polynomial deconvolution
// (reference deblurring.py:122-138 / :141-169) or one edgetaper blend (edgetaper.py:30-32)
// for every plane of the batch.  Replaces filters.convolve2d / conv2d_ (filters.py:14-49),
// utils.pad_with_kernel / crop_with_kernel (utils.py:48-61, folded into index arithmetic)
// and the intent of separable_gaussian2d.cpp:47-88 (x-then-y 1-D Gaussian passes).
//
// Two bodies live in the one kernel; every image picks its own at run time from its pb_blur_info record (so a
// batch may mix them and the host never synchronises).  A 256-thread workgroup owns a 64x64 output tile of one
// plane and stages the (64+2R)^2 samples it needs in LDS once (R = 4, 8 or 12: the image's support class):
//
// * rank-1 kernels (theta % 90 == 0 or sigma == rho): every wave stages the rows it x-filters itself, filters them
//   in place along x (ds_read_b128 x7 -> 52 packed FMAs -> ds_write_b128 per 4 outputs), one barrier, then the
//   y pass accumulates a 4x4 output block per thread (28 ds_read_b128, 200 packed FMAs).  Per sample the pass moves
//   its 3 algorithmic words of HBM traffic (measured 1.05x) -- both 1-D passes in one launch.
// * general (rotated anisotropic) kernels: the exact 2-D stencil from the same LDS tile, 4x4 outputs per thread, the
//   25-tap rows of the kernel streamed through SGPRs (208 packed FMAs per kernel row).  fp32 VALU-bound by design.
//
// No MFMA anywhere: the pass is bound by HBM / latency (rank-1) or by the fp32 vector rate (general).

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {


// =============================================================================================
// general kernels: workgroup tile body
// =============================================================================================

// One window row of the general stencil: element m of the thread's window (column c0 - R + m) feeds
// the four outputs c0 .. c0+3 with taps t[m+3], t[m+2], t[m+1], t[m] (t = the kernel row, zero padded by
// 3).  Two packed FMAs per element: (x,y) += (t[m+3], t[m+2]) * s,  (z,w) += (t[m+1], t[m]) * s, with
// the tap pair in an aligned SGPR pair -- TA[i] = (t[2i], t[2i+1]) or TB[i] = (t[2i+1], t[2i+2]) -- read
// swapped through op_sel, and the sample broadcast from its register pair.
template <int R, int M, int NP> struct GenRow {
    static __device__ __forceinline__ void run(f2 &axy, f2 &azw, const f2 (&TA)[NP], const f2 (&TB)[NP], const f2 (&d)[R + 2]) {
        constexpr bool full = R == PB_KRAD;              // at R = 12 the padding taps are exact zeros: skip them
        if constexpr (!(full && M >= 2 * R + 2)) {       // (t[m+3], t[m+2])
            constexpr int n = M + 2;
            if constexpr ((n & 1) == 0) pk_bcast_data<1, M & 1>(axy, TA[n >> 1], d[M >> 1]);
            else pk_bcast_data<1, M & 1>(axy, TB[(n - 1) >> 1], d[M >> 1]);
        }
        if constexpr (!(full && M < 2)) {                // (t[m+1], t[m])
            if constexpr ((M & 1) == 0) pk_bcast_data<1, M & 1>(azw, TA[M >> 1], d[M >> 1]);
            else pk_bcast_data<1, M & 1>(azw, TB[(M - 1) >> 1], d[M >> 1]);
        }
        if constexpr (M + 1 < 2 * R + 4) GenRow<R, M + 1, NP>::run(axy, azw, TA, TB, d);
    }
};

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void body_tile(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                          TOut *opl, int tile, int tiles_x, float *smem) {
    constexpr int LW = GT + 2 * R, LH = GT + 2 * R, NTAP = 2 * R + 1;
    constexpr int LP = LW;                     // unpadded LDS rows (5 workgroups per CU); conflicts avoided by YROT
    constexpr int YROT = (16 - (LP % 16)) % 16;  // odd row groups start YROT column groups further along the row
    constexpr int WCH = 1 + R / 2;             // float4 chunks per window row
    constexpr int NP = 2 * WCH + 2;            // tap pairs per copy (taps n = 0 .. 4 WCH + 3)
    constexpr int PR = 4;
    const OutRegion rg = out_region(a);
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    const int tid = threadIdx.x;
    // 16 column groups x 16 row groups.  The two row groups that share a 32-lane half are 4 LDS rows
    // (4 * LP floats) apart; rotating the odd one's column group keeps every ds_read_b128 conflict-free.
    const int rgp = tid >> 4, g = ((tid & 15) + YROT * (rgp & 1)) & 15;
    load_tile<TIn, LH, LW, LP>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    __syncthreads();
    f2 axy[PR], azw[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
    // gtaps[y] = {0,0,0, k[y][0..24], 0,0,0,0}, gtaps_odd[y][n] = gtaps[y][n+1]; class R reads from column 12-R
    const PB_CONSTANT float *ta = as_constant(info->gtaps) + (PB_KRAD - R) * 32 + (PB_KRAD - R);
    const PB_CONSTANT float *tb = as_constant(info->gtaps_odd) + (PB_KRAD - R) * 32 + (PB_KRAD - R);
    const float *base = smem + (rgp * PR) * LP + 4 * g;
#pragma unroll 1
    for (int dy = 0; dy < NTAP; ++dy) {
        f2 TA[NP], TB[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            TA[i] = (f2){ta[dy * 32 + 2 * i], ta[dy * 32 + 2 * i + 1]};
            TB[i] = (f2){tb[dy * 32 + 2 * i], tb[dy * 32 + 2 * i + 1]};
        }
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const float4 *src = reinterpret_cast<const float4 *>(base + (r + dy) * LP);
            f2 d[R + 2];
#pragma unroll
            for (int q = 0; q < WCH; ++q) {
                const float4 v = src[q];
                d[2 * q] = (f2){v.x, v.y};
                d[2 * q + 1] = (f2){v.z, v.w};
            }
            GenRow<R, 0, NP>::run(axy[r], azw[r], TA, TB, d);
        }
    }
    float4 acc[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
    // VALU-bound body: fetch the x operand only now (keeps 16 registers free during the stencil)
    Block4x4Epilogue<TX, TOut> epi;
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * PR, ox0 + 4 * g);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * PR, ox0 + 4 * g, acc);
}

// ---------------------------------------------------------------------------------------------
// rank-1 kernels in the tile geometry: x pass in place in LDS, y pass into registers.
// Both passes run on packed FMAs (conv_common.h); rank-1 records carry symmetric marginals.
// ---------------------------------------------------------------------------------------------
template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void body_tile_sep(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                              TOut *opl, int tile, int tiles_x, float *smem) {
    constexpr int LW = GT + 2 * R, LH = GT + 2 * R;
    constexpr int LP = LW;                        // unpadded LDS rows: 30 976 B at R = 12 -> 5 workgroups per CU
    constexpr int XROT = (16 - ((LP / 4) % 16)) % 16, YROT = (16 - (LP % 16)) % 16;   // lane -> column-group rotations
    const OutRegion rg = out_region(a);
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    Block4x4Epilogue<TX, TOut> epi;
    const int rgp = threadIdx.x >> 4, gy = ((threadIdx.x & 15) + YROT * (rgp & 1)) & 15;   // y-pass / output mapping
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
    constexpr int RPW = (LH + 3) / 4;                  // rows staged and x-filtered by each wave
    load_rows_wave<TIn, LH, LW, LP, RPW>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary);
    // taps: TP[p] = (h[p], h[p-1]),  HY[m] = (hy[2m], hy[2m+1]),  h = marginal taps 0..R of the class
    const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
    f2 TP[R + 1], HY[(R + 2) / 2];
#pragma unroll
    for (int t = 0; t <= R; ++t) TP[t] = (f2){ckx[t], t ? ckx[t - 1] : 0.f};
#pragma unroll
    for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
    wave_lds_fence();                                   // a wave reads back only rows it staged itself
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // ---- x pass, in place: each wave owns LH/4 rows; a wave instruction covers 4 rows x 16 groups ----
    {
        // consecutive rows are LP/4 sixteen-byte slots apart; the two rows that share a 32-lane half are
        // read conflict-free when the odd one starts XROT groups further along the row
        const int rsub = lane >> 4, g = ((lane & 15) + XROT * (rsub & 1)) & 15;
        for (int it = 0; it < (RPW + 3) / 4; ++it) {
            const int rr = wave * RPW + it * 4 + rsub;
            const bool ok = (it * 4 + rsub) < RPW && rr < LH;
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
    const int g = gy;
    f2 axy[4], azw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
    YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * LP + 4 * g, LP);
    float4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * g, acc);
}

constexpr size_t kTileLds = sizeof(float) * (GT + 2 * PB_KRAD) * (GT + 2 * PB_KRAD);   // 88 x 88 floats

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT, 5) void conv_tile_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only), so
    // give every XCD one contiguous run of tiles -- row-neighbours then share their halos in that
    // XCD's L2 instead of each fetching them from memory.  The grid is padded to a multiple of 8.
    const int chunk = gridDim.x >> 3;
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    // (integer division runs on the vector ALU even for uniform operands: pin the results back into SGPRs so that the
    // record pointer -- and with it every tap address in the stencil loops -- stays scalar)
    const int plane = __builtin_amdgcn_readfirstlane(tile_id / tiles_per_plane);
    const int local = tile_id - plane * tiles_per_plane;
    const pb_blur_info *info = a.info + __builtin_amdgcn_readfirstlane(plane / a.C);
    const PB_CONSTANT pb_blur_info *cinfo = as_constant(info);
    const bool sep = cinfo->separable != 0;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = a.force_full ? PB_KRAD : cinfo->radius;
    if (sep) {
        if (R <= 4) body_tile_sep<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else if (R <= 8) body_tile_sep<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else body_tile_sep<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        return;
    }
    if (R <= 4) body_tile<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else if (R <= 8) body_tile<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else body_tile<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, local, tiles_x, smem);
}

template <typename TIn, typename TX, typename TOut>
int launch_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    const long tpp = (long)tiles_x * tiles_y;
    const long blocks = tpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    const long grid = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_tile_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(NT), kTileLds, ctx->stream, p,
                       (int)tpp, tiles_x, (int)blocks);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// One launch per pass: each image's record decides on the device which body evaluates its tiles, so a batch
// may mix rank-1 and general kernels and the host never has to read the estimates back.
int pb_launch_conv(pb_ctx *ctx, const ConvPass &p) {
    ProfScope prof(ctx, PB_PROF_CONV);
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    typedef unsigned char u8;
    switch (key) {
        case 0: return launch_typed<float, float, float>(ctx, p);
        case 1: return launch_typed<float, float, __half>(ctx, p);
        case 3: return launch_typed<float, __half, float>(ctx, p);
        case 4: return launch_typed<float, __half, __half>(ctx, p);
        case 12: return launch_typed<__half, __half, float>(ctx, p);
        case 13: return launch_typed<__half, __half, __half>(ctx, p);
        // 8-bit images: first pass of the first iteration, the later passes that still read the 8-bit x, and the
        // store of the last pass (fp32 in between)
        case 24: return launch_typed<u8, u8, float>(ctx, p);
        case 6: return launch_typed<float, u8, float>(ctx, p);
        case 8: return launch_typed<float, u8, u8>(ctx, p);
        case 2: return launch_typed<float, float, u8>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: unsupported dtype combination %d", key);
    }
}
