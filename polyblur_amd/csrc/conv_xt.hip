// method='direct_separable' in ONE launch per Horner step: the x-t separable approximation of the image's Gaussian
// (pb_options.separable_approx in include/polyblur_hip.h; intent of separable_gaussian2d.cpp:91-183) evaluated as two
// 1-D filters on a tile that never leaves LDS:
//
//   xt_first = 1:  88 x 112 window  ->  1-D Gaussian along x, in place (rank-1 body's x pass)  ->  1-D Gaussian down an
//                  oblique line: for every row offset i the line sits m_i + f_i samples to the side, the two neighbours
//                  weighted g2[i] (1 - f_i) and g2[i] f_i (linear interpolation)  ->  Horner epilogue;
//   xt_first = 0:  the transpose: 112 x 88 window, 1-D Gaussian along y (into registers, then written back), then the
//                  oblique pass walks columns and interpolates between two rows.
//
// 75 multiply-adds per sample instead of the 625 of the exact oblique stencil, and one pass over memory per step: this is
// the north star's "separable Gaussian filters with LDS line staging" for the 28 of 30 directions whose exact kernel is
// not rank-1 -- as an APPROXIMATION (tests state its distance to the exact result), hence opt-in.
//
// Boundary: the intermediate image u = K1 * t lives on the padded domain like every other plane of the polynomial, so
// with the zero boundary ('direct' family) the samples of u outside the domain are zero: border tiles clear them after
// the first pass (with the circular boundary the window was staged through the wrap and they are right as computed).
#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

constexpr int XR = PB_KRAD;                 // reach of either 1-D filter
constexpr int XW = GT + 4 * XR;             // 112: the long side of the window (1-D pass + oblique reach)
constexpr int XS = GT + 2 * XR;             // 88: the short side
typedef float xf4 __attribute__((ext_vector_type(4)));
#define PB_LDS __attribute__((address_space(3)))
// volatile LDS accesses: emitted one by one, in program order, as ds_read with a 16-bit immediate offset off one address
// register (merged into ds_read2 pairs with 8-bit offsets the compiler needs an address register per row and, hoisting
// all of them to the top of an unrolled loop, spills)
typedef const volatile PB_LDS float lds_f1;
typedef const volatile PB_LDS f2 lds_f2;
typedef const volatile PB_LDS xf4 lds_f4;
// acc += w * v with the (wave-uniform) weight in a scalar register; opaque to the compiler, which would otherwise pair
// neighbouring FMAs into v_pk_fma_f32 behind a thicket of register moves
__device__ __forceinline__ void xt_fma(float &acc, float w, float v) { asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(w), "v"(v)); }
template <typename T> __device__ __forceinline__ T *to_lds(const float *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (T *)(PB_LDS float *)p;
#pragma clang diagnostic pop
}

template <int LP>
__device__ __forceinline__ void xpass_rows(float *smem, int row0, int nrows, int lane, const f2 (&TP)[XR + 1]) {
    // output column groups 0..15: 4 rows x 16 groups per wave instruction; then groups 16..21: 10 rows x 6 groups.  The
    // second sweep must follow the first (which still reads the chunks the second overwrites).
    constexpr int XROT = (16 - ((LP / 4) % 16)) % 16;
    {
        const int rsub = lane >> 4, g = ((lane & 15) + XROT * (rsub & 1)) & 15;
        for (int it = 0; it * 4 < nrows; ++it) {
            const bool ok = it * 4 + rsub < nrows;
            float *row = smem + (ok ? row0 + it * 4 + rsub : row0) * LP;
            f2 d[XR + 2];
#pragma unroll
            for (int q = 0; q < 1 + XR / 2; ++q) {
                const float4 t4 = *reinterpret_cast<const float4 *>(row + 4 * (g + q));
                d[2 * q] = (f2){t4.x, t4.y};
                d[2 * q + 1] = (f2){t4.z, t4.w};
            }
            f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};
            XPassR<XR, 0>::run(vxy, vzw, TP, d);
            wave_lds_fence();
            if (ok) *reinterpret_cast<float4 *>(row + 4 * g) = make_float4(vxy.x, vxy.y, vzw.x, vzw.y);
            wave_lds_fence();
        }
    }
    {
        constexpr int E = XR / 2, RB = 64 / E;
        const int lr = lane / E, g = 16 + (lane - lr * E);
        for (int it = 0; it * RB < nrows; ++it) {
            const bool ok = lr < RB && it * RB + lr < nrows;
            float *row = smem + (ok ? row0 + it * RB + lr : row0) * LP;
            f2 d[XR + 2];
#pragma unroll
            for (int q = 0; q < 1 + XR / 2; ++q) {
                const float4 t4 = *reinterpret_cast<const float4 *>(row + 4 * (g + q));
                d[2 * q] = (f2){t4.x, t4.y};
                d[2 * q + 1] = (f2){t4.z, t4.w};
            }
            f2 vxy = (f2){0.f, 0.f}, vzw = (f2){0.f, 0.f};
            XPassR<XR, 0>::run(vxy, vzw, TP, d);
            wave_lds_fence();
            if (ok) *reinterpret_cast<float4 *>(row + 4 * g) = make_float4(vxy.x, vxy.y, vzw.x, vzw.y);
            wave_lds_fence();
        }
    }
}

// after the first pass the tile holds u on an 88 x 88 patch whose sample (r, c) is padded position (py0 + r, px0 + c):
// zero what lies outside the padded domain (zero boundary only; border tiles only)
template <int LP>
__device__ __forceinline__ void clear_outside(float *smem, int py0, int px0, int Hp, int Wp) {
    if (py0 >= 0 && px0 >= 0 && py0 + XS <= Hp && px0 + XS <= Wp) return;
    for (int e = threadIdx.x; e < XS * XS; e += NT) {
        const int r = e / XS, c = e - r * XS;
        const int py = py0 + r, px = px0 + c;
        if (py < 0 || py >= Hp || px < 0 || px >= Wp) smem[r * LP + c] = 0.f;
    }
    __syncthreads();
}

template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ void body_xt_rows(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                             TOut *opl, int tile, int tiles_x, float *smem) {
    // xt_first = 1: window 88 rows x 112 columns, origin (oy0 - 12, ox0 - 24)
    constexpr int LP = XW;
    const OutRegion rg = out_region(a);
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rgp = tid >> 4, gy = tid & 15;             // 4 rows x 112 samples = 7 x 64 banks: no rotation needed
    Block4x4Epilogue<TX, TOut> epi;
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
    constexpr int RPW = XS / 4;
    load_rows_wave<TIn, XS, XW, LP, RPW>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - XR, ox0 - 2 * XR, a.boundary, a.pad);
    const PB_CONSTANT pb_blur_info *ci = as_constant(info);
    f2 TP[XR + 1];
#pragma unroll
    for (int t = 0; t <= XR; ++t) TP[t] = (f2){ci->xt_g1[t], t ? ci->xt_g1[t - 1] : 0.f};
    wave_lds_fence();
    xpass_rows<LP>(smem, wave * RPW, RPW, lane, TP);
    __syncthreads();
    if (a.boundary == PB_ZERO) clear_outside<LP>(smem, oy0 - XR, ox0 - XR, a.H + 2 * a.pad, a.W + 2 * a.pad);
    // u(py, px) now sits at smem[(py - oy0 + 12) * LP + (px - ox0 + 12)]
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
    const float *base = smem + (rgp * 4) * LP + 4 * gy;
    // the 25 (offset, weight, weight) triples ride in lanes 0..24 of three registers and are read back with
    // v_readlane: no scalar loads in the loop (they would force every LDS wait to a full drain)
    const int ltap = min(lane, PB_KSIZE - 1);
    const int mv = info->xt_m[ltap];
    const float wav = info->xt_wa[ltap], wbv = info->xt_wb[ltap];
    // For offset i the four outputs of a row need the five samples from column cb = 12 + m_i on: two aligned 16-byte
    // reads (scalar reads at this lane -> column stride would be 4-way bank-conflicted) and a wave-uniform choice among
    // the four alignments.  One offset ahead: the 8 chunks of offset i+1 are in flight while the 32 FMAs of offset i issue.
    xf4 lo[4], hi[4], nlo[4], nhi[4];
    int cb = XR + __builtin_amdgcn_readlane(mv, 0);
    {
        const float *p = base + (cb & ~3);
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[r] = *reinterpret_cast<const xf4 *>(p + r * LP); hi[r] = *reinterpret_cast<const xf4 *>(p + r * LP + 4); }
    }
#define PB_XT_ROW(A0, A1, A2, A3, A4)                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
        xt_fma(acc[r][0], wa, A0); xt_fma(acc[r][1], wa, A1); xt_fma(acc[r][2], wa, A2); xt_fma(acc[r][3], wa, A3); \
        xt_fma(acc[r][0], wb, A1); xt_fma(acc[r][1], wb, A2); xt_fma(acc[r][2], wb, A3); xt_fma(acc[r][3], wb, A4); \
    }
    // two offsets per trip, the chunk registers ping-ponging (a copy per offset would cost as many moves as FMAs)
#define PB_XT_OFFSET(I, LO, HI, NLO, NHI)                                                       \
    {                                                                                          \
        const float wa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wav), (I)));  \
        const float wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wbv), (I)));  \
        const int s = cb & 3;                                                                  \
        const int inext = min((I) + 1, PB_KSIZE - 1);                                          \
        cb = XR + __builtin_amdgcn_readlane(mv, inext);                                        \
        {                                                                                      \
            const float *p = base + inext * LP + (cb & ~3);                                    \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                NLO[r] = *reinterpret_cast<const xf4 *>(p + r * LP); NHI[r] = *reinterpret_cast<const xf4 *>(p + r * LP + 4); } \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (s == 0) { PB_XT_ROW(LO[r].x, LO[r].y, LO[r].z, LO[r].w, HI[r].x) }                 \
        else if (s == 1) { PB_XT_ROW(LO[r].y, LO[r].z, LO[r].w, HI[r].x, HI[r].y) }            \
        else if (s == 2) { PB_XT_ROW(LO[r].z, LO[r].w, HI[r].x, HI[r].y, HI[r].z) }            \
        else { PB_XT_ROW(LO[r].w, HI[r].x, HI[r].y, HI[r].z, HI[r].w) }                        \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#pragma unroll 1
    for (int i = 0; i < PB_KSIZE - 1; i += 2) {
        PB_XT_OFFSET(i, lo, hi, nlo, nhi)
        PB_XT_OFFSET(i + 1, nlo, nhi, lo, hi)
    }
    PB_XT_OFFSET(PB_KSIZE - 1, lo, hi, nlo, nhi)
#undef PB_XT_OFFSET
#undef PB_XT_ROW
    float4 out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, out);
}

template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ void body_xt_cols(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                             TOut *opl, int tile, int tiles_x, float *smem) {
    // xt_first = 0: window 112 rows x 88 columns, origin (oy0 - 24, ox0 - 12)
    constexpr int LP = XS;
    constexpr int YROT = (16 - (LP % 16)) % 16;
    constexpr int NB = XS / 4;                          // 22 x 22 blocks of 4 x 4 in the patch of u
    const OutRegion rg = out_region(a);
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    const int tid = threadIdx.x;
    constexpr int RPW = XW / 4;
    load_rows_wave<TIn, XW, XS, LP, RPW>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - 2 * XR, ox0 - XR, a.boundary, a.pad);
    const PB_CONSTANT pb_blur_info *ci = as_constant(info);
    f2 HY[(XR + 2) / 2];
#pragma unroll
    for (int m = 0; m < (XR + 2) / 2; ++m) HY[m] = (f2){ci->xt_g1[2 * m], 2 * m + 1 <= XR ? ci->xt_g1[2 * m + 1] : 0.f};
    __syncthreads();
    // ---- 1-D pass along y: 22 x 22 blocks over 256 threads in two rounds, held in registers until everybody has read ----
    int by[2], bx[2];
    bool have[2];
    by[0] = tid >> 4; bx[0] = ((tid & 15) + YROT * (by[0] & 1)) & 15; have[0] = true;
    constexpr int E = NB - 16, NRIGHT = NB * E;
    if (tid < NRIGHT) { by[1] = tid / E; bx[1] = 16 + (tid - by[1] * E); have[1] = true; }
    else { const int k = tid - NRIGHT; by[1] = 16 + (k >> 4); bx[1] = k & 15; have[1] = k < E * 16; }
    if (!have[1]) { by[1] = 0; bx[1] = 0; }
    float4 u[2][4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        f2 axy[4], azw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
        YPassR<XR, 0>::run(axy, azw, HY, smem + (by[rd] * 4) * LP + 4 * bx[rd], LP);
#pragma unroll
        for (int r = 0; r < 4; ++r) u[rd][r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
        if (have[rd])
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4 *>(smem + (by[rd] * 4 + r) * LP + 4 * bx[rd]) = u[rd][r];
    const int rgp = tid >> 4, gy = ((tid & 15) + YROT * (rgp & 1)) & 15;
    Block4x4Epilogue<TX, TOut> epi;
    epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
    __syncthreads();
    if (a.boundary == PB_ZERO) clear_outside<LP>(smem, oy0 - XR, ox0 - XR, a.H + 2 * a.pad, a.W + 2 * a.pad);
    // u(py, px) now sits at smem[(py - oy0 + 12) * LP + (px - ox0 + 12)]; the oblique pass walks columns: offset i along x,
    // the line m_i + f_i rows to the side
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
    const float *base = smem + (rgp * 4 + XR) * LP + 4 * gy;
    const int ltap = min(tid & 63, PB_KSIZE - 1);
    const int mv = info->xt_m[ltap];
    const float wav = info->xt_wa[ltap], wbv = info->xt_wb[ltap];
    // offset i along x: the four outputs read samples i .. i+3 of their row -- an alignment known at compile time (the
    // loop is unrolled) -- from the five rows m_i .. m_i+4: one or two aligned 16-byte reads per row
#define PB_XT_STEP(I)                                                                            \
    {                                                                                            \
        const int m_ = __builtin_amdgcn_readlane(mv, (I));                                       \
        const float wa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wav), (I)));    \
        const float wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wbv), (I)));    \
        const float *p_ = base + m_ * LP + ((I) & ~3);                                           \
        constexpr int s_ = (I) & 3;                                                              \
        xf4 lo_[5], hi_[5];                                                                      \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) {                                          \
            /* (the fifth row carries a zero weight when it would fall below the patch: keep the read inside) */ \
            const float *q_ = (r == 4 && m_ >= XR) ? p_ + 3 * LP : p_ + r * LP;                  \
            lo_[r] = *reinterpret_cast<const xf4 *>(q_);                                         \
            if (s_ != 0) hi_[r] = *reinterpret_cast<const xf4 *>(q_ + 4); else hi_[r] = lo_[r];  \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
            _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                      \
                const xf4 L = lo_[r + k], Hh = hi_[r + k];                                       \
                const float w_ = k ? wb : wa;                                                    \
                if (s_ == 0) { xt_fma(acc[r][0], w_, L.x); xt_fma(acc[r][1], w_, L.y); xt_fma(acc[r][2], w_, L.z); xt_fma(acc[r][3], w_, L.w); } \
                else if (s_ == 1) { xt_fma(acc[r][0], w_, L.y); xt_fma(acc[r][1], w_, L.z); xt_fma(acc[r][2], w_, L.w); xt_fma(acc[r][3], w_, Hh.x); } \
                else if (s_ == 2) { xt_fma(acc[r][0], w_, L.z); xt_fma(acc[r][1], w_, L.w); xt_fma(acc[r][2], w_, Hh.x); xt_fma(acc[r][3], w_, Hh.y); } \
                else { xt_fma(acc[r][0], w_, L.w); xt_fma(acc[r][1], w_, Hh.x); xt_fma(acc[r][2], w_, Hh.y); xt_fma(acc[r][3], w_, Hh.z); } \
            }                                                                                    \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
    PB_XT_STEP(0) PB_XT_STEP(1) PB_XT_STEP(2) PB_XT_STEP(3) PB_XT_STEP(4) PB_XT_STEP(5) PB_XT_STEP(6) PB_XT_STEP(7) PB_XT_STEP(8)
    PB_XT_STEP(9) PB_XT_STEP(10) PB_XT_STEP(11) PB_XT_STEP(12) PB_XT_STEP(13) PB_XT_STEP(14) PB_XT_STEP(15) PB_XT_STEP(16)
    PB_XT_STEP(17) PB_XT_STEP(18) PB_XT_STEP(19) PB_XT_STEP(20) PB_XT_STEP(21) PB_XT_STEP(22) PB_XT_STEP(23) PB_XT_STEP(24)
#undef PB_XT_STEP
    float4 out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, out);
}

constexpr size_t kXtLds = sizeof(float) * XS * XW;          // 39 424 B: four workgroups per CU

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT, 4) void conv_xt_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int chunk = gridDim.x >> 3;                          // XCD-aware order, as conv_tile_kernel
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int plane = __builtin_amdgcn_readfirstlane(tile_id / tiles_per_plane);
    const int local = tile_id - plane * tiles_per_plane;
    const pb_blur_info *info = a.info + __builtin_amdgcn_readfirstlane(plane / a.C);
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    if (as_constant(info)->xt_exact_rank1) return;              // exact and cheaper in the rank-1 body (conv.hip, same step)
    if (as_constant(info)->xt_first) body_xt_rows<TIn, TX, TOut>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else body_xt_cols<TIn, TX, TOut>(a, info, ipl, xpl, opl, local, tiles_x, smem);
}

template <typename TIn, typename TX, typename TOut>
int launch_xt(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    const long tpp = (long)tiles_x * tiles_y;
    const long blocks = tpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "x-t pass: bad grid");
    const long grid = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_xt_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(NT), kXtLds, ctx->stream, p, (int)tpp,
                       tiles_x, (int)blocks);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// One Horner step  out = scale (K2 (K1 in)) + coef x  with p.info = the SECOND record of every image's separable pair.
// Returns PB_ERR_UNSUPPORTED (nothing launched) for dtype combinations it is not instantiated for.
int pb_launch_conv_xt(pb_ctx *ctx, const ConvPass &p) {
    if (p.epilogue != EPI_HORNER) return PB_ERR_UNSUPPORTED;
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    if (key != 0 && key != 1 && key != 3 && key != 4 && key != 12) return PB_ERR_UNSUPPORTED;
    ProfScope prof(ctx, PB_PROF_CONV);
    switch (key) {
        case 0: return launch_xt<float, float, float>(ctx, p);
        case 1: return launch_xt<float, float, __half>(ctx, p);
        case 3: return launch_xt<float, __half, float>(ctx, p);
        case 4: return launch_xt<float, __half, __half>(ctx, p);
        default: return launch_xt<__half, __half, float>(ctx, p);
    }
}
