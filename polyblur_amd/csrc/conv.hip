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
// * dense kernels with enough live phases take a third body in a second launch of the same step (conv_fft.hip: the same
//   stencil evaluated per 64 x 64 window in the frequency domain inside LDS); this kernel's tiles of such images exit.
//
// No MFMA anywhere: the pass is bound by HBM / latency (rank-1) or by the fp32 vector rate (2-D stencil).

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {


// =============================================================================================
// general kernels: workgroup tile body
// =============================================================================================

// The general stencil is evaluated as a list of PHASES read from the image's record (estimate.hip: finish_record):
// phase = (kernel row dy, window chunk q).  In a phase a thread reads, for each of its four output rows r, the four
// window elements m = 4q .. 4q+3 of LDS row r + dy (one ds_read_b128) and feeds its four outputs c0 .. c0+3:
// element m meets the taps t[m+3], t[m+2], t[m+1], t[m] (t = the kernel row, zero padded by 3 on the left), i.e. two
// packed FMAs  (x,y) += (t[m+3], t[m+2]) * s,  (z,w) += (t[m+1], t[m]) * s  with the tap pair in an aligned SGPR pair
// read swapped through op_sel and the sample broadcast from its register.  The seven taps a chunk meets are loaded
// twice, from t[n0] and from t[n0+1], which gives both alignments of adjacent pairs.  Phases whose taps are all dead
// (exact zeros; under the adaptive policy also the corners outside the Gaussian's ellipse) are simply not in the list.
//
// The loop is software-pipelined by hand, half a phase ahead: while the 16 packed FMAs of two output rows issue, the
// window chunks of the next two rows (of this phase or the next) are in flight from LDS, and during a whole phase the
// taps of the next phase and the descriptor of the one after are in flight from the scalar cache.  Every wait is
// therefore for something issued at least 16 FMAs earlier, and is placed (wait_for) BEFORE the next prefetch is
// issued, so that the lgkmcnt(0) the mixed LDS / scalar traffic forces never drains a load that has only just been issued.
struct PhaseTaps { f2 a[3], b[3]; };      // a[k] = (t[n0+2k], t[n0+2k+1]),  b[k] = (t[n0+2k+1], t[n0+2k+2])

// (the odd-aligned pairs come from the record's second, shifted copy of the taps: building them from the first with
// scalar moves would put a wait for the scalar loads right behind their issue)
__device__ __forceinline__ void load_phase_taps(PhaseTaps &T, const PB_CONSTANT float *t, const PB_CONSTANT float *todd) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T.a[k] = (f2){t[2 * k], t[2 * k + 1]};
        T.b[k] = (f2){todd[2 * k], todd[2 * k + 1]};
    }
}
// a use of v the compiler cannot see through: whatever is still loading v is waited for HERE
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wait_for(f4 &v) { asm volatile("" : "+v"(v)); }

// The packed FMAs of one half-phase -- two output rows, 16 instructions -- as ONE asm statement (nothing is scheduled
// in between, no pad states are inserted), the four accumulators interleaved so that dependent FMAs are four issue
// slots apart.  Element j of a chunk feeds (x,y) with the tap pair starting at n0+j+2 and (z,w) with the pair starting
// at n0+j, both read swapped.  KIND 1: first chunk of a window row -- its elements 0, 1 meet only padding on the (z,w)
// side; KIND 2: last chunk -- elements 2, 3 meet only padding on the (x,y) side.
#define PB_FMA(acc, tap, dat, sel) "v_pk_fma_f32 " acc ", " tap ", " dat ", " acc " op_sel:[1," sel ",0] op_sel_hi:[0," sel ",1]\n\t"
template <int KIND>
__device__ __forceinline__ void chunk_fma2(f2 &axy0, f2 &azw0, f2 &axy1, f2 &azw1, const PhaseTaps &T, const f4 &v0, const f4 &v1) {
    const f2 lo0 = v0.xy, hi0 = v0.zw, lo1 = v1.xy, hi1 = v1.zw;
    // operands: 0-3 accumulators (xy0, zw0, xy1, zw1); 4-9 taps a0 a1 a2 b0 b1 b2; 10-13 data lo0 hi0 lo1 hi1
    if (KIND == 0)
        asm(PB_FMA("%0", "%5", "%10", "0") PB_FMA("%1", "%4", "%10", "0") PB_FMA("%2", "%5", "%12", "0") PB_FMA("%3", "%4", "%12", "0")
            PB_FMA("%0", "%8", "%10", "1") PB_FMA("%1", "%7", "%10", "1") PB_FMA("%2", "%8", "%12", "1") PB_FMA("%3", "%7", "%12", "1")
            PB_FMA("%0", "%6", "%11", "0") PB_FMA("%1", "%5", "%11", "0") PB_FMA("%2", "%6", "%13", "0") PB_FMA("%3", "%5", "%13", "0")
            PB_FMA("%0", "%9", "%11", "1") PB_FMA("%1", "%8", "%11", "1") PB_FMA("%2", "%9", "%13", "1") PB_FMA("%3", "%8", "%13", "1")
            : "+v"(axy0), "+v"(azw0), "+v"(axy1), "+v"(azw1)
            : "s"(T.a[0]), "s"(T.a[1]), "s"(T.a[2]), "s"(T.b[0]), "s"(T.b[1]), "s"(T.b[2]), "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1));
    if (KIND == 1)
        asm(PB_FMA("%0", "%5", "%10", "0") PB_FMA("%2", "%5", "%12", "0")
            PB_FMA("%0", "%8", "%10", "1") PB_FMA("%2", "%8", "%12", "1")
            PB_FMA("%0", "%6", "%11", "0") PB_FMA("%1", "%5", "%11", "0") PB_FMA("%2", "%6", "%13", "0") PB_FMA("%3", "%5", "%13", "0")
            PB_FMA("%0", "%9", "%11", "1") PB_FMA("%1", "%8", "%11", "1") PB_FMA("%2", "%9", "%13", "1") PB_FMA("%3", "%8", "%13", "1")
            : "+v"(axy0), "+v"(azw0), "+v"(axy1), "+v"(azw1)
            : "s"(T.a[0]), "s"(T.a[1]), "s"(T.a[2]), "s"(T.b[0]), "s"(T.b[1]), "s"(T.b[2]), "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1));
    if (KIND == 2)
        asm(PB_FMA("%0", "%5", "%10", "0") PB_FMA("%1", "%4", "%10", "0") PB_FMA("%2", "%5", "%12", "0") PB_FMA("%3", "%4", "%12", "0")
            PB_FMA("%0", "%8", "%10", "1") PB_FMA("%1", "%7", "%10", "1") PB_FMA("%2", "%8", "%12", "1") PB_FMA("%3", "%7", "%12", "1")
            PB_FMA("%1", "%5", "%11", "0") PB_FMA("%3", "%5", "%13", "0")
            PB_FMA("%1", "%8", "%11", "1") PB_FMA("%3", "%8", "%13", "1")
            : "+v"(axy0), "+v"(azw0), "+v"(axy1), "+v"(azw1)
            : "s"(T.a[0]), "s"(T.a[1]), "s"(T.a[2]), "s"(T.b[0]), "s"(T.b[1]), "s"(T.b[2]), "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1));
}
#undef PB_FMA

// descriptor: byte offset of the chunk in the LDS tile (of the record's radius class) | index of its first tap << 16
#define PB_TAP_OFF(d) ((unsigned)(d) >> 16)
#define PB_LDS_OFF(d) ((unsigned)(d) & 0xffffu)

struct PhaseCursor {
    const char *rowc;                       // LDS address of the current phase's row-0 chunk (this thread's window)
    unsigned d1, d2;                        // descriptors of the two phases after the current one
    const PB_CONSTANT unsigned *next;       // where the descriptor after those is read from
    const PB_CONSTANT float *taps, *taps_odd;
};

// One phase: TC = its taps, `cur` = its chunks of rows 0 and 1 (requested half a phase ago).  Two half-phases of 16
// packed FMAs; the first requests the NEXT phase's taps (into TN), the descriptor three phases ahead and the chunks of
// rows 2, 3; the second the next phase's rows 0, 1, which are left in `cur`.
template <int KIND, int LP>
__device__ __forceinline__ void run_phase(f2 (&axy)[4], f2 (&azw)[4], const PhaseTaps &TC, PhaseTaps &TN, f4 (&cur)[2],
                                          PhaseCursor &pc, const char *base) {
    f4 nxt[2];
    // rows 0, 1: their chunks were requested half a phase ago; request rows 2, 3 and the next phase's scalars
    wait_for(cur[0]); wait_for(cur[1]);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned d3 = *pc.next;
    load_phase_taps(TN, pc.taps + PB_TAP_OFF(pc.d1), pc.taps_odd + PB_TAP_OFF(pc.d1));
    nxt[0] = *reinterpret_cast<const f4 *>(pc.rowc + 2 * LP * 4);
    nxt[1] = *reinterpret_cast<const f4 *>(pc.rowc + 3 * LP * 4);
    __builtin_amdgcn_sched_barrier(0);
    chunk_fma2<KIND>(axy[0], azw[0], axy[1], azw[1], TC, cur[0], cur[1]);
    __builtin_amdgcn_sched_barrier(0);
    // rows 2, 3; request rows 0, 1 of the next phase
    wait_for(nxt[0]); wait_for(nxt[1]);
    __builtin_amdgcn_sched_barrier(0);
    pc.rowc = base + PB_LDS_OFF(pc.d1);
    cur[0] = *reinterpret_cast<const f4 *>(pc.rowc);
    cur[1] = *reinterpret_cast<const f4 *>(pc.rowc + LP * 4);
    __builtin_amdgcn_sched_barrier(0);
    chunk_fma2<KIND>(axy[2], azw[2], axy[3], azw[3], TC, nxt[0], nxt[1]);
    __builtin_amdgcn_sched_barrier(0);
    pc.d1 = pc.d2; pc.d2 = d3; ++pc.next;
}

// All `count` (even, > 0) phases of one kind.  On entry and on exit TA holds the taps of the cursor's current phase.
template <int KIND, int LP>
__device__ __forceinline__ void run_phases(f2 (&axy)[4], f2 (&azw)[4], PhaseTaps &TA, f4 (&cur)[2], int count, PhaseCursor &pc,
                                           const char *base) {
    PhaseTaps TB;
#pragma unroll 1
    for (int i = 0; i < count; i += 2) {
        run_phase<KIND, LP>(axy, azw, TA, TB, cur, pc, base);
        run_phase<KIND, LP>(axy, azw, TB, TA, cur, pc, base);
    }
}

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ void body_tile(const ConvPass &a, const pb_blur_info *info, const TIn *ipl, const TX *xpl,
                                          TOut *opl, int tile, int tiles_x, float *smem) {
    constexpr int LW = GT + 2 * R, LH = GT + 2 * R;
    constexpr int LP = LW;                     // unpadded LDS rows; conflicts avoided by YROT
    constexpr int YROT = (16 - (LP % 16)) % 16;  // odd row groups start YROT column groups further along the row
    constexpr int PR = 4;
    const OutRegion rg = out_region(a);
    const int ty = __builtin_amdgcn_readfirstlane(tile / tiles_x), tx = tile - ty * tiles_x;
    const int oy0 = rg.y_lo + ty * GT, ox0 = rg.x_lo + tx * GT;
    if (oy0 >= rg.y_hi) return;
    const int tid = threadIdx.x;
    // 16 column groups x 16 row groups.  The two row groups that share a 32-lane half are 4 LDS rows
    // (4 * LP floats) apart; rotating the odd one's column group keeps every ds_read_b128 conflict-free.
    const int rgp = tid >> 4, g = ((tid & 15) + YROT * (rgp & 1)) & 15;
    load_tile<TIn, LH, LW, LP>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary, a.pad);
    __syncthreads();
    f2 axy[PR], azw[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
    // gtaps[y] = {0,0,0, k[y][0..24], 0,0,0,0}; class R starts at kernel row / column 12 - R
    const PB_CONSTANT pb_blur_info *ci = as_constant(info);
    const PB_CONSTANT unsigned *plist = reinterpret_cast<const PB_CONSTANT unsigned *>(ci->phase);
    const char *base = reinterpret_cast<const char *>(smem + (rgp * PR) * LP + 4 * g);
    const int n_mid = ci->nphase[0], n_first = ci->nphase[1], n_last = ci->nphase[2];
    if (n_mid + n_first + n_last > 0) {
        PhaseCursor pc;
        const unsigned d0 = plist[0];
        pc.d1 = plist[1]; pc.d2 = plist[2]; pc.next = plist + 3;
        pc.taps = ci->gtaps;
        pc.taps_odd = ci->gtaps_odd;
        pc.rowc = base + PB_LDS_OFF(d0);
        PhaseTaps TA;
        load_phase_taps(TA, pc.taps + PB_TAP_OFF(d0), pc.taps_odd + PB_TAP_OFF(d0));
        f4 cur[2];
        cur[0] = *reinterpret_cast<const f4 *>(pc.rowc);
        cur[1] = *reinterpret_cast<const f4 *>(pc.rowc + LP * 4);
        if (n_mid) run_phases<0, LP>(axy, azw, TA, cur, n_mid, pc, base);
        if (n_first) run_phases<1, LP>(axy, azw, TA, cur, n_first, pc, base);
        if (n_last) run_phases<2, LP>(axy, azw, TA, cur, n_last, pc, base);
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
    load_rows_wave<TIn, LH, LW, LP, RPW>(smem, ipl, a.in_kind, a.in_pitch, a.H, a.W, oy0 - R, ox0 - R, a.boundary, a.pad);
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
__global__ __launch_bounds__(NT, 4) void conv_tile_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
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
    if (sep ? a.skip_sep : a.skip_general) return;                 // another launch of this step does this image
    if (sep && a.strip && cinfo->radius > 8) return;               // the streaming strip body (conv_strip.hip) does this image
    if (a.fsel) {                                                  // the tile-spectrum body (conv_fft.hip) does this image:
        const PB_CONSTANT pb_fft_sel *fs = as_constant(a.fsel + plane / a.C);
        if (fs->poly || (!sep && fs->use_fft)) return;             // its whole polynomial in one pass, or this step (dense kernels)
    }
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int R = cinfo->radius;
    if (sep) {
        if (R <= 4) body_tile_sep<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else if (R <= 8) body_tile_sep<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        else body_tile_sep<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, local, tiles_x, smem);
        return;
    }
    if (R <= 4) body_tile<TIn, TX, TOut, 4>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else if (R <= 6) body_tile<TIn, TX, TOut, 6>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else if (R <= 8) body_tile<TIn, TX, TOut, 8>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else if (R <= 10) body_tile<TIn, TX, TOut, 10>(a, info, ipl, xpl, opl, local, tiles_x, smem);
    else body_tile<TIn, TX, TOut, 12>(a, info, ipl, xpl, opl, local, tiles_x, smem);
}

template <typename TIn, typename TX, typename TOut>
int launch_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
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
static int launch_stencil(pb_ctx *ctx, const ConvPass &p);
static int launch_tile_spectrum(pb_ctx *ctx, const ConvPass &p);

// Dense kernels above the context's phase threshold take the tile-spectrum body: a second launch of the same step, in
// which -- as in the first -- every image's tiles exit at once unless the image's record selects that body.  For record
// sets the host built and read back itself (rec_cache) the launch no image needs is not issued at all.
// What the tile-spectrum body needs of a pass: planes within its 32-bit byte offsets, window counts within its index
// arithmetic.
static bool fft_pass_ok(const ConvPass &p) {
    const long plane_max = (1L << 31) - 4096;
    return !p.skip_general && p.in_plane * 4 < plane_max && p.x_plane * 4 < plane_max && p.out_plane * 4 < plane_max &&
           pb_conv_fft_feasible(p);
}

// What the host knows about the B records at `info` under the spec the pass in progress wants (ctx->poly_want): nullptr =
// nothing -- records built on the device, or a one-pass spec these records have not met yet.
static const pb_ctx::BodyFlags *known_flags(pb_ctx *ctx, const void *info, int B, const std::vector<pb_fft_sel> **sel = nullptr) {
    const auto it = ctx->rec_cache.find(info);
    if (it == ctx->rec_cache.end() || it->second.B != B) return nullptr;
    if (!ctx->poly_want.on) { if (sel) *sel = nullptr; return &it->second.plain; }
    if (!it->second.poly_valid || !same_spec(it->second.spec, ctx->poly_want)) return nullptr;
    if (sel) *sel = &it->second.sel;
    return &it->second.poly;
}
static pb_ctx::BodyFlags flags_of(const std::vector<pb_fft_sel> &h) {
    pb_ctx::BodyFlags f;
    for (const pb_fft_sel &e : h) {
        if (e.use_fft) { f.any_fft = true; if (!e.poly) f.any_fft3 = true; }
        else { f.any_other = true; if (e.strip) f.any_strip = true; else f.any_tile = true; }
    }
    return f;
}

int pb_launch_conv(pb_ctx *ctx, const ConvPass &p0) {
    ConvPass p = p0;
    bool fft = ctx->fft_min_phases >= 0 && !p.no_fft && fft_pass_ok(p);
    const int B = p.P / p.C;
    const std::vector<pb_fft_sel> *ksel = nullptr;
    const pb_ctx::BodyFlags *kf = known_flags(ctx, p.info, B, &ksel);
    const bool have = kf != nullptr;
    // (a step of a polynomial whose one-pass images another launch does: only the images that go step by step count)
    if (fft && have && !(ctx->poly_want.on && p.poly == 0 ? kf->any_fft3 : kf->any_fft)) fft = false;
    p.fsel = nullptr; p.khat = nullptr;
    if (fft || (have && ctx->poly_want.on)) {
        float *k = nullptr; pb_fft_sel *s = nullptr;
        // (spectra an earlier pass built count only while the scratch still holds them for these records)
        const bool built = (p.khat_ready || have || ctx->khat_by_estimate) && ctx->khat_owner == p.info && ctx->khat_B == B;   // (pb_build_khat also rebuilds spectra of another PolySpec)
        const int rc = pb_build_khat(ctx, p.info, B, &k, &s, !built);
        if (rc) return rc;
        p.khat = k; p.fsel = s;      // (the stencil bodies read the selection too: they skip the one-pass images)
    } else if (!p.khat_ready && !have) {
        ctx->khat_owner = nullptr; ctx->khat_B = 0;     // device-built records took a pass without spectra: whatever the scratch holds is not theirs
    }
    // rank-1 kernels of full support on fp32 planes: the streaming strip body (host-built records only: with device-built
    // ones it would be one more launch that usually finds no work)
    const bool strip = ctx->strip_mode && have && kf->any_strip && p.in_dtype == PB_F32 && p.x_dtype == PB_F32 &&
                       p.out_dtype == PB_F32 && p.epilogue == EPI_HORNER && p.pad == PB_KRAD && !p.skip_sep;
    bool strip_done = false;
#ifdef PB_EXPERIMENTAL      // (python -m polyblur_amd.build --experimental: conv_strip.hip is a measured experiment, NOTEBOOK.md)
    if (strip) {
        p.strip = 1;
        const int rc = pb_launch_conv_strip(ctx, p);
        if (rc == PB_OK) strip_done = true;
        else if (rc != PB_ERR_UNSUPPORTED) return rc;
        else p.strip = 0;
    }
#else
    (void)strip;
#endif
    // (PolySpec.always == 2, records of the estimation: every image takes the window form -- no stencil launch to find that out)
    const bool windows_only = !have && fft && !ctx->poly_want.on && ctx->poly_want.always == 2 && ctx->fft_wave && pb_conv_wfft_types(p) &&
                              pb_conv_wfft_feasible(p, false, ctx->poly_min_area);
    const bool tile_needed = windows_only ? false : (!have || (strip_done ? kf->any_tile : kf->any_other));
    if (tile_needed) {
        const int rc = launch_stencil(ctx, p);
        if (rc) return rc;
    }
    if (!fft) return PB_OK;
    ctx->known_sel = ksel;
    const int rc = launch_tile_spectrum(ctx, p);
    ctx->known_sel = nullptr;
    return rc;
}

// the tile-spectrum launch of a pass: the wave form where it is built and worth it, else the workgroup form -- which cannot
// run one-pass images whose composite halo is beyond its classes (PolySpec.on == 2 is only asked for where the wave form
// takes every launch, pb_poly_spec)
static int launch_tile_spectrum(pb_ctx *ctx, const ConvPass &p) {
    if (ctx->fft_wave) {
        const int rc = pb_launch_conv_wfft(ctx, p);
        if (rc != PB_ERR_UNSUPPORTED) return rc;
    }
    if (p.poly != 0 && pb_spec_of_spectra(ctx, p.khat).on >= 2)
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "one-pass polynomial with per-axis halos: the wave form does not take this pass");
    return pb_launch_conv_fft(ctx, p);
}

// Which one-pass form a polynomial with these steps may ask for (PolySpec.on): 2 = the composite filter's own halos, where
// every tile-spectrum launch of the polynomial goes to the wave form (conv_wfft.hip); 3 = as 2, and on 128 x 128 windows
// where those are cheaper still (conv_w128.hip: fp32 / fp16 planes, a launch of its own); 1 = kernels
// within the 4-sample halo class only (either form); 0 = never (ctx->poly_mode == 0, or a step the form does not suit).
int pb_poly_spec_mode(pb_ctx *ctx, const ConvPass *steps) {
    if (ctx->poly_mode == 0 || ctx->fft_min_phases < 0) return 0;
    // (the zero boundary, method='direct': one window pass for the interior plus three ring steps -- pb_launch_conv_poly --
    // where PolySpec.always holds; the caller drops the spec otherwise)
    // (... and only for images whose three-step pass is several rounds of window pairs: a ring step is one or two rounds of
    // ~20 us whatever its area, so on a 1080p image -- 2100 pairs, one round per step -- window pass + ring is SLOWER than three
    // plain steps: 0.583 against 0.504 ms per call, 720p 0.467 against 0.392; at 4K, 8085 pairs, 1.02 against 1.16.  The rule
    // looks at one image, never at the batch: what an image gets must not depend on the batch it travels in)
    bool zero_ok = ctx->zero_ring != 0;
    if (zero_ok && steps[0].boundary == PB_ZERO) {
        const int Hp = steps[0].H + 2 * steps[0].pad, Wp = steps[0].W + 2 * steps[0].pad;
        const long pairs3 = (long)(((Wp + 39) / 40 + 1) / 2) * ((Hp + 39) / 40) * steps[0].C;
        zero_ok = pairs3 >= ctx->zero_ring_min_pairs;
    }
    for (int s = 0; s < 3; ++s)
        if ((steps[s].boundary != PB_WRAP && !(steps[s].boundary == PB_ZERO && zero_ok)) || steps[s].boundary != steps[0].boundary ||
            steps[s].epilogue != EPI_HORNER || !fft_pass_ok(steps[s])) return 0;
    // (after an edgetaper the polynomial's operand is the padded, tapered plane: the window loaders read padded planes like
    // virtual ones -- every Horner step but the first always did -- PB_POLY_PADDED=0: three steps there, as in rounds 3 - 4)
    if (steps[0].in_kind != SRC_VIRTUAL && !ctx->poly_padded) return 0;
    // who runs the one pass: the first step's launch where it stores the type the last step stores (ConvPass.poly = 2), else
    // a composite launch from the first step's input to the last step's output -- whose types must be built
    ConvPass pc = steps[0];
    pc.out_dtype = steps[2].out_dtype;
    const bool fold = steps[0].out_dtype == steps[2].out_dtype;
    bool wave = ctx->poly_mode != 1 && ctx->fft_wave && (fold || pb_conv_wfft_types(pc));
    for (int s = 0; s < 3; ++s) wave = wave && pb_conv_wfft_types(steps[s]);
    // (job lists sized for the smallest one-pass tiles must fit the grid: a batch too large for them keeps the forms whose
    // lists are shorter -- three Horner steps in the end -- instead of failing the call)
    if (wave && !pb_conv_wfft_feasible(pc, true, ctx->poly_min_area)) wave = false;
    if (wave) return (ctx->poly_mode >= 3 && ctx->poly_cost128 > 0.f && pb_conv_w128_types(pc.in_dtype, pc.out_dtype) &&
                      pb_conv_w128_feasible(pc)) ? 3 : 2;
    return (fold || pb_conv_fft_types(pc)) ? 1 : 0;
}

// Whether the three Horner steps of a polynomial can all go to the wave form of the tile-spectrum body (PolySpec.always == 2)
bool pb_poly_three_steps_ok(pb_ctx *ctx, const ConvPass *steps) {
    if (ctx->poly_mode == 0 || ctx->fft_min_phases < 0 || !ctx->fft_wave) return false;
    for (int s = 0; s < 3; ++s)
        if (steps[s].epilogue != EPI_HORNER || !fft_pass_ok(steps[s]) || !pb_conv_wfft_types(steps[s]) || !pb_conv_wfft_feasible(steps[s], false, ctx->poly_min_area))
            return false;
    return true;
}

// The three Horner steps of one polynomial.  Whether the tile-spectrum body may be used is decided ONCE, from all three
// geometries (a body decided per step could meet spectra no earlier step had built); its spectra are built by the first
// step.  One launch sequence per step.
//
// Records built on the device (the pipeline) leave the host ignorant of which body an image takes, so every step issues
// both kernels and, for the usual all-dense or all-separable batch, one of them finds no work: three launches of a few
// thousand workgroups that leave at once, 6 us each on the critical path.  The two kernels of a step touch different
// images, so the stencil bodies' three launches go to the context's side stream -- forked behind the spectra, joined
// before the call returns -- and run (or evaporate) beside the tile-spectrum launches instead of between them.
int pb_launch_conv_poly(pb_ctx *ctx, const ConvPass *steps) {
    bool fft = ctx->fft_min_phases >= 0;
    for (int s = 0; s < 3; ++s) fft = fft && fft_pass_ok(steps[s]);
    const int B = steps[0].P / steps[0].C;
    // Records the host built (pb_make_kernels / pb_set_kernels) meeting a one-pass spec for the first time: the spectra of
    // that spec are built and the device's choice read back now -- one synchronisation per record set and spec -- so that
    // every later polynomial on them issues exactly the launches it needs, with job grids of exactly the size it needs.
    {
        const auto known = ctx->rec_cache.find(steps[0].info);
        if (fft && ctx->poly_want.on && known != ctx->rec_cache.end() && known->second.B == B &&
            !(known->second.poly_valid && same_spec(known->second.spec, ctx->poly_want))) {
            float *k0 = nullptr; pb_fft_sel *s0 = nullptr;
            int rc0 = pb_build_khat(ctx, steps[0].info, B, &k0, &s0, true);
            if (rc0) return rc0;
            std::vector<pb_fft_sel> h((size_t)B);
            PB_HIP(hipMemcpyAsync(h.data(), s0, sizeof(pb_fft_sel) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
            PB_HIP(hipStreamSynchronize(ctx->stream));
            pb_ctx::RecFlags &rf = known->second;
            rf.poly_valid = true; rf.spec = ctx->poly_want; rf.poly = flags_of(h); rf.sel = h;
        }
    }
    const bool have = known_flags(ctx, steps[0].info, B) != nullptr;
    // (per-launch profiling keeps everything on one stream: events on the side stream would time its launches' wait
    // behind the other kernel's workgroups, not their work)
    // One-pass polynomial: the images whose spectrum is the polynomial's take ONE window pass from the first step's input
    // to the last step's output; the later steps' launches skip them.  When the first and the last step store the same
    // type, the first step's launch takes them along (ConvPass.poly = 2: no launch of their own, nothing to pay when no
    // image qualifies); otherwise a composite launch does them.  (The spectra the steps meet are those ctx->poly_want
    // asks for: pb_build_khat rebuilds any others.)
    // PolySpec.always == 2 (records of the estimation): every image on three window steps, three launches of the wave body
    if (fft && !ctx->poly_want.on && ctx->poly_want.always == 2 && !have) {
        float *k = nullptr; pb_fft_sel *sel = nullptr;
        const bool built = ctx->khat_by_estimate && ctx->khat_owner == steps[0].info && ctx->khat_B == B;
        int rc = pb_build_khat(ctx, steps[0].info, B, &k, &sel, !built);
        for (int s = 0; s < 3 && !rc; ++s) {
            ConvPass p = steps[s];
            p.khat = k; p.fsel = sel; p.poly = 0;
            rc = pb_launch_conv_wfft(ctx, p);
            if (rc == PB_ERR_UNSUPPORTED) rc = pb_fail(ctx, PB_ERR_UNSUPPORTED, "three window steps: the wave form does not take this pass");
        }
        return rc;
    }
    const bool poly_on = fft && ctx->poly_want.on;
    const bool fold = poly_on && steps[0].out_dtype == steps[2].out_dtype;
    auto first_step = [&](ConvPass &p) {
        if (!fold) return;
        p.poly = 2;
        p.out2 = steps[2].out; p.out2_kind = steps[2].out_kind; p.out2_pitch = steps[2].out_pitch; p.out2_plane = steps[2].out_plane;
        p.clamp2 = steps[2].clamp01;
    };
    auto composite = [&](float *k, pb_fft_sel *sel) -> int {
        ConvPass pc = steps[0];
        pc.out = steps[2].out; pc.out_kind = steps[2].out_kind; pc.out_dtype = steps[2].out_dtype;
        pc.out_pitch = steps[2].out_pitch; pc.out_plane = steps[2].out_plane;
        pc.scale = 1.f; pc.coef = 0.f; pc.clamp01 = steps[2].clamp01; pc.poly = 1;
        pc.khat = k; pc.fsel = sel;
        if (!fft_pass_ok(pc)) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "one-pass polynomial: composite pass not feasible");
        return launch_tile_spectrum(ctx, pc);
    };
    // the images whose one pass runs on 128 x 128 windows (pb_fft_sel.poly == 2): a launch of their own, beside the others
    auto composite128 = [&](float *k, pb_fft_sel *sel) -> int {
        if (ctx->poly_want.on != 3) return PB_OK;
        ConvPass pc = steps[0];
        pc.out = steps[2].out; pc.out_kind = steps[2].out_kind; pc.out_dtype = steps[2].out_dtype;
        pc.out_pitch = steps[2].out_pitch; pc.out_plane = steps[2].out_plane;
        pc.scale = 1.f; pc.coef = 0.f; pc.clamp01 = steps[2].clamp01; pc.poly = 1;
        pc.khat = k; pc.fsel = sel;
        const std::vector<pb_fft_sel> *ksel = nullptr;
        (void)known_flags(ctx, steps[0].info, B, &ksel);
        ctx->known_sel = ksel;
        const int rc = pb_launch_conv_w128(ctx, pc);
        ctx->known_sel = nullptr;
        return rc;
    };
    // The side stream pays where the launches that may find no work are expensive -- one workgroup per 64 x 64 tile of every
    // plane: 6 us at 4K, 33 us for 32 x 1080p -- against two queue-to-queue waits per polynomial (~5 - 8 us each); a single
    // image of up to 4K runs everything on the caller's stream (measured: 4K 0.738 -> 0.722 ms of device time per call, while
    // 32 x 1080p would go from 6.09 to 6.61 ms without the side stream).
    // PolySpec.always (records of the estimation, nothing the host has read back): every image takes one window pass, on 64 x 64
    // or on 128 x 128 windows -- two launches on the caller's stream, each skipping the other's images, and nothing else.
    if (poly_on && ctx->poly_want.always && !have) {
        float *k = nullptr; pb_fft_sel *sel = nullptr;
        const bool built = ctx->khat_by_estimate && ctx->khat_owner == steps[0].info && ctx->khat_B == B;
        int rc = pb_build_khat(ctx, steps[0].info, B, &k, &sel, !built);
        if (rc) return rc;
        // The zero boundary: the window pass is the polynomial of the zero-EXTENDED image, which is the three-step result
        // (every step truncated to the padded domain, filters.py:40-49) everywhere but within 24 samples of the padded border
        // -- 12 of the image's.  That frame is recomputed by three Horner steps over the ring of window pairs it depends on
        // (conv_wfft.hip: ring_live; the kernels' own spectra in a second scratch set), the last of which overwrites the
        // frame's tiles of the output.  A ring step is a race of single window pairs (one or two rounds of ~20 us whatever
        // its size), so the first two -- which touch neither the output nor the first set of spectra -- run on the side
        // stream BESIDE the window pass; only the third waits for both.
        const bool zero = steps[0].boundary == PB_ZERO;
        const bool ring_aside = zero && ctx->aux && !ctx->prof_on && ctx->zero_ring_aside;
        auto ring_steps = [&](int s0, int s1, float *k2, pb_fft_sel *sel2) -> int {
            int e = PB_OK;
            for (int s = s0; s < s1 && !e; ++s) {
                ConvPass p = steps[s];
                p.khat = k2; p.fsel = sel2; p.ring = s + 1; p.poly = 0;
                e = pb_launch_conv_wfft(ctx, p);
                if (e == PB_ERR_UNSUPPORTED) e = pb_fail(ctx, PB_ERR_UNSUPPORTED, "zero-boundary ring: the wave form does not take this pass");
            }
            return e;
        };
        float *k2 = nullptr; pb_fft_sel *sel2 = nullptr;
        int rc_side = PB_OK;
        if (ring_aside) {
            PB_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            PB_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->aux;
            rc_side = pb_build_khat_ring(ctx, steps[0].info, B, &k2, &sel2);
            if (!rc_side) rc_side = ring_steps(0, 2, k2, sel2);
            ctx->stream = main_stream;
            PB_HIP(hipEventRecord(ctx->ev_join, ctx->aux));
        }
        rc = composite128(k, sel);
        ConvPass pc = steps[0];
        pc.out_dtype = steps[2].out_dtype;
        if (!rc) {
            if (pb_conv_wfft_types(pc)) rc = composite(k, sel);
            else {
                ConvPass p = steps[0];                           // (the composite's types are not built: the first step's launch takes them along)
                p.khat = k; p.fsel = sel;
                first_step(p);
                rc = launch_tile_spectrum(ctx, p);
            }
        }
        if (ring_aside) PB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));     // (joined on every way out)
        if (rc || rc_side || !zero) return rc ? rc : rc_side;
        if (!ring_aside) {
            rc = pb_build_khat_ring(ctx, steps[0].info, B, &k2, &sel2);
            if (!rc) rc = ring_steps(0, 2, k2, sel2);
            if (rc) return rc;
        }
        return ring_steps(2, 3, k2, sel2);
    }
    const long side_min_tiles = ctx->side_min_tiles;
    const long stencil_tiles = (long)((steps[0].H + 2 * steps[0].pad + 63) / 64) * ((steps[0].W + 2 * steps[0].pad + 63) / 64) * steps[0].P;
    const bool side = ctx->aux && stencil_tiles >= side_min_tiles;
    if (!fft || have || !side || ctx->prof_on) {
        if (fft && ctx->poly_want.on == 3) {
            float *k = nullptr; pb_fft_sel *sel = nullptr;
            const bool built = (have || ctx->khat_by_estimate) && ctx->khat_owner == steps[0].info && ctx->khat_B == B;
            int rc = pb_build_khat(ctx, steps[0].info, B, &k, &sel, !built);
            if (rc) return rc;
            rc = composite128(k, sel);
            if (rc) return rc;
        }
        for (int s = 0; s < 3; ++s) {
            ConvPass p = steps[s];
            p.khat_ready = s > 0;
            if (!fft) p.no_fft = 1;
            if (s == 0) first_step(p);
            const int rc = pb_launch_conv(ctx, p);
            if (rc) return rc;
        }
        if (poly_on && !fold) {
            // (the spectra the three steps have just used -- the first set's, or the second's behind an edgetaper -- and the spec they
            // were built under: images with a one-pass record wait for this launch)
            float *k = nullptr; pb_fft_sel *sel = nullptr;
            const int rc = pb_build_khat(ctx, steps[0].info, B, &k, &sel, false);
            if (rc) return rc;
            if (pb_spec_of_spectra(ctx, k).on) return composite(k, sel);
        }
        return PB_OK;
    }
    float *k = nullptr; pb_fft_sel *sel = nullptr;
    // (records the estimation has just built bring their spectra with them: blur_params_kernel ends with them)
    int rc = pb_build_khat(ctx, steps[0].info, B, &k, &sel, !(ctx->khat_by_estimate && ctx->khat_owner == steps[0].info && ctx->khat_B == B));
    if (rc) return rc;
    PB_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    PB_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    hipStream_t main_stream = ctx->stream;
    // Which launches stay on the caller's stream: those that probably do the work -- crossing to the side stream and back
    // costs two queue-to-queue waits (~5 us each) on the critical path.  Where the spec admits 128 x 128 windows (images of
    // poly_min_pairs128 window pairs or more: decided from the sizes alone, api.hip) that is the 128 x 128 launch, and the
    // wave body's three launches join the stencil launches on the side stream; otherwise the wave body's.
    const int main_env = ctx->main_stream_body;     // 0 = wave body, 1 = 128 x 128
    const bool w128_main = poly_on && ctx->poly_want.on == 3 && ctx->poly_want.cost128 > 0.f && main_env != 0;
    auto wave_steps = [&]() {
        for (int s = 0; s < 3 && !rc; ++s) {
            ConvPass p = steps[s];
            p.khat = k; p.fsel = sel;
            if (s == 0) first_step(p);
            rc = launch_tile_spectrum(ctx, p);
        }
    };
    ctx->stream = ctx->aux;
    // (the composite pass touches other images than the steps' launches do: it, too, runs -- or finds no work -- beside them)
    if (poly_on && !w128_main) rc = composite128(k, sel);
    if (poly_on && !fold && !rc) rc = composite(k, sel);
    if (w128_main) wave_steps();
    for (int s = 0; s < 3 && !rc; ++s) {
        ConvPass p = steps[s];
        p.khat = k; p.fsel = sel;
        rc = launch_stencil(ctx, p);
    }
    ctx->stream = main_stream;
    // (whatever was queued on the side stream is joined on every path out of here: later calls share the scratch planes)
    PB_HIP(hipEventRecord(ctx->ev_join, ctx->aux));
    if (w128_main) { if (!rc) rc = composite128(k, sel); }
    else wave_steps();
    PB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return rc;
}

// The host has just (re)built these B records and is synchronising anyway: build their spectra, read the per-image choice
// back and remember it.
int pb_cache_records(pb_ctx *ctx, const pb_blur_info *info, int B) {
    pb_forget_records(ctx, info, B);
    if (ctx->fft_min_phases < 0) return PB_OK;
    float *k = nullptr; pb_fft_sel *s = nullptr;
    int rc = pb_build_khat(ctx, info, B, &k, &s, true);
    if (rc) return rc;
    std::vector<pb_fft_sel> h(B);
    PB_HIP(hipMemcpyAsync(h.data(), s, sizeof(pb_fft_sel) * B, hipMemcpyDeviceToHost, ctx->stream));
    PB_HIP(hipStreamSynchronize(ctx->stream));
    pb_ctx::RecFlags f;
    f.B = B; f.plain = flags_of(h); f.poly_valid = false; f.spec = no_poly();
    ctx->rec_cache[info] = f;
    return PB_OK;
}
// a write of `bytes` bytes at dst: forget what is known about the record sets it overlaps
void pb_forget_range(pb_ctx *ctx, const void *dst, size_t bytes) {
    const char *lo = static_cast<const char *>(dst), *hi = lo + bytes;
    for (auto it = ctx->rec_cache.begin(); it != ctx->rec_cache.end();) {
        const char *a = static_cast<const char *>(it->first), *b = a + sizeof(pb_blur_info) * (size_t)it->second.B;
        if (a < hi && lo < b) {
            if (ctx->khat_owner == it->first) { ctx->khat_owner = nullptr; ctx->khat_B = 0; ctx->khat_by_estimate = false; }
            it = ctx->rec_cache.erase(it);
        } else ++it;
    }
    for (auto it = ctx->flip_sets.begin(); it != ctx->flip_sets.end();) {
        const char *a = static_cast<const char *>(it->first), *b = a + sizeof(pb_blur_info) * (size_t)it->second.B;
        if (a < hi && lo < b) it = ctx->flip_sets.erase(it); else ++it;
    }
}
void pb_forget_records(pb_ctx *ctx, const void *info, int B) {
    if (info) {
        pb_forget_range(ctx, info, sizeof(pb_blur_info) * (size_t)B);
        // (device-built records are not in the cache, but the spectra scratch may be theirs)
        const char *a = static_cast<const char *>(info), *b = a + sizeof(pb_blur_info) * (size_t)B, *o = static_cast<const char *>(ctx->khat_owner);
        // (spectra of a run of records that overlaps the rewritten ones anywhere, not only at its first record)
        const char *oe = o ? o + sizeof(pb_blur_info) * (size_t)ctx->khat_B : nullptr;
        if (o && o < b && a < oe) { ctx->khat_owner = nullptr; ctx->khat_B = 0; ctx->khat_by_estimate = false; }
        const char *o2 = static_cast<const char *>(ctx->khat2_owner);
        const char *oe2 = o2 ? o2 + sizeof(pb_blur_info) * (size_t)ctx->khat2_B : nullptr;
        if (o2 && o2 < b && a < oe2) { ctx->khat2_owner = nullptr; ctx->khat2_B = 0; }
        return;
    }
    ctx->rec_cache.clear();
    ctx->flip_sets.clear();
    ctx->khat2_owner = nullptr; ctx->khat2_B = 0;
    ctx->khat_owner = nullptr;
    ctx->khat_B = 0;
    ctx->khat_by_estimate = false;
}

static int launch_stencil(pb_ctx *ctx, const ConvPass &p) {
    ProfScope prof(ctx, PB_PROF_CONV);
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    typedef unsigned char u8;
    switch (key) {
        case 0: return launch_typed<float, float, float>(ctx, p);
        case 1: return launch_typed<float, float, __half>(ctx, p);
        case 3: return launch_typed<float, __half, float>(ctx, p);
        case 4: return launch_typed<float, __half, __half>(ctx, p);
        // fp16 Horner temporaries (pb_options.half_temporaries) next to an fp32 x operand
        case 9: return launch_typed<__half, float, float>(ctx, p);
        case 10: return launch_typed<__half, float, __half>(ctx, p);
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
