// Streaming body of the reblurring pass for rank-1 (separable) kernels: one WAVE walks down a column strip.
//
// Same pass, same operands, same result as the rank-1 body of conv.hip (one Horner step  t <- K*t + coef*x  of the
// polynomial deconvolution, reference deblurring.py:122-138 / :141-169, with the boundary models of filters.py:14-49; the
// x-then-y separable evaluation is the intent of separable_gaussian2d.cpp:47-88).  The tile body stages (64 + 24)^2
// samples per 64^2 outputs -- 1.89 x the bytes, 1.375 x the x-pass rows -- and is bound by its L1 miss path (DESIGN.md
// section 4).  Here a wave owns a strip of 232 output columns (256 input columns = ONE 1-KiB LDS-DMA instruction per
// input row, lane l <-> input columns 4l .. 4l+3) and streams down `seg_h` output rows:
//
//   input row i   global -> LDS ring (5 rows), requested 4 rows ahead          1 vector-memory instruction
//   x pass        7 x ds_read_b128 (the lane's 28-sample window), 52 packed FMAs, taps in scalar registers
//   ring          the x-filtered row (4 samples per lane) replaces the oldest of 25 rows held in REGISTERS
//   y pass        25 rows x 4 samples from the register ring, 50 packed FMAs -- no LDS traffic at all
//   epilogue      scale, + coef * x (x operand rows travel through a second LDS ring), clamp, one 16-byte store per lane
//
// The vertical halo is paid once per segment (24 of seg_h + 24 input rows), the horizontal one is 24 of 256 columns:
// 1.34 x the input bytes at seg_h = 110 instead of 1.89 x, and per output row a wave issues three vector-memory
// instructions.  The 25-row ring is indexed statically (the row loop is unrolled 25 times).  Every step issues the same
// vector-memory instructions in the same order (rows or lanes without work use an out-of-range offset), so the wait for
// "the row requested six steps ago" is a constant vmcnt.  Strips that touch the plane's border fetch their rows sample by
// sample through the boundary model (4-byte LDS-DMA, four instructions per row) and store sample by sample.
// No MFMA, no library.

#include <cstdlib>
#include <type_traits>

#include "conv_fft_common.h"
#include "conv_tile_common.h"

namespace {

constexpr int ST_W = 232;                  // output columns per strip
constexpr int ST_IN = 256;                 // input columns per strip (ST_W + 24)
constexpr int ST_SLOTS = 5;                // LDS ring: rows (25 % 5 == 0: a step's slot is a compile-time constant)
constexpr int ST_AHEAD = 4;                // rows requested ahead
constexpr int ST_R = PB_KRAD;              // 12: the body always evaluates 25 taps per axis
constexpr size_t kStripLds = 2 * ST_SLOTS * ST_IN * sizeof(float);      // input ring + x-operand ring

struct StripGeom {
    int strips_x, segs_y, seg_h, per;      // strips per row of segments, segments per plane, output rows per segment, jobs per plane and XCD
    float inv_strips, inv_per;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
typedef __attribute__((address_space(3))) void st_lds_void;
__device__ __forceinline__ void st_dma16(brsrc r, char *dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (st_lds_void *)(__attribute__((address_space(3))) char *)dst, 16, (int)voffset, soffset, 0, 0);
}
__device__ __forceinline__ void st_dma4(brsrc r, char *dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (st_lds_void *)(__attribute__((address_space(3))) char *)dst, 4, (int)voffset, soffset, 0, 0);
}
#pragma clang diagnostic pop

// the x pass of conv_tile_common.h (XPassR) on two accumulators per output pair: a lone wave issues dependent packed FMAs
// eight cycles apart, four independent chains keep the pipe full
template <int R, int J> struct XPass4 {
    static __device__ __forceinline__ void run(f2 (&vxy)[2], f2 (&vzw)[2], const f2 (&TP)[R + 1], const f2 (&d)[R + 2]) {
        if constexpr (J <= 2 * R + 1) XTapApply<R, J, J & 1>::run(vxy[(J >> 1) & 1], TP, d[J >> 1]);
        if constexpr (J >= 2) XTapApply<R, J - 2, J & 1>::run(vzw[(J >> 1) & 1], TP, d[J >> 1]);
        if constexpr (J < 2 * R + 3) XPass4<R, J + 1>::run(vxy, vzw, TP, d);
    }
};

#ifdef PB_ST_TRACE
// debug build only: per wave, cycles spent in the vmcnt wait, in the LDS reads, in all steps; steps (tools/st_trace.py)
__device__ unsigned long long g_st_trace[4096 * 6];
#define PB_ST_CLK() __builtin_readcyclecounter()
#endif

typedef float sf4 __attribute__((ext_vector_type(4)));
typedef unsigned su4 __attribute__((ext_vector_type(4)));

template <bool INTERIOR>
__device__ __forceinline__ void strip_body(const ConvPass &a, const pb_blur_info *info, int plane, int sx, int sy, const StripGeom &g,
                                           char *zb) {
    constexpr int OPS = INTERIOR ? 3 : 12;                     // vector-memory instructions per step
    const int lane = threadIdx.x & 63;
    const OutRegion rg = out_region(a);
    const int Hp = a.H + 2 * a.pad;
    const int ox0 = rg.x_lo + sx * ST_W, oy0 = rg.y_lo + sy * g.seg_h;
    const int nout = min(g.seg_h, rg.y_hi - oy0), nrows = nout + 2 * ST_R;
    const float *ipl = static_cast<const float *>(a.in) + (long)plane * a.in_plane;
    const float *xpl = static_cast<const float *>(a.x) + (long)plane * a.x_plane;
    float *opl = static_cast<float *>(a.out) + (long)plane * a.out_plane;
    const brsrc rin = plane_rsrc(ipl, a.in_plane), rx = plane_rsrc(xpl, a.x_plane), ro = plane_rsrc(opl, a.out_plane);
    const unsigned ipitchb = (unsigned)a.in_pitch * 4u, xpitchb = (unsigned)a.x_pitch * 4u, opitchb = (unsigned)a.out_pitch * 4u;
    const bool virt_in = a.in_kind == SRC_VIRTUAL, virt_x = a.x_kind == SRC_VIRTUAL;
    const int lo_in = virt_in ? a.pad : 0, xsh = virt_x ? a.pad : 0, oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const int xw = virt_x ? a.W : a.W + 2 * a.pad, xh = virt_x ? a.H : Hp;

    // ---- per-lane column offsets ----
    unsigned vin[INTERIOR ? 1 : 4], vx[INTERIOR ? 1 : 4], vout[INTERIOR ? 1 : 4];
    const bool out_lane = lane >= 3 && lane <= 60;             // lanes 3 .. 60 own the 232 outputs, four each
    const int ocol = ox0 + 4 * (lane - 3);                     // first of the lane's output columns (padded coordinates)
    if constexpr (INTERIOR) {
        vin[0] = (unsigned)(ox0 - ST_R + 4 * lane - lo_in) * 4u;
        vx[0] = out_lane ? (unsigned)(ocol - xsh) * 4u : kNoAccess;
        vout[0] = out_lane ? (unsigned)(ocol - oo) * 4u : kNoAccess;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // input sample f = 64 k + lane of the row; x operand / output sample of float index f: output column f - 12
            const int f = 64 * k + lane;
            const int ix = map_axis(ox0 - ST_R + f, a.W, a.in_kind, a.boundary, a.pad);
            vin[k] = ix >= 0 ? (unsigned)ix * 4u : kNoAccess;
            const int c = f - ST_R, pc = ox0 + c;
            const bool ok = c >= 0 && c < ST_W && pc < rg.x_hi;
            vx[k] = ok ? (unsigned)min(max(pc - xsh, 0), xw - 1) * 4u : kNoAccess;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pc = ocol + e;
            vout[e] = (out_lane && pc < rg.x_hi) ? (unsigned)(pc - oo) * 4u : kNoAccess;
        }
    }

    // ---- taps (scalar registers): TP[p] = (h[p], h[p-1]),  HY[m] = (hy[2m], hy[2m+1]),  h = marginal taps 0 .. 12 ----
    const PB_CONSTANT float *ckx = as_constant(info->kx), *cky = as_constant(info->ky);
    f2 TP[ST_R + 1], HY[(ST_R + 2) / 2];
#pragma unroll
    for (int t = 0; t <= ST_R; ++t) TP[t] = (f2){ckx[t], t ? ckx[t - 1] : 0.f};
#pragma unroll
    for (int m = 0; m < (ST_R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= ST_R ? cky[2 * m + 1] : 0.f};
    const float sc = a.scale, cfx = a.coef;
    const float clo = a.clamp01 ? 0.f : -INFINITY, chi = a.clamp01 ? 1.f : INFINITY;

    char *in_ring = zb, *x_ring = zb + ST_SLOTS * ST_IN * 4;
    // the lane's window in an input row: 16-byte pieces lane - 3 .. lane + 3 (lanes near the strip's edge read a clamped
    // piece: they produce no output)
    const unsigned lds_in = lds_addr(in_ring), lds_x = lds_addr(x_ring) + (unsigned)lane * 16u;
    unsigned wa[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) wa[q] = lds_in + (unsigned)min(max(lane - 3 + q, 0), 63) * 16u;

    // request input row j (and the x operand of the output row it completes, j - 24) into slot `slot` (= j % ST_SLOTS)
    const bool wrap = a.boundary == PB_WRAP;
    auto request = [&](int j, int slot) {
        // source row of padded row py under the boundary model (map_axis without its division: |py| < 2 Hp here)
        int py = oy0 - ST_R + j;
        if (wrap) { py = py < 0 ? py + Hp : py; py = py >= Hp ? py - Hp : py; }
        const bool rok = j < nrows && py >= 0 && py < Hp;
        const int iy = virt_in ? min(max(py - a.pad, 0), a.H - 1) : py;
        const int o = j - 2 * ST_R;                                            // output row completed by input row j
        const bool xok = o >= 0 && o < nout;
        const int xr = min(max(oy0 + o - xsh, 0), xh - 1);
        const int si = rok ? (int)((unsigned)iy * ipitchb) : 0, sxo = (int)((unsigned)xr * xpitchb);
        if constexpr (INTERIOR) {
            st_dma16(rin, in_ring + slot * (ST_IN * 4), rok ? vin[0] : kNoAccess, si);
            st_dma16(rx, x_ring + slot * (ST_IN * 4), xok ? vx[0] : kNoAccess, sxo);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) st_dma4(rin, in_ring + slot * (ST_IN * 4) + k * 256, rok ? vin[k] : kNoAccess, si);
#pragma unroll
            for (int k = 0; k < 4; ++k) st_dma4(rx, x_ring + slot * (ST_IN * 4) + k * 256, xok ? vx[k] : kNoAccess, sxo);
        }
    };
    // the store slot of a step (a step without an output row issues it with out-of-range offsets: constant counts)
    auto store = [&](int o, bool valid, sf4 v) {
        const int po = oy0 + o;
        const int so = valid ? (int)((unsigned)(po - oo) * opitchb) : 0;
        if constexpr (INTERIOR) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su4, v), ro, (int)((valid ? vout[0] : kNoAccess) + (unsigned)so), 0, 0);
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e[k]), ro, (int)((valid ? vout[k] : kNoAccess) + (unsigned)so), 0, 0);
        }
    };

#ifdef PB_ST_TRACE
    unsigned long long t_vm = 0, t_lds = 0, t_n = 0;
    const unsigned long long t_begin = PB_ST_CLK(), rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    f2 rxy[25], rzw[25];                                         // the ring of x-filtered rows
#pragma unroll
    for (int t = 0; t < 25; ++t) { rxy[t] = (f2){0.f, 0.f}; rzw[t] = (f2){0.f, 0.f}; }

    // prologue: the first ST_AHEAD rows, each behind a store slot as in the steady state
#pragma unroll 1
    for (int j = 0; j < ST_AHEAD; ++j) {
        store(0, false, (sf4){0.f, 0.f, 0.f, 0.f});
        asm volatile("" ::: "memory");
        request(j, j);
    }

    auto step = [&](auto uc, int i) {
        constexpr int U = decltype(uc)::value;                  // i % 25: the ring slot of this step's row
        // rows i (input) and i - 24 (x operand), requested ST_AHEAD steps ago, have landed
#ifdef PB_ST_TRACE
        const unsigned long long c0 = PB_ST_CLK();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((ST_AHEAD - 1) * OPS) : "memory");
#ifdef PB_ST_TRACE
        const unsigned long long c1 = PB_ST_CLK();
#endif
        constexpr int SO = (U % ST_SLOTS) * (ST_IN * 4);       // the slot's byte offset: an immediate of the reads
        sf4 w[7], x4;
        // (LDS reads behind the compiler's back: it would wait for every outstanding LDS-DMA before each of them)
        asm volatile("ds_read_b128 %0, %8 offset:%16\n\tds_read_b128 %1, %9 offset:%16\n\tds_read_b128 %2, %10 offset:%16\n\t"
                     "ds_read_b128 %3, %11 offset:%16\n\tds_read_b128 %4, %12 offset:%16\n\tds_read_b128 %5, %13 offset:%16\n\t"
                     "ds_read_b128 %6, %14 offset:%16\n\tds_read_b128 %7, %15 offset:%16\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(x4)
                     : "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(lds_x), "n"(SO)
                     : "memory");
#ifdef PB_ST_TRACE
        const unsigned long long c2 = PB_ST_CLK();
        t_vm += c1 - c0; t_lds += c2 - c1; ++t_n;
#endif
        // ---- x pass ----
        f2 d[ST_R + 2];
#pragma unroll
        for (int q = 0; q < 7; ++q) { d[2 * q] = (f2){w[q].x, w[q].y}; d[2 * q + 1] = (f2){w[q].z, w[q].w}; }
        f2 vxy[2] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}}, vzw[2] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}};
        XPass4<ST_R, 0>::run(vxy, vzw, TP, d);
        rxy[U] = vxy[0] + vxy[1]; rzw[U] = vzw[0] + vzw[1];
        // ---- y pass: output row i - 24 from the 25 rows i - 24 .. i (ring slots U + 1 .. U + 25 mod 25) ----
        // (the newest row comes last: 24 of the 25 taps do not wait for the x pass; two accumulators per output pair)
        f2 bxy[2] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}}, bzw[2] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int slot = (U + 1 + t) % 25, qq = t <= ST_R ? t : 2 * ST_R - t;
            if (qq & 1) { pk_bcast_tap<1>(bxy[t & 1], HY[qq >> 1], rxy[slot]); pk_bcast_tap<1>(bzw[t & 1], HY[qq >> 1], rzw[slot]); }
            else { pk_bcast_tap<0>(bxy[t & 1], HY[qq >> 1], rxy[slot]); pk_bcast_tap<0>(bzw[t & 1], HY[qq >> 1], rzw[slot]); }
        }
        const f2 axy = bxy[0] + bxy[1], azw = bzw[0] + bzw[1];
        sf4 o4;
        o4.x = fmaf(sc, axy.x, cfx * x4.x); o4.y = fmaf(sc, axy.y, cfx * x4.y);
        o4.z = fmaf(sc, azw.x, cfx * x4.z); o4.w = fmaf(sc, azw.y, cfx * x4.w);
        o4.x = fminf(fmaxf(o4.x, clo), chi); o4.y = fminf(fmaxf(o4.y, clo), chi);      // (branch-free: infinite bounds without clamp)
        o4.z = fminf(fmaxf(o4.z, clo), chi); o4.w = fminf(fmaxf(o4.w, clo), chi);
        const int o = i - 2 * ST_R;
        store(o, o >= 0 && o < nout, o4);
        asm volatile("" ::: "memory");
        request(i + ST_AHEAD, (U + ST_AHEAD) % ST_SLOTS);
    };
#define PB_STEP(u) if (base + u < nrows) step(std::integral_constant<int, u>{}, base + u);
#pragma unroll 1
    for (int base = 0; base < nrows; base += 25) {
        PB_STEP(0) PB_STEP(1) PB_STEP(2) PB_STEP(3) PB_STEP(4) PB_STEP(5) PB_STEP(6) PB_STEP(7) PB_STEP(8) PB_STEP(9)
        PB_STEP(10) PB_STEP(11) PB_STEP(12) PB_STEP(13) PB_STEP(14) PB_STEP(15) PB_STEP(16) PB_STEP(17) PB_STEP(18) PB_STEP(19)
        PB_STEP(20) PB_STEP(21) PB_STEP(22) PB_STEP(23) PB_STEP(24)
    }
#undef PB_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (no LDS-DMA may land after the wave has given its LDS back)
#ifdef PB_ST_TRACE
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long *tr = g_st_trace + (long)blockIdx.x * 6;
        tr[0] = t_vm; tr[1] = t_lds; tr[2] = PB_ST_CLK() - t_begin; tr[3] = t_n; tr[4] = rt_begin; tr[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// whether a strip takes the 16-byte path: every input, x-operand and output column it touches lies inside its plane
// without boundary mapping or clamping, on 16-byte boundaries, and the strip is complete
__device__ __forceinline__ bool strip_is_interior(const ConvPass &a, int sx) {
    const OutRegion rg = out_region(a);
    const int Wp = a.W + 2 * a.pad;
    const int ox0 = rg.x_lo + sx * ST_W;
    const int lo_in = a.in_kind == SRC_VIRTUAL ? a.pad : 0, xsh = a.x_kind == SRC_VIRTUAL ? a.pad : 0;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const int xw = a.x_kind == SRC_VIRTUAL ? a.W : Wp;
    return ox0 - ST_R >= lo_in && ox0 - ST_R + ST_IN <= Wp - lo_in && ox0 + ST_W <= rg.x_hi && ox0 - xsh >= 0 && ox0 + ST_W - xsh <= xw &&
           ((a.in_pitch | a.x_pitch | a.out_pitch | (ox0 - ST_R - lo_in) | (ox0 - xsh) | (ox0 - oo)) & 3) == 0;
}

// One wave (= one workgroup) per (plane, segment, strip); workgroup b belongs to list b % 8 (the XCD it is observed to run
// on, used for speed only) at position b / 8, and every list owns the same contiguous eighth of every plane's strips.
// Planes of images whose kernel is not rank-1 with full support leave at once (their records select another body).
__global__ __launch_bounds__(64, 3) void conv_strip_kernel(const ConvPass a, const StripGeom g) {
    extern __shared__ __attribute__((aligned(16))) char zb[];
    const int q = (int)(blockIdx.x & 7u), rem = (int)(blockIdx.x >> 3);
    const int plane = __builtin_amdgcn_readfirstlane(div_small(rem, g.inv_per));
    if (plane >= a.P) return;
    const int job = q * g.per + (rem - plane * g.per);
    if (job >= g.strips_x * g.segs_y) return;
    const int img = plane / a.C;
    const PB_CONSTANT pb_blur_info *ci = as_constant(a.info + img);
    if (!ci->separable || ci->radius <= 8) return;
    const int sy = __builtin_amdgcn_readfirstlane(div_small(job, g.inv_strips)), sx = job - sy * g.strips_x;
    if (strip_is_interior(a, sx)) strip_body<true>(a, a.info + img, plane, sx, sy, g, zb);
    else strip_body<false>(a, a.info + img, plane, sx, sy, g, zb);
}

}  // namespace

// PB_ERR_UNSUPPORTED: not an all-fp32 plain Horner pass (the caller keeps the tile body).
int pb_launch_conv_strip(pb_ctx *ctx, const ConvPass &p) {
    if (p.in_dtype != PB_F32 || p.x_dtype != PB_F32 || p.out_dtype != PB_F32 || p.epilogue != EPI_HORNER || p.pad != PB_KRAD)
        return PB_ERR_UNSUPPORTED;
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    StripGeom g;
    g.strips_x = (ow + ST_W - 1) / ST_W;
    // segment height: about two waves per SIMD over the whole batch, but at least 64 rows (the 24-row halo is paid per segment)
    long target = (2048 + (long)g.strips_x * p.P - 1) / ((long)g.strips_x * p.P);
    if (target < 1) target = 1;
    int seg_h = (int)((oh + target - 1) / target);
    const int forced = ctx->strip_seg;
    if (forced > 0) seg_h = forced;
    if (seg_h < 64) seg_h = 64;
    if (seg_h > oh) seg_h = oh;
    g.seg_h = seg_h;
    g.segs_y = (oh + seg_h - 1) / seg_h;
    const long jobs = (long)g.strips_x * g.segs_y;
    g.per = (int)((jobs + 7) / 8);
    g.inv_strips = 1.0f / (float)g.strips_x;
    g.inv_per = 1.0f / (float)g.per;
    const long groups = 8L * g.per * p.P;
    if (groups <= 0 || groups > (1L << 22)) return PB_ERR_UNSUPPORTED;
    const long plane_max = (1L << 31) - 4096;
    if (p.in_plane * 4 >= plane_max || p.x_plane * 4 >= plane_max || p.out_plane * 4 >= plane_max) return PB_ERR_UNSUPPORTED;
    ProfScope prof(ctx, PB_PROF_CONV);
    hipLaunchKernelGGL(conv_strip_kernel, dim3((unsigned)groups), dim3(64), kStripLds, ctx->stream, p, g);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

#ifdef PB_ST_TRACE
extern "C" int pb_debug_st_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_st_trace), sizeof(unsigned long long) * 4096 * 6);
}
#endif
