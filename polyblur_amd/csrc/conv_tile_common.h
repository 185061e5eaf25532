// Tile-geometry pieces of the stencil kernels (conv.hip).
#pragma once
#include "common.h"
#include "conv_common.h"

namespace {

constexpr int NT = 256;
constexpr int GT = 64;   // 64 x 64 outputs per workgroup tile, 4 x 4 per thread
#ifndef PB_TILE_DMA
#define PB_TILE_DMA 1
#endif

// What a workgroup needs to know about one tile.
struct TileJob {
    int plane, ty, tx;
    const pb_blur_info *info;
    int cls;                       // 8 * separable + support class index (0: R=4, 1: R=8, 2: R=12)
};
__device__ __forceinline__ TileJob decode_tile(const ConvPass &a, int tile_id, int tiles_per_plane, int tiles_x, int) {
    TileJob j;
    j.plane = tile_id / tiles_per_plane;
    const int local = tile_id - j.plane * tiles_per_plane;
    j.ty = local / tiles_x;
    j.tx = local - j.ty * tiles_x;
    j.info = a.info + j.plane / a.C;
    const PB_CONSTANT pb_blur_info *ci = as_constant(j.info);
    const int sep = ci->separable != 0;
    const int R = ci->radius;
    j.cls = 8 * sep + (R <= 4 ? 0 : (R <= 8 ? 1 : 2));
    return j;
}

// Four consecutive padded columns px..px+3 of source row iy (iy < 0: the row reads as zero): one 16-byte load
// where the columns map to themselves and are aligned, four mapped loads in the pad / wrap region.
template <typename T>
__device__ __forceinline__ float4 load_chunk_mapped(const T *plane, int pitch, int iy, int px, int W, int kind, int boundary,
                                                    bool aligned, int pad) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy < 0) return v;
    const T *row = plane + (long)iy * pitch;
    const int shift = (kind == SRC_VIRTUAL) ? pad : 0;
    const int lo = shift, hi = (kind == SRC_VIRTUAL) ? pad + W : W + 2 * pad;
    if (aligned && px >= lo && px + 3 < hi) return ld4<T>(row + (px - shift));
    const int i0 = map_axis(px, W, kind, boundary, pad), i1 = map_axis(px + 1, W, kind, boundary, pad);
    const int i2 = map_axis(px + 2, W, kind, boundary, pad), i3 = map_axis(px + 3, W, kind, boundary, pad);
    if (i0 >= 0) v.x = pb_ld(row + i0);
    if (i1 >= 0) v.y = pb_ld(row + i1);
    if (i2 >= 0) v.z = pb_ld(row + i2);
    if (i3 >= 0) v.w = pb_ld(row + i3);
    return v;
}

template <typename T, int LH, int LW, int LP>
__device__ __forceinline__ void load_tile(float *s, const T *plane, int kind, int pitch, int H, int W, int py0, int px0,
                                          int boundary, int pad) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + LH <= Hp && px0 + LW <= Wp;
    int sy0 = py0, sx0 = px0;
    if (kind == SRC_VIRTUAL) {
        inside = inside && py0 >= pad && px0 >= pad && py0 + LH <= pad + H && px0 + LW <= pad + W;
        sy0 -= pad; sx0 -= pad;
    }
    const int tid = threadIdx.x;
    if (inside && ((pitch | sx0) & 3) == 0) {
        // interior tile: every 16-byte load of the tile is issued before the first one is consumed
        const T *base = plane + (long)sy0 * pitch + sx0;
        constexpr int C4 = LW / 4;
        constexpr int NLD = (LH * C4 + NT - 1) / NT;
        if constexpr (PB_TILE_DMA && __is_same(T, float) && LP == LW) {
            // fp32 interior tile: global -> LDS directly, 1 KiB of the row-major tile per wave instruction (see load_rows_wave)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            typedef __attribute__((address_space(3))) char lds_char;
            lds_char *dst = (lds_char *)s + (tid >> 6) * 1024;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int e = tid + k * NT;
                const int r = e / C4, c = e - r * C4;
                if (e < LH * C4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (long)r * pitch + 4 * c),
                                                     (__attribute__((address_space(3))) void *)(dst + k * (NT * 16)), 16, 0, 0);
            }
#pragma clang diagnostic pop
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        float4 buf[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int r = e / C4, c = e - r * C4;
            if (e < LH * C4) buf[k] = ld4<T>(base + (long)r * pitch + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int r = e / C4, c = e - r * C4;
            if (e < LH * C4) *reinterpret_cast<float4 *>(s + r * LP + 4 * c) = buf[k];
        }
    } else {
        // border tile: rows mapped (wrap / zero / clamp) once per chunk, columns per chunk or per sample
        const bool aligned = ((pitch | sx0) & 3) == 0;
        constexpr int C4 = LW / 4;
        constexpr int NLD = (LH * C4 + NT - 1) / NT;
        float4 buf[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int r = e / C4, c = e - r * C4;
            if (e < LH * C4)
                buf[k] = load_chunk_mapped<T>(plane, pitch, map_axis(py0 + r, H, kind, boundary, pad), px0 + 4 * c, W, kind, boundary, aligned, pad);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int r = e / C4, c = e - r * C4;
            if (e < LH * C4) *reinterpret_cast<float4 *>(s + r * LP + 4 * c) = buf[k];
        }
    }
}

// Wave-private variant for the rank-1 body: wave w stages rows [w*RPW, (w+1)*RPW) of the tile -- exactly
// the rows it x-filters -- so no workgroup barrier is needed between the load and the x pass and the four
// waves of a workgroup drift apart (one's loads overlap another's arithmetic).
template <typename T, int LH, int LW, int LP, int RPW>
__device__ __forceinline__ void load_rows_wave(float *s, const T *plane, int kind, int pitch, int H, int W, int py0, int px0,
                                               int boundary, int pad, int tid = threadIdx.x) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + LH <= Hp && px0 + LW <= Wp;
    int sy0 = py0, sx0 = px0;
    if (kind == SRC_VIRTUAL) {
        inside = inside && py0 >= pad && px0 >= pad && py0 + LH <= pad + H && px0 + LW <= pad + W;
        sy0 -= pad; sx0 -= pad;
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = wave * RPW;
    const int nrows = min(RPW, LH - r0);
    constexpr int C4 = LW / 4;
    if (inside && ((pitch | sx0) & 3) == 0) {
        const T *base = plane + (long)(sy0 + r0) * pitch + sx0;
        constexpr int NLD = (RPW * C4 + 63) / 64;
        if constexpr (PB_TILE_DMA && __is_same(T, float) && LP == LW) {
            // fp32 interior tile: global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, lane l
            // lands at the wave-uniform base + 16 l, which with unpadded rows IS the row-major tile), no staging
            // registers, no ds_write pass.  The wave waits for its own rows only.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            typedef __attribute__((address_space(3))) char lds_char;
            lds_char *dst = (lds_char *)(s + r0 * LP);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int e = lane + k * 64;
                const int r = e / C4, c = e - r * C4;
                if (r < nrows)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (long)r * pitch + 4 * c),
                                                     (__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
            }
#pragma clang diagnostic pop
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        float4 buf[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows) buf[k] = ld4<T>(base + (long)r * pitch + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows) *reinterpret_cast<float4 *>(s + (r0 + r) * LP + 4 * c) = buf[k];
        }
    } else {
        const bool aligned = ((pitch | sx0) & 3) == 0;
        constexpr int NLD = (RPW * C4 + 63) / 64;
        float4 buf[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows)
                buf[k] = load_chunk_mapped<T>(plane, pitch, map_axis(py0 + r0 + r, H, kind, boundary, pad), px0 + 4 * c, W, kind,
                                              boundary, aligned, pad);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = lane + k * 64;
            const int r = e / C4, c = e - r * C4;
            if (r < nrows) *reinterpret_cast<float4 *>(s + (r0 + r) * LP + 4 * c) = buf[k];
        }
    }
}

// Per-thread epilogue of a 4 x 4 output block.  `prefetch` is called right after the tile loads have
// been issued so that the x operand arrives while the stencil is evaluated; blocks that touch the
// border of the output region (or a clamped / unaligned x operand, or the taper blend) take finish4.
template <typename TX, typename TOut> struct Block4x4Epilogue {
    bool fast;
    float4 xr[4];
    const TX *xp;
    TOut *op;
    __device__ __forceinline__ void prefetch(const ConvPass &a, const TX *xpl, TOut *opl, const OutRegion &rg, int py, int px) {
        const int xo = a.x_kind == SRC_VIRTUAL ? a.pad : 0, oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
        const int xrows = a.x_kind == SRC_VIRTUAL ? a.H : a.H + 2 * a.pad, xcols = a.x_kind == SRC_VIRTUAL ? a.W : a.W + 2 * a.pad;
        fast = a.epilogue == EPI_HORNER && py + 3 < rg.y_hi && px >= rg.x_lo && px + 3 < rg.x_hi && py - xo >= 0 && py - xo + 3 < xrows &&
               px - xo >= 0 && px - xo + 3 < xcols && ((a.x_pitch | a.out_pitch) & 3) == 0;
        if (fast) {
            xp = xpl + (long)(py - xo) * a.x_pitch + (px - xo);
            op = opl + (long)(py - oo) * a.out_pitch + (px - oo);
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = ld4<TX>(xp + (long)r * a.x_pitch);
        } else {
            // border blocks: the (clamped) x operand is fetched up front all the same, one row at a time
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xr[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (py + r >= rg.y_lo && py + r < rg.y_hi && px < rg.x_hi && px + 3 >= rg.x_lo) xr[r] = load_x4<TX>(a, xpl, py + r, px);
            }
        }
    }
    __device__ __forceinline__ void finish(const ConvPass &a, const pb_blur_info *info, const TX *xpl, TOut *opl,
                                           const OutRegion &rg, int py, int px, const float4 (&acc)[4]) {
        if (fast) {
            const float sc = a.scale, cf = a.coef;
            const bool cl = a.clamp01 != 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 v;
                v.x = fmaf(sc, acc[r].x, cf * xr[r].x); v.y = fmaf(sc, acc[r].y, cf * xr[r].y);
                v.z = fmaf(sc, acc[r].z, cf * xr[r].z); v.w = fmaf(sc, acc[r].w, cf * xr[r].w);
                if (cl) {
                    v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                    v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                }
                st4<TOut>(op + (long)r * a.out_pitch, v);
            }
        } else {
            // (statically indexed: a run-time index would push acc[] -- and a 64-byte store per thread -- to scratch)
            finish4<TOut>(a, info, opl, rg, py, px, acc[0], xr[0]);
            finish4<TOut>(a, info, opl, rg, py + 1, px, acc[1], xr[1]);
            finish4<TOut>(a, info, opl, rg, py + 2, px, acc[2], xr[2]);
            finish4<TOut>(a, info, opl, rg, py + 3, px, acc[3], xr[3]);
        }
    }
};

template <int R, int P> __device__ __forceinline__ void xtap(f2 &acc, const f2 (&TP)[R + 1], f2 dpair, int) {}
template <int R, int P, int HI> struct XTapApply {
    static __device__ __forceinline__ void run(f2 &acc, const f2 (&TP)[R + 1], f2 dpair) {
        if constexpr (P <= R) pk_bcast_data<0, HI>(acc, TP[P], dpair);
        else pk_bcast_data<1, HI>(acc, TP[2 * R + 1 - P], dpair);
    }
};
// window element J (0 .. 2R+3) feeds (x,y) with the tap pair T[J] and (z,w) with T[J-2]
template <int R, int J> struct XPassR {
    static __device__ __forceinline__ void run(f2 &vxy, f2 &vzw, const f2 (&TP)[R + 1], const f2 (&d)[R + 2]) {
        if constexpr (J <= 2 * R + 1) XTapApply<R, J, J & 1>::run(vxy, TP, d[J >> 1]);
        if constexpr (J >= 2) XTapApply<R, J - 2, J & 1>::run(vzw, TP, d[J >> 1]);
        if constexpr (J < 2 * R + 3) XPassR<R, J + 1>::run(vxy, vzw, TP, d);
    }
};
// input row I (0 .. 2R+3) of the thread's window feeds output row r with tap I - r.  The rows travel through a
// ring of four registers quads, read three steps ahead of their use (the scheduler barrier keeps each ds_read in
// front of the step's arithmetic instead of sinking it to its first use and exposing the LDS latency every step).
template <int R, int I> struct YPassR {
    static __device__ __forceinline__ void step(f2 (&axy)[4], f2 (&azw)[4], const f2 (&HY)[(R + 2) / 2], const float *col,
                                                int pitch, float4 (&ring)[4]) {
        if constexpr (I + 3 <= 2 * R + 3) ring[(I + 3) & 3] = *reinterpret_cast<const float4 *>(col + (I + 3) * pitch);
        __builtin_amdgcn_sched_barrier(0);
        const float4 v4 = ring[I & 3];
        const f2 vxy = (f2){v4.x, v4.y}, vzw = (f2){v4.z, v4.w};
#define PB_YROW(RR)                                                                   \
        if constexpr (I - RR >= 0 && I - RR <= 2 * R) {                               \
            constexpr int t = I - RR, q = t <= R ? t : 2 * R - t;                     \
            pk_bcast_tap<q & 1>(axy[RR], HY[q >> 1], vxy);                            \
            pk_bcast_tap<q & 1>(azw[RR], HY[q >> 1], vzw);                            \
        }
        PB_YROW(0) PB_YROW(1) PB_YROW(2) PB_YROW(3)
#undef PB_YROW
        if constexpr (I < 2 * R + 3) YPassR<R, I + 1>::step(axy, azw, HY, col, pitch, ring);
    }
    static __device__ __forceinline__ void run(f2 (&axy)[4], f2 (&azw)[4], const f2 (&HY)[(R + 2) / 2], const float *col,
                                               int pitch) {
        static_assert(I == 0, "start at row 0");
        float4 ring[4];
        ring[0] = *reinterpret_cast<const float4 *>(col);
        ring[1] = *reinterpret_cast<const float4 *>(col + pitch);
        ring[2] = *reinterpret_cast<const float4 *>(col + 2 * pitch);
        step(axy, azw, HY, col, pitch, ring);
    }
};


}  // namespace
