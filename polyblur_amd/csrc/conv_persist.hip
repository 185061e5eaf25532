// Stencil pass, rank-1 kernels: persistent workgroups that keep the NEXT tile's loads in flight in registers
// while the current tile is filtered out of LDS.
//
// A one-shot tile workgroup (conv.hip) holds its 31 KB of LDS from the moment it starts until its last store,
// but the LDS is idle while the tile's loads travel; five such workgroups per CU cannot keep more than about
// one tile per CU in flight (measured: 3.6 TB/s of algorithmic bytes).  Here a workgroup stays resident and
// walks a list of tiles.  Each wave stages the rows it x-filters itself: at the top of an iteration it writes the
// 8 float4 it prefetched during the previous iteration into LDS, immediately issues the loads of the following
// tile into the same registers, and only then filters.  Geometry and arithmetic are conv.hip's in-LDS body
// (64x64 outputs per tile, in-place x pass, y pass into 4x4 register blocks, packed FMAs).
#include <cstdlib>

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

// LDS visibility + workgroup barrier WITHOUT draining outstanding vector-memory operations (the prefetch must
// stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int R> struct PGeom {
    static constexpr int LW = GT + 2 * R, LH = GT + 2 * R, LP = LW, C4 = LW / 4;
    static constexpr int RPW = (LH + 3) / 4;                       // rows staged and x-filtered by each wave
    static constexpr int NLD = (RPW * C4 + 63) / 64;               // float4 per lane
    static constexpr int XROT = (16 - ((LP / 4) % 16)) % 16, YROT = (16 - (LP % 16)) % 16;
};

// issue this wave's share of a simple tile's loads: rows [wave*RPW, ...) of the (64+2R)^2 window whose first
// sample is at `base` (source coordinates)
template <typename TIn, int R>
__device__ __forceinline__ void prefetch_rows(float4 (&buf)[PGeom<R>::NLD], const TIn *base, int pitch) {
    using G = PGeom<R>;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));            // recompute the lane's offsets every tile instead of keeping 16 registers of them alive
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = wave * G::RPW;
    const int nrows = min(G::RPW, G::LH - r0);
    // every load is issued unconditionally (lanes past the wave's share repeat its last chunk): the compiler's
    // s_waitcnt bookkeeping can then let all NLD of them stay in flight across the x-operand wait
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        const int e = min(lane + k * 64, nrows * G::C4 - 1);
        const int r = e / G::C4, c = e - r * G::C4;
        buf[k] = ld4<TIn>(base + (unsigned)((r0 + r) * pitch + 4 * c));
    }
}

template <int R>
__device__ __forceinline__ void stage_rows(float *s, const float4 (&buf)[PGeom<R>::NLD]) {
    using G = PGeom<R>;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = wave * G::RPW;
    const int nrows = min(G::RPW, G::LH - r0);
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        const int e = lane + k * 64;
        const int r = e / G::C4, c = e - r * G::C4;
        if (r < nrows) *reinterpret_cast<float4 *>(s + (r0 + r) * G::LP + 4 * c) = buf[k];
    }
}

// What the loop needs of a pass, in scalars (the ConvPass itself stays in the kernel-argument segment).
struct PersistArgs {
    const void *in, *x;
    void *out;
    const pb_blur_info *info;
    long in_plane, x_plane, out_plane;
    int in_pitch, x_pitch, out_pitch;
    int in_off, x_off, out_off;          // padded coordinate -> source / destination coordinate
    int y_lo, x_lo;                       // first output sample (padded coordinates)
    int C;
    float scale, coef;
    int clamp01;
};

// next simple rank-1 tile of class R at or after t (stride apart), or t_end
template <int R>
__device__ __forceinline__ int next_simple(const ConvPass &a, int t, int t_end, int stride, int tiles_per_plane, int tiles_x,
                                           bool &other_class) {
    other_class = false;
    for (; t < t_end; t += stride) {
        const int plane = t / tiles_per_plane;
        const PB_CONSTANT pb_blur_info *ci = as_constant(a.info + plane / a.C);
        if (ci->separable == 0) continue;
        const int Rt = ci->radius <= 4 ? 4 : (ci->radius <= 8 ? 8 : PB_KRAD);
        if (Rt != R) { other_class = true; return t; }
        const int local = t - plane * tiles_per_plane;
        const int ty = local / tiles_x, tx = local - ty * tiles_x;
        if (pb_tile_is_simple(a, R, ty, tx)) return t;
    }
    return t_end;
}

template <typename TIn, typename TX, typename TOut, int R>
__device__ __forceinline__ int run_tiles(const ConvPass &a, const PersistArgs &q, float *smem, int t, int t_end, int stride,
                                         int tiles_per_plane, int tiles_x) {
    using G = PGeom<R>;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rgp = tid >> 4, gy = ((tid & 15) + G::YROT * (rgp & 1)) & 15;
    float4 buf[G::NLD];
    bool other = false;
    t = next_simple<R>(a, t, t_end, stride, tiles_per_plane, tiles_x, other);
    if (t >= t_end || other) return t;
    int plane = t / tiles_per_plane;
    int local = t - plane * tiles_per_plane;
    int ty = local / tiles_x, tx = local - ty * tiles_x;
    prefetch_rows<TIn, R>(buf, static_cast<const TIn *>(q.in) + plane * q.in_plane +
                                   (long)(q.y_lo + ty * GT - R - q.in_off) * q.in_pitch + (q.x_lo + tx * GT - R - q.in_off),
                          q.in_pitch);
    while (true) {
        const int oy0 = q.y_lo + ty * GT, ox0 = q.x_lo + tx * GT;
        const pb_blur_info *info = q.info + plane / q.C;
        const TX *xp = static_cast<const TX *>(q.x) + plane * q.x_plane + (long)(oy0 + rgp * 4 - q.x_off) * q.x_pitch +
                       (ox0 + 4 * gy - q.x_off);
        TOut *op = static_cast<TOut *>(q.out) + plane * q.out_plane + (long)(oy0 + rgp * 4 - q.out_off) * q.out_pitch +
                   (ox0 + 4 * gy - q.out_off);
        // ---- stage this wave's rows (the previous tile has been fully consumed: barrier at the loop's end) ----
        stage_rows<R>(smem, buf);
        // ---- this tile's x operand, then the next tile's loads (they stay in flight through the arithmetic) ----
        float4 xr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = ld4<TX>(xp + (long)r * q.x_pitch);
        bool nother = false;
        const int tn = next_simple<R>(a, t + stride, t_end, stride, tiles_per_plane, tiles_x, nother);
        const bool more = tn < t_end && !nother;
        int nplane = plane, nty = ty, ntx = tx;
        if (more) {
            nplane = tn / tiles_per_plane;
            const int nl = tn - nplane * tiles_per_plane;
            nty = nl / tiles_x; ntx = nl - nty * tiles_x;
        }
        // issued on every path (after the last tile it re-reads the current one, which is in cache): only then
        // does the compiler's s_waitcnt for the x operand leave all of these loads outstanding
        prefetch_rows<TIn, R>(buf, static_cast<const TIn *>(q.in) + nplane * q.in_plane +
                                       (long)(q.y_lo + nty * GT - R - q.in_off) * q.in_pitch + (q.x_lo + ntx * GT - R - q.in_off),
                              q.in_pitch);
        const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
        {
            f2 TP[R + 1];
#pragma unroll
            for (int p = 0; p <= R; ++p) TP[p] = (f2){ckx[p], p ? ckx[p - 1] : 0.f};
            wave_lds_fence();                               // a wave reads back only rows it staged itself
            // ---- x pass, in place ----
            const int rsub = lane >> 4, g = ((lane & 15) + G::XROT * (rsub & 1)) & 15;
            for (int it = 0; it < (G::RPW + 3) / 4; ++it) {
                const int rr = wave * G::RPW + it * 4 + rsub;
                const bool ok = (it * 4 + rsub) < G::RPW && rr < G::LH;
                float *row = smem + (ok ? rr : 0) * G::LP;
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
        lds_barrier();
        // ---- y pass: 4 x 4 outputs per thread, Horner epilogue, store ----
        {
            f2 HY[(R + 2) / 2];
#pragma unroll
            for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
            f2 axy[4], azw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
            YPassR<R, 0>::run(axy, azw, HY, smem + (rgp * 4) * G::LP + 4 * gy, G::LP);
            const float sc = q.scale, cf = q.coef;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 v;
                v.x = fmaf(sc, axy[r].x, cf * xr[r].x); v.y = fmaf(sc, axy[r].y, cf * xr[r].y);
                v.z = fmaf(sc, azw[r].x, cf * xr[r].z); v.w = fmaf(sc, azw[r].y, cf * xr[r].w);
                if (q.clamp01) {
                    v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                    v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                }
                st4<TOut>(op + (long)r * q.out_pitch, v);
            }
        }
        lds_barrier();                                      // everybody has finished reading the tile
        t = tn;
        if (!more) break;
        plane = nplane; ty = nty; tx = ntx;
    }
    return t;
}

constexpr size_t kPersistLds = sizeof(float) * PGeom<PB_KRAD>::LH * PGeom<PB_KRAD>::LP;      // 30 976 B

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(NT, 4) void conv_persist_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const OutRegion rg = out_region(a);
    PersistArgs q;
    q.in = a.in; q.x = a.x; q.out = a.out; q.info = a.info;
    q.in_plane = a.in_plane; q.x_plane = a.x_plane; q.out_plane = a.out_plane;
    q.in_pitch = a.in_pitch; q.x_pitch = a.x_pitch; q.out_pitch = a.out_pitch;
    q.in_off = a.in_kind == SRC_VIRTUAL ? PB_PAD : 0;
    q.x_off = a.x_kind == SRC_VIRTUAL ? PB_PAD : 0;
    q.out_off = a.out_kind == OUT_INTERIOR ? PB_PAD : 0;
    q.y_lo = rg.y_lo; q.x_lo = rg.x_lo; q.C = a.C; q.scale = a.scale; q.coef = a.coef; q.clamp01 = a.clamp01;
    // XCD k (= blockIdx % 8, speed only) walks its own contiguous range of tiles, its workgroups interleaved
    const int nx = gridDim.x >> 3;
    const int chunk = (total_tiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int t = xcd * chunk + (blockIdx.x >> 3);
    const int t_end = min(total_tiles, (xcd + 1) * chunk);
    while (t < t_end) {
        const int plane = t / tiles_per_plane;
        const PB_CONSTANT pb_blur_info *ci = as_constant(a.info + plane / a.C);
        if (ci->separable == 0) { t += nx; continue; }       // general taps: conv_tile_kernel does this image
        const int R = ci->radius;
        if (R <= 4) t = run_tiles<TIn, TX, TOut, 4>(a, q, smem, t, t_end, nx, tiles_per_plane, tiles_x);
        else if (R <= 8) t = run_tiles<TIn, TX, TOut, 8>(a, q, smem, t, t_end, nx, tiles_per_plane, tiles_x);
        else t = run_tiles<TIn, TX, TOut, 12>(a, q, smem, t, t_end, nx, tiles_per_plane, tiles_x);
    }
}

template <typename TIn, typename TX, typename TOut>
int launch_persist_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    const long tpp = (long)tiles_x * tiles_y;
    const long total = tpp * p.P;
    if (total <= 0 || total > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    static long resident = 0;
    if (!resident) {
        const char *e = getenv("PB_PERSIST_WGS");
        resident = e ? atol(e) : 1024;                       // 256 CUs x 4 workgroups (109 VGPRs, 31 KB LDS each)
        if (resident < 8) resident = 1024;
    }
    long grid = total < resident ? total : resident;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_persist_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(NT), kPersistLds, ctx->stream, p,
                       (int)tpp, tiles_x, (int)total);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// rank-1 images of float / half passes; other images are skipped on the device
int pb_launch_conv_persist(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    switch (key) {
        case 0: return launch_persist_typed<float, float, float>(ctx, p);
        case 1: return launch_persist_typed<float, float, __half>(ctx, p);
        case 3: return launch_persist_typed<float, __half, float>(ctx, p);
        case 4: return launch_persist_typed<float, __half, __half>(ctx, p);
        case 12: return launch_persist_typed<__half, __half, float>(ctx, p);
        case 13: return launch_persist_typed<__half, __half, __half>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "persistent stencil: unsupported dtype combination %d", key);
    }
}
