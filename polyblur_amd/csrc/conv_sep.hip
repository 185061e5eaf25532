// Stencil pass, rank-1 kernels, fp32 input: persistent workgroups with double-buffered LDS tiles filled
// by direct global->LDS loads (global_load_lds_dwordx4, no staging registers).
//
// The arithmetic of a rank-1 Horner step is cheap (30 packed FMAs per sample); what bounds the pass is
// keeping enough bytes in flight.  A one-shot tile workgroup loads, waits, computes, stores, and five of
// them per CU do not keep HBM busy (measured: 71 us per 4K launch with the arithmetic removed, against
// 44 us for a plain 2-read/1-write stream).  Here every workgroup is resident for the whole launch and
// walks its list of tiles; while tile n is filtered out of LDS buffer n&1, the DMA engine fills buffer
// (n+1)&1 with tile n+1, so a tile's worth of loads (31 KB) per workgroup is always outstanding.
//
// Geometry and arithmetic are those of conv.hip's in-LDS body: 64x64 outputs per tile, (64+2R)^2 staged
// samples in un-padded rows (the DMA writes LDS linearly), x pass in place, y pass into 4x4 register
// blocks, packed FMAs with symmetric marginal taps in SGPR pairs.  Tiles that touch the image border
// (wrap / zero / replicate clamp) are loaded sample by sample, synchronously.
#include <cstdlib>

#include "common.h"
#include "conv_common.h"
#include "conv_tile_common.h"

namespace {

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// LDS visibility + workgroup barrier WITHOUT draining outstanding vector-memory operations (the next
// tile's DMA must stay in flight across the barrier); "memory" keeps the compiler from moving LDS
// accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// vmcnt(0) through the builtin, so that hipcc's own bookkeeping also learns that nothing is outstanding
// (otherwise it re-waits -- and drains the freshly issued DMA -- before it reuses a store's data registers)
__device__ __forceinline__ void dma_wait_all() {
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
}

template <int R> struct SepGeom {
    static constexpr int LW = GT + 2 * R, LH = GT + 2 * R, LP = LW;
    static constexpr int C4 = LW / 4, NV4 = LH * C4;             // float4 per tile
    static constexpr int XROT = (16 - ((LP / 4) % 16)) % 16, YROT = (16 - (LP % 16)) % 16;
    static constexpr int BUF = LH * LP;                           // floats per buffer
};

// is the (LH x LW) tile whose first staged sample sits at padded (py0, px0) free of border handling?
template <int R>
__device__ __forceinline__ bool tile_interior(const ConvPass &a, int py0, int px0) {
    using G = SepGeom<R>;
    const int H = a.H, W = a.W, Hp = H + 2 * PB_PAD, Wp = W + 2 * PB_PAD;
    bool inside = py0 >= 0 && px0 >= 0 && py0 + G::LH <= Hp && px0 + G::LW <= Wp;
    int sx0 = px0;
    if (a.in_kind == SRC_VIRTUAL) {
        inside = inside && py0 >= PB_PAD && px0 >= PB_PAD && py0 + G::LH <= PB_PAD + H && px0 + G::LW <= PB_PAD + W;
        sx0 -= PB_PAD;
    }
    return inside && ((a.in_pitch | sx0) & 3) == 0;
}

// One LDS-DMA instruction: each lane moves 16 bytes from its own global address to
// (wave-uniform LDS byte address in M0) + lane * 16.  Issued through inline asm on purpose: hipcc drains
// every LDS-DMA it knows about (s_waitcnt vmcnt(0)) before the next ds_read, which would serialise the
// prefetch with the arithmetic; an asm statement is invisible to that bookkeeping, so the wait is ours
// (dma_wait_all + lds_barrier before the buffer is read).  M0 is saved and restored in the same
// statement (compiler-reserved register); recipe from the CDNA HIP guide, section 5.7.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_bytes)
                 : "memory");
}

// start the DMA of an interior tile into the buffer at LDS byte offset `buf_bytes` of the dynamic LDS block
template <int R>
__device__ __forceinline__ void tile_dma(const ConvPass &a, const float *plane, int py0, int px0, unsigned buf_bytes) {
    using G = SepGeom<R>;
    const int off = a.in_kind == SRC_VIRTUAL ? PB_PAD : 0;
    const float *base = plane + (long)(py0 - off) * a.in_pitch + (px0 - off);
    const int tid = threadIdx.x;
    const unsigned wave_base = buf_bytes + 1024u * (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int k = 0; k < (G::NV4 + NT - 1) / NT; ++k) {
        const int e = k * NT + tid;
        if (e < G::NV4) {
            const int r = e / G::C4, c = e - r * G::C4;
            glds16(base + (long)r * a.in_pitch + 4 * c, wave_base + 16u * (unsigned)(k * NT));
        }
    }
}

// border tile: every sample mapped (wrap / zero / clamp), 8 independent loads in flight per thread
template <int R>
__device__ __forceinline__ void tile_load_mapped(const ConvPass &a, const float *plane, int py0, int px0, float *buf) {
    using G = SepGeom<R>;
    constexpr int N = G::LH * G::LW;
    for (int base = 0; base < N; base += 8 * NT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * NT + threadIdx.x;
            v[u] = 0.f;
            if (e < N) {
                const int r = e / G::LW, c = e - r * G::LW;
                const int iy = map_axis(py0 + r, a.H, a.in_kind, a.boundary), ix = map_axis(px0 + c, a.W, a.in_kind, a.boundary);
                if (iy >= 0 && ix >= 0) v[u] = plane[(long)iy * a.in_pitch + ix];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * NT + threadIdx.x;
            if (e < N) buf[e] = v[u];                 // LP == LW: the tile is linear in LDS
        }
    }
}

template <typename TX, typename TOut, int R>
__device__ __forceinline__ int run_tiles(const ConvPass &a, float *smem, int t, int t_end, int stride, int tiles_per_plane,
                                         int tiles_x, int my_cls) {
    using G = SepGeom<R>;
    const OutRegion rg = out_region(a);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rgp = tid >> 4, gy = ((tid & 15) + G::YROT * (rgp & 1)) & 15;
    // LDS byte address of the dynamic block (what M0 must hold for buffer 0)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    // first tile
    TileJob job = decode_tile(a, t, tiles_per_plane, tiles_x, 1);
    int cur = 0;
    bool cur_dma = tile_interior<R>(a, rg.y_lo + job.ty * GT - R, rg.x_lo + job.tx * GT - R);
    if (cur_dma)
        tile_dma<R>(a, static_cast<const float *>(a.in) + (long)job.plane * a.in_plane, rg.y_lo + job.ty * GT - R,
                    rg.x_lo + job.tx * GT - R, lds0);
    while (true) {
        const int oy0 = rg.y_lo + job.ty * GT, ox0 = rg.x_lo + job.tx * GT;
        const pb_blur_info *info = job.info;
        const float *ipl = static_cast<const float *>(a.in) + (long)job.plane * a.in_plane;
        const TX *xpl = static_cast<const TX *>(a.x) + (long)job.plane * a.x_plane;
        TOut *opl = static_cast<TOut *>(a.out) + (long)job.plane * a.out_plane;
        float *buf = smem + cur * G::BUF;           // stays an LDS-address-space pointer (no flat accesses)
        if (cur_dma) dma_wait_all();                 // this wave's share of the tile has landed
        else tile_load_mapped<R>(a, ipl, oy0 - R, ox0 - R, buf);
        lds_barrier();                               // ... and everybody else's; previous tile fully consumed
        // this tile's x operand first (hipcc may wait on its own older operations here) ...
        Block4x4Epilogue<TX, TOut> epi;
        epi.prefetch(a, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy);
        // ... then the next tile: class check + start its DMA into the other buffer
        const int tn = t + stride;
        TileJob nxt = job;
        bool more = false, nxt_dma = false;
        if (tn < t_end) {
            nxt = decode_tile(a, tn, tiles_per_plane, tiles_x, 1);
            more = nxt.cls == my_cls;
            if (more) {
                nxt_dma = tile_interior<R>(a, rg.y_lo + nxt.ty * GT - R, rg.x_lo + nxt.tx * GT - R);
                if (nxt_dma)
                    tile_dma<R>(a, static_cast<const float *>(a.in) + (long)nxt.plane * a.in_plane,
                                rg.y_lo + nxt.ty * GT - R, rg.x_lo + nxt.tx * GT - R,
                                lds0 + (unsigned)((cur ^ 1) * G::BUF * sizeof(float)));
            }
        }
        // taps of this tile's image: TP[p] = (h[p], h[p-1]),  HY[m] = (hy[2m], hy[2m+1])
        const PB_CONSTANT float *ckx = as_constant(info->kx) + (PB_KRAD - R), *cky = as_constant(info->ky) + (PB_KRAD - R);
        f2 TP[R + 1], HY[(R + 2) / 2];
#pragma unroll
        for (int q = 0; q <= R; ++q) TP[q] = (f2){ckx[q], q ? ckx[q - 1] : 0.f};
#pragma unroll
        for (int m = 0; m < (R + 2) / 2; ++m) HY[m] = (f2){cky[2 * m], 2 * m + 1 <= R ? cky[2 * m + 1] : 0.f};
        // ---- x pass, in place: each wave owns LH/4 rows; a wave instruction covers 4 rows x 16 groups ----
        {
            constexpr int RPW = (G::LH + 3) / 4;
            const int rsub = lane >> 4, g = ((lane & 15) + G::XROT * (rsub & 1)) & 15;
            for (int it = 0; it < (RPW + 3) / 4; ++it) {
                const int rr = wave * RPW + it * 4 + rsub;
                const bool ok = (it * 4 + rsub) < RPW && rr < G::LH;
                float *row = buf + (ok ? rr : 0) * G::LP;
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
        lds_barrier();
        // ---- y pass: 4 x 4 outputs per thread from the x-filtered tile ----
        f2 axy[4], azw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { axy[r] = (f2){0.f, 0.f}; azw[r] = (f2){0.f, 0.f}; }
        YPassR<R, 0>::run(axy, azw, HY, buf + (rgp * 4) * G::LP + 4 * gy, G::LP);
        float4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = make_float4(axy[r].x, axy[r].y, azw[r].x, azw[r].y);
        if (oy0 < rg.y_hi) epi.finish(a, info, xpl, opl, rg, oy0 + rgp * 4, ox0 + 4 * gy, acc);
        t = tn;
        if (!more) break;
        job = nxt;
        cur ^= 1;
        cur_dma = nxt_dma;
    }
    if (false) dma_wait_all();
    return t;
}

constexpr size_t kSepLds = 2 * sizeof(float) * SepGeom<PB_KRAD>::BUF;      // 61 952 B: two workgroups per CU

template <typename TX, typename TOut>
__global__ __launch_bounds__(NT, 2) void conv_sep_kernel(const ConvPass a, int tiles_per_plane, int tiles_x, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // persistent workgroups; XCD k (= blockIdx % 8, speed only) walks its own contiguous range of tiles
    const int nx = gridDim.x >> 3;
    const int chunk = (total_tiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int t = xcd * chunk + (blockIdx.x >> 3);
    const int t_end = min(total_tiles, (xcd + 1) * chunk);
    while (t < t_end) {
        const TileJob job = decode_tile(a, t, tiles_per_plane, tiles_x, 1);
        switch (job.cls) {                                   // 8 + support class: rank-1 images only
            case 8: t = run_tiles<TX, TOut, 4>(a, smem, t, t_end, nx, tiles_per_plane, tiles_x, 8); break;
            case 9: t = run_tiles<TX, TOut, 8>(a, smem, t, t_end, nx, tiles_per_plane, tiles_x, 9); break;
            case 10: t = run_tiles<TX, TOut, 12>(a, smem, t, t_end, nx, tiles_per_plane, tiles_x, 10); break;
            default: t += nx; break;                         // general taps: conv_tile_kernel does this image
        }
        lds_barrier();                                       // a new run reuses buffer 0
    }
}

template <typename TX, typename TOut>
int launch_sep_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * PB_PAD;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * PB_PAD;
    const int tiles_x = (ow + GT - 1) / GT, tiles_y = (oh + GT - 1) / GT;
    const long tpp = (long)tiles_x * tiles_y;
    const long total = tpp * p.P;
    if (total <= 0 || total > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    static long resident = 0;
    if (!resident) {
        const char *e = getenv("PB_SEP_WGS");
        resident = e ? atol(e) : 512;                        // 256 CUs x 2 workgroups of 62 KB LDS
        if (resident < 8) resident = 512;
    }
    long grid = total < resident ? total : resident;
    grid = (grid + 7) / 8 * 8;
    PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_sep_kernel<TX, TOut>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSepLds));
    hipLaunchKernelGGL((conv_sep_kernel<TX, TOut>), dim3((unsigned)grid), dim3(NT), kSepLds, ctx->stream, p, (int)tpp,
                       tiles_x, (int)total);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// fp32-input rank-1 images (in_dtype must be PB_F32); other images are skipped on the device
int pb_launch_conv_sep(pb_ctx *ctx, const ConvPass &p) {
    const int key = p.x_dtype * 2 + p.out_dtype;
    switch (key) {
        case 0: return launch_sep_typed<float, float>(ctx, p);
        case 1: return launch_sep_typed<float, __half>(ctx, p);
        case 2: return launch_sep_typed<__half, float>(ctx, p);
        default: return launch_sep_typed<__half, __half>(ctx, p);
    }
}
