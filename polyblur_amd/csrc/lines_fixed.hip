// The line transforms of the estimation (blur_estimation.py:112-134, filters.py:159-186) for the line lengths whose plan is
// known when the library is compiled -- the BASELINE sides (columns 2160 = 9 x 16 x 15, 1080 = 6 x 15 x 12, 4320 = 15 x 16 x 18;
// rows 3840 = 15 x 16 x 16, 1920 = 8 x 16 x 15, 7680 = 16 x 20 x 24) and sixteen common picture sides from 512 to 4096
// (PB_COLS_PLANS / PB_ROWS_PLANS below), in the stage orders estimate.hip:launch_cols / rows_plan give those lengths:
//   cols_fixed_kernel        the column transform with the directional maxima folded behind it (grad_cols_kernel's MODE 1), or
//                            storing the y derivative as float / fp16 planes (MODE 0: the gradients halo masking keeps);
//   gray_rows_fixed_kernel   gray + range partials + the row transform from the image's channels (gray_rows_kernel);
//   grad_rows_fixed_kernel   the row transform of float planes, float or fp16 planes out (grad_rows_kernel<.., true>).
//
// What the run-time-plan kernels of estimate.hip compute, operation for operation (the butterflies, the twiddle products and
// the fold of the maxima are the SAME inline functions of fft.h -- every multiply-add in them an explicit FMA --, so every
// record and every gradient sample is bit-identical: tests/test_gpu_estimation_paths.py), but as kernels that hold ONE plan:
//   * no switch over fifteen radices per stage: grad_cols_kernel<1, 7, 1024> is 60 k lines of ISA whose register allocation is
//     that of its widest butterfly (10 vector + 176 scalar registers spilled under the 128-register cap of a 1024-thread
//     workgroup); these are the five stages they run, every trip count and index division a constant, nothing
//     spilled in any of the 300 instantiations;
//   * columns: the gray tile travels global -> LDS by LDS-DMA (16 bytes per lane, 16 rows of a 16-column tile per wave
//     instruction, no staging register), every request of the tile issued at entry; the first stage is then an ordinary
//     in-place LDS stage.  The twiddle table of the line (n complex values) sits in LDS beside the tile where the two fit,
//     fetched by LDS-DMA with the tile: a stage's four twiddle bases are LDS reads, not four dependent gathers from L2 in
//     front of every butterfly;
//   * rows: the complex line sits in LDS with one pad element per 32 (PHI): no stage of the 3840-point plan is left with
//     more than the two passes a 64-lane 8-byte access takes anyway (the run-time-plan kernel: 61 % bank-conflict cycles).
// Which kernel transforms a line is a function of the image's own shape and the options alone: an image gets the same bits
// alone and in a batch.  PB_COLS_FIXED=0 / PB_ROWS_FIXED=0 select the run-time-plan kernels (the tests' reference).
#include "common.h"
#include "fft.h"
#include "lines_fixed.h"

#include <type_traits>

// Lab build only (tools/build_variant.sh lftrace "-DPB_EXPERIMENTAL -DPB_LINES_TRACE" lines_fixed.hip): wall-clock stamps (100 MHz)
// of the one-plan line transforms -- stage by stage from thread 0 of workgroups 0, 1/4, 1/2 and the last of the launch (the
// first 4 x 8 slots the row kernel's, the next the column kernel's; a stamp sees its own wave), and entry / exit / XCC of
// EVERY workgroup (exit = its stores completed) -- read by tools/lines_fixed_trace.py.  Not in the product build.
#if defined(PB_EXPERIMENTAL) && defined(PB_LINES_TRACE)
__device__ unsigned long long g_lines_fixed_trace[2 * 4 * 8];
__device__ unsigned long long g_lines_fixed_span[4];
__device__ unsigned long long g_lines_fixed_wg[2 * 2048 * 3];
__device__ __forceinline__ void pb_lf_stamp(int kernel, int i) {
    if (threadIdx.x != 0) return;
    const unsigned g = gridDim.x, b = blockIdx.x;
    const int w = b == 0 ? 0 : (b == g / 4 ? 1 : (b == g / 2 ? 2 : (b == g - 9 ? 3 : -1)));
    if (w >= 0) g_lines_fixed_trace[(kernel * 4 + w) * 8 + i] = wall_clock64();
}
__device__ __forceinline__ void pb_lf_enter(int kernel) {
    if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        atomicMin(&g_lines_fixed_span[2 * kernel], t);
        if (blockIdx.x < 2048) {
            g_lines_fixed_wg[(kernel * 2048 + blockIdx.x) * 3] = t;
            g_lines_fixed_wg[(kernel * 2048 + blockIdx.x) * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        }
    }
}
__device__ __forceinline__ void pb_lf_exit(int kernel) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        atomicMax(&g_lines_fixed_span[2 * kernel + 1], t);
        if (blockIdx.x < 2048) g_lines_fixed_wg[(kernel * 2048 + blockIdx.x) * 3 + 1] = t;
    }
}
#define PB_LF(k, i) pb_lf_stamp(k, i)
#define PB_LF_ENTER(k) pb_lf_enter(k)
#define PB_LF_EXIT(k) pb_lf_exit(k)
extern "C" int pb_debug_lines_fixed_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lines_fixed_trace), sizeof(unsigned long long) * 64);
}
extern "C" int pb_debug_lines_fixed_span(unsigned long long *host, int reset) {
    if (reset) {
        const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lines_fixed_span), init, sizeof(init));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lines_fixed_span), sizeof(unsigned long long) * 4);
}
extern "C" int pb_debug_lines_fixed_wg(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lines_fixed_wg), sizeof(unsigned long long) * 2 * 2048 * 3);
}
#else
#define PB_LF(k, i)
#define PB_LF_ENTER(k)
#define PB_LF_EXIT(k)
#endif

namespace {

using pbfft::cf;

typedef __amdgpu_buffer_rsrc_t brsrc;
typedef __attribute__((address_space(3))) void lds_void_t;

template <typename T> __device__ __forceinline__ brsrc rsrc_of(const T *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(p), 0, (int)bytes, 0x00020000);
}
// 16 bytes per lane straight into LDS: lane i's bytes land at dst + 16 i (dst wave-uniform); an offset beyond the
// descriptor writes zeros
__device__ __forceinline__ void dma16(brsrc r, void *dst, unsigned voffset, int soffset) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 16, (int)voffset, soffset, 0, 0);
#pragma clang diagnostic pop
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// pbfft::stage<R, DIT> with every size a constant: in place on the NB = 1 << LOGNB interleaved lines of s
template <int R, bool DIT, int N, int L, int LOGNB, int NTH>
__device__ __forceinline__ void fstage(float2 *s, const float2 *tw) {
    constexpr int M = L / R, TW_STEP = N / L, NB = 1 << LOGNB, WORK = (N / R) << LOGNB, STRIDE = M << LOGNB;
#pragma unroll 1
    for (int w = threadIdx.x; w < WORK; w += NTH) {
        const int j = w & (NB - 1), t = w >> LOGNB;
        const int blk = t / M, np = t - blk * M;
        cf *base = reinterpret_cast<cf *>(s) + ((blk * L + np) << LOGNB) + j;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = base[q * STRIDE];
        if (DIT) {
            if (M > 1) {
                cf wq[R];
                pbfft::twiddle_powers<R>(wq, tw, np * TW_STEP);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            }
            pbfft::dft_small<R>(v);
        } else {
            pbfft::dft_small<R>(v);
            if (M > 1) {
                cf wq[R];
                pbfft::twiddle_powers<R>(wq, tw, np * TW_STEP);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) base[q * STRIDE] = v[q];
    }
}

// pbfft::centre_stage<R>: innermost DIF stage, derivative multiplier, innermost DIT stage, in registers
template <int R, int N, int LOGNB, int NTH>
__device__ __forceinline__ void fcentre(float2 *s, const float *__restrict__ drev) {
    constexpr int NB = 1 << LOGNB, WORK = (N / R) << LOGNB;
#pragma unroll 1
    for (int w = threadIdx.x; w < WORK; w += NTH) {
        const int j = w & (NB - 1), t = w >> LOGNB;
        cf *base = reinterpret_cast<cf *>(s) + ((t * R) << LOGNB) + j;
        cf v[R];
        float d[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { v[q] = base[q << LOGNB]; d[q] = drev[t * R + q]; }
        pbfft::dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = (cf){-d[q] * v[q].y, -d[q] * v[q].x};                // conj(i d z)
        pbfft::dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) base[q << LOGNB] = v[q];
    }
}

struct AngleTable7 { float cs[7], sn[7]; };

// m_k = max |cos(t_k) gx - sin(t_k) gy|  (blur_estimation.py:129-133): ColsIO<1, 7>::fold
__device__ __forceinline__ void fold7(float (&best)[7], const AngleTable7 &ang, float dx, float dy) {
#pragma unroll
    for (int k = 0; k < 7; ++k) best[k] = fmaxf(best[k], pbfft::dir_abs(ang.cs[k], ang.sn[k], dx, dy));
}

// R0 x R1 x R2 = the line length; a workgroup = 2 << LOGNB adjacent columns of all rows of one plane.
// TWLDS: the line's twiddle table in LDS behind the tile.  SAT: gradients under the saturation mask (gray > thr) are zero.
// TGY = void: fold the directional maxima (grad_cols_kernel's MODE 1); float / __half: store the derivative instead (MODE 0:
// gray = any float planes, gy_out = their y derivative -- the gradients halo masking keeps of the input image).
template <typename T> __device__ __forceinline__ void st_pair(T *p, float a, float b);
template <> __device__ __forceinline__ void st_pair<float>(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void st_pair<__half>(__half *p, float a, float b) { *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(a, b); }

template <int R0, int R1, int R2, int LOGNB, int NTH, bool TWLDS, bool SAT, typename TGY>
__global__ __launch_bounds__(NTH) void cols_fixed_kernel(const float *__restrict__ gray, const float *__restrict__ gx, int W,
                                                         unsigned *__restrict__ mags, TGY *__restrict__ gy_out, int total_tiles,
                                                         const float2 *__restrict__ tw_g, const float *__restrict__ drev,
                                                         AngleTable7 ang, float thr) {
    constexpr int N = R0 * R1 * R2, NB = 1 << LOGNB, TC = 2 * NB;
    constexpr int TILE_BYTES = N * NB * 8, ROW_BYTES = TC * 4;                 // one tile row: 64 bytes (LOGNB = 3) or 32
    constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES, LANES_PER_ROW = ROW_BYTES / 16;
    constexpr int TILE_DMAS = (TILE_BYTES + 1023) / 1024, TW_DMAS = (N * 8 + 1023) / 1024;
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    // (LDS: [ twiddle table, rounded up to whole 1-KB requests ][ tile, rounded up likewise ])
    float2 *twl = sfft;
    float2 *s = TWLDS ? sfft + TW_DMAS * 128 : sfft;
    const int tiles = W / TC;
    // adjacent column tiles share 128-byte lines: each XCD (= blockIdx % 8) takes a contiguous run of tiles (as grad_cols_kernel)
    const int chunk = gridDim.x >> 3;
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int plane = tile_id / tiles;
    const int c0 = (tile_id - plane * tiles) * TC;
    const int tiles_pad = (tiles + 3) & ~3;
    unsigned *mags_tile = mags + (long)plane * PB_MAX_ANGLES * tiles_pad + (tile_id - plane * tiles);
    const long plane_off = (long)plane * N * W;
    const float *src = gray + plane_off;
    const float *gxp = gx + plane_off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    PB_LF(1, 0);
    PB_LF_ENTER(1);
    // ---- every request of the tile (and of the twiddle table) at entry: global -> LDS, no register in between -------------
    {
        const brsrc rg = rsrc_of(src, (long)N * W * 4);
        const int pitchb = W * 4;
        const unsigned vo = (unsigned)((lane / LANES_PER_ROW) * pitchb + c0 * 4 + (lane % LANES_PER_ROW) * 16);
        char *sb = reinterpret_cast<char *>(s);
#pragma unroll 1
        for (int i = wave; i < TILE_DMAS; i += NTH / 64) {
            // rows ROWS_PER_DMA i ...: the lanes of a last request that runs past the line read beyond the descriptor (zeros,
            // into the tile's rounded-up tail, which nobody reads)
            const int row0 = i * ROWS_PER_DMA;
            if (row0 + ROWS_PER_DMA <= N) dma16(rg, sb + i * 1024, vo, row0 * pitchb);
            else dma16(rg, sb + i * 1024, row0 + (int)(lane / LANES_PER_ROW) < N ? vo + (unsigned)(row0 * pitchb) : 0xfffffff0u, 0);
        }
        if (TWLDS) {
            const brsrc rt = rsrc_of(tw_g, (long)N * 8);
            char *tb = reinterpret_cast<char *>(twl);
#pragma unroll 1
            for (int i = wave; i < TW_DMAS; i += NTH / 64) dma16(rt, tb + i * 1024, (unsigned)(lane * 16), i * 1024);
        }
        PB_LF(1, 1);
        wait_vm0();
        PB_LF(1, 2);
    }
    __syncthreads();
    const float2 *tw = TWLDS ? twl : tw_g;
    // ---- the five stages --------------------------------------------------------------------------------------------------
    fstage<R0, false, N, N, LOGNB, NTH>(s, tw);
    __syncthreads();
    PB_LF(1, 3);
    fstage<R1, false, N, N / R0, LOGNB, NTH>(s, tw);
    __syncthreads();
    PB_LF(1, 4);
    fcentre<R2, N, LOGNB, NTH>(s, drev);
    __syncthreads();
    PB_LF(1, 5);
    fstage<R1, true, N, R1 * R2, LOGNB, NTH>(s, tw);
    __syncthreads();
    PB_LF(1, 6);
    if constexpr (!std::is_void<TGY>::value) {
        // ---- last stage, storing d/dy (pbfft::last_stage<R0> with ColsIO<0>) ----------------------------------------------------
        constexpr int M = N / R0, STRIDE = M << LOGNB;
        TGY *dst = gy_out + plane_off;
#pragma unroll 1
        for (int w = threadIdx.x; w < STRIDE; w += NTH) {
            const int j = w & (NB - 1), np = w >> LOGNB;
            const long idx0 = (long)np * W + c0 + 2 * j;
            cf v[R0];
            const cf *base = reinterpret_cast<const cf *>(s) + w;
#pragma unroll
            for (int q = 0; q < R0; ++q) v[q] = base[q * STRIDE];
            cf wq[R0];
            pbfft::twiddle_powers<R0>(wq, tw, np);
#pragma unroll
            for (int q = 1; q < R0; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            pbfft::dft_small<R0>(v);
#pragma unroll
            for (int q = 0; q < R0; ++q) st_pair<TGY>(dst + idx0 + (long)q * M * W, v[q].x, -v[q].y);
        }
        return;
    }
    // ---- last stage + maxima (pbfft::last_stage<R0> with ColsIO<1, 7>) ------------------------------------------------------
    float best[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) best[k] = 0.f;
    {
        constexpr int M = N / R0, STRIDE = M << LOGNB;
#pragma unroll 1
        for (int w = threadIdx.x; w < STRIDE; w += NTH) {
            const int j = w & (NB - 1), np = w >> LOGNB;
            const long idx0 = (long)np * W + c0 + 2 * j;
            float2 dx[R0], g[SAT ? R0 : 1];
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                dx[q] = *reinterpret_cast<const float2 *>(gxp + idx0 + (long)q * M * W);
                if (SAT) g[q] = *reinterpret_cast<const float2 *>(src + idx0 + (long)q * M * W);
            }
            cf v[R0];
            const cf *base = reinterpret_cast<const cf *>(s) + w;
#pragma unroll
            for (int q = 0; q < R0; ++q) v[q] = base[q * STRIDE];
            cf wq[R0];
            pbfft::twiddle_powers<R0>(wq, tw, np);
#pragma unroll
            for (int q = 1; q < R0; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            pbfft::dft_small<R0>(v);
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                // gradients are zeroed under the saturation mask (blur_estimation.py:117-118): they cannot raise a maximum
                if (!(SAT && g[SAT ? q : 0].x > thr)) fold7(best, ang, dx[q].x, v[q].x);
                if (!(SAT && g[SAT ? q : 0].y > thr)) fold7(best, ang, dx[q].y, -v[q].y);
            }
        }
    }
    PB_LF(1, 7);
    __syncthreads();
    // workgroup maximum of every direction -> one partial per column tile (estimate.hip: reduce_maxima)
    float *red = reinterpret_cast<float *>(sfft);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        float m = best[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave * PB_MAX_ANGLES + k] = m;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float m = red[threadIdx.x];
        for (int w = 1; w < NTH / 64; ++w) m = fmaxf(m, red[w * PB_MAX_ANGLES + threadIdx.x]);
        mags_tile[(long)threadIdx.x * tiles_pad] = __float_as_uint(m);   // m >= 0
    }
    PB_LF_EXIT(1);
}

// ------------------------------------------------------------------------------------------------------------------------
// Row transform (one workgroup = one PAIR of rows as one complex line): gray_rows_kernel / grad_rows_kernel<.., true> of
// estimate.hip for the line lengths 3840 = 15 x 16 x 16, 1920 = 8 x 16 x 15 and 7680 = 16 x 20 x 24 (estimate.hip:rows_plan's
// orders), the same butterflies in the same order.  Beside what a one-plan kernel saves (above), the line sits in LDS with
// one pad element behind every 32 (PHI): the run-time-plan kernel's innermost stage reads s[16 t + q] from lane t -- a stride
// of 128 bytes, every second lane on the same banks (61 % of its LDS cycles were bank conflicts, VERDICT r5 #1d) -- and its
// middle stages meet four blocks of 16 lanes on the same banks; padded, every stage of the 3840-point plan is at the two
// passes a 64-lane 8-byte access takes anyway.  31.7 KB per 3840-point line: still five workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int PHI(int p) { return p + (p >> 5); }

template <int R, bool DIT, int N, int L, int NTH>
__device__ __forceinline__ void rstage(cf *s, const float2 *__restrict__ tw) {
    constexpr int M = L / R, TW_STEP = N / L, WORK = N / R;
#pragma unroll 1
    for (int t = threadIdx.x; t < WORK; t += NTH) {
        const int blk = t / M, np = t - blk * M;
        const int p0 = blk * L + np;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = s[PHI(p0 + q * M)];
        if (DIT) {
            if (M > 1) {
                cf wq[R];
                pbfft::twiddle_powers<R>(wq, tw, np * TW_STEP);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            }
            pbfft::dft_small<R>(v);
        } else {
            pbfft::dft_small<R>(v);
            if (M > 1) {
                cf wq[R];
                pbfft::twiddle_powers<R>(wq, tw, np * TW_STEP);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) s[PHI(p0 + q * M)] = v[q];
    }
}

template <int R, int N, int NTH>
__device__ __forceinline__ void rcentre(cf *s, const float *__restrict__ drev) {
#pragma unroll 1
    for (int t = threadIdx.x; t < N / R; t += NTH) {
        cf v[R];
        float d[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { v[q] = s[PHI(t * R + q)]; d[q] = drev[t * R + q]; }
        pbfft::dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = (cf){-d[q] * v[q].y, -d[q] * v[q].x};                // conj(i d z)
        pbfft::dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) s[PHI(t * R + q)] = v[q];
    }
}

// how a row pair's samples arrive: the image's channels, forming the gray samples exactly as gray_minmax_kernel does
// (blur_estimation.py:36; estimate.hip:GrayRowsIO) and keeping them for the column pass, with the pair's (min, max) ...
template <int CC> struct GrayIn {
    const float *row0;         // channel 0, first row of the pair
    long cstride;
    float *g0;                 // gray, first row of the pair
    int W;
    bool has1;
    float lo, hi;
    __device__ __forceinline__ float2 load(int p) {
        float a = row0[p], b = has1 ? row0[W + p] : 0.f;
        if (CC == 3) {
            const float a1 = row0[cstride + p], a2 = row0[2 * cstride + p];
            const float b1 = has1 ? row0[cstride + W + p] : 0.f, b2 = has1 ? row0[2 * cstride + W + p] : 0.f;
            a = a + a1 + a2; b = b + b1 + b2;
            a = a / 3.0f; b = b / 3.0f;
        }
        g0[p] = a;
        lo = fminf(lo, a); hi = fmaxf(hi, a);
        if (has1) { g0[W + p] = b; lo = fminf(lo, b); hi = fmaxf(hi, b); }
        return make_float2(a, b);
    }
};
// ... or the rows of a float plane as they are (estimate.hip:RowsIO without normalisation)
struct PlainIn {
    const float *row0;
    int W;
    bool has1;
    __device__ __forceinline__ float2 load(int p) const { return make_float2(row0[p], has1 ? row0[W + p] : 0.f); }
};

template <int R0, int R1, int R2, int NTH, class IN, typename TO>
__device__ __forceinline__ void rows_fixed_body(cf *s, IN &in, TO *o0, int W, bool has1, const float2 *__restrict__ tw,
                                                const float *__restrict__ drev) {
    constexpr int N = R0 * R1 * R2, M = N / R0;
    PB_LF(0, 0);
    PB_LF_ENTER(0);
    // first stage: pbfft::first_stage<R0>
#pragma unroll 1
    for (int np = threadIdx.x; np < M; np += NTH) {
        cf v[R0];
#pragma unroll
        for (int q = 0; q < R0; ++q) v[q] = pbfft::to_cf(in.load(np + q * M));
        pbfft::dft_small<R0>(v);
        cf wq[R0];
        pbfft::twiddle_powers<R0>(wq, tw, np);
#pragma unroll
        for (int q = 1; q < R0; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
#pragma unroll
        for (int q = 0; q < R0; ++q) s[PHI(np + q * M)] = v[q];
    }
    PB_LF(0, 1);
    __syncthreads();
    PB_LF(0, 2);
    rstage<R1, false, N, N / R0, NTH>(s, tw);
    __syncthreads();
    PB_LF(0, 3);
    rcentre<R2, N, NTH>(s, drev);
    __syncthreads();
    PB_LF(0, 4);
    rstage<R1, true, N, R1 * R2, NTH>(s, tw);
    __syncthreads();
    PB_LF(0, 5);
    // last stage: pbfft::last_stage<R0> with RowsIO::store
#pragma unroll 1
    for (int np = threadIdx.x; np < M; np += NTH) {
        cf v[R0];
#pragma unroll
        for (int q = 0; q < R0; ++q) v[q] = s[PHI(np + q * M)];
        cf wq[R0];
        pbfft::twiddle_powers<R0>(wq, tw, np);
#pragma unroll
        for (int q = 1; q < R0; ++q) v[q] = pbfft::cmul(v[q], wq[q]);
        pbfft::dft_small<R0>(v);
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            pb_st(o0 + np + q * M, v[q].x);
            if (has1) pb_st(o0 + W + np + q * M, -v[q].y);
        }
    }
    PB_LF(0, 6);
    PB_LF_EXIT(0);
}

constexpr int rows_lds_bytes(int n) { return (n + (n >> 5) + 1) * 8; }
// (the second launch bound = waves per SIMD the register allocation must leave room for: five 256-thread workgroups per CU
// -- what 31.7 KB of LDS allow a 3840-point line -- are five waves per SIMD)
constexpr int rows_waves(int n, int nth) {
    const int wgs = 163840 / rows_lds_bytes(n), w = (wgs * nth + 255) / 256;       // what LDS lets a CU hold, in waves per SIMD
    return w < 1 ? 1 : (w > 5 ? 5 : w);
}

template <int R0, int R1, int R2, int NTH, int CC>
__global__ __launch_bounds__(NTH, rows_waves(R0 * R1 * R2, NTH)) void gray_rows_fixed_kernel(const float *__restrict__ in, float *__restrict__ gray,
                                                                                     float *__restrict__ gx, float2 *__restrict__ part, int H,
                                                                                     const float2 *__restrict__ tw, const float *__restrict__ drev) {
    constexpr int W = R0 * R1 * R2;
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    const int pairs = (H + 1) / 2;
    const int b = blockIdx.x / pairs, pr = blockIdx.x - b * pairs;
    const int r0 = 2 * pr;
    const long HW = (long)H * W;
    GrayIn<CC> io;
    io.row0 = in + (long)b * CC * HW + (long)r0 * W;
    io.cstride = HW;
    io.g0 = gray + (long)b * HW + (long)r0 * W;
    io.W = W; io.has1 = r0 + 1 < H;
    io.lo = INFINITY; io.hi = -INFINITY;
    rows_fixed_body<R0, R1, R2, NTH>(reinterpret_cast<cf *>(sfft), io, gx + (long)b * HW + (long)r0 * W, W, io.has1, tw, drev);
    float lo = io.lo, hi = io.hi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __syncthreads();                                               // (the transform's last reads of sfft)
    float *red = reinterpret_cast<float *>(sfft);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = lo; red[2 * (threadIdx.x >> 6) + 1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NTH / 64; ++w) { lo = fminf(lo, red[2 * w]); hi = fmaxf(hi, red[2 * w + 1]); }
        part[(long)b * pairs + pr] = make_float2(lo, hi);          // one partial per row pair, folded by the parameter kernel
    }
}

template <int R0, int R1, int R2, int NTH, typename TO>
__global__ __launch_bounds__(NTH, rows_waves(R0 * R1 * R2, NTH)) void grad_rows_fixed_kernel(const float *__restrict__ planes, TO *__restrict__ gx, int H,
                                                                                     const float2 *__restrict__ tw, const float *__restrict__ drev) {
    constexpr int W = R0 * R1 * R2;
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    const int pairs = (H + 1) / 2;
    const int plane = blockIdx.x / pairs;
    const int r0 = 2 * (blockIdx.x - plane * pairs);
    PlainIn io{planes + ((long)plane * H + r0) * W, W, r0 + 1 < H};
    rows_fixed_body<R0, R1, R2, NTH>(reinterpret_cast<cf *>(sfft), io, gx + ((long)plane * H + r0) * W, W, io.has1, tw, drev);
}

// whether the line's twiddle table joins the tile in LDS: both (rounded up to whole 1-KB LDS-DMA requests) within what a
// workgroup of that size may take -- all 160 KB on 1024 threads (one per CU), half of it on 512 (two per CU)
constexpr bool cols_twlds(int n, int lognb, int nth) {
    return ((n * (1 << lognb) * 8 + 1023) / 1024 + (n * 8 + 1023) / 1024) * 1024 <= (nth == 1024 ? 160 : 80) * 1024;
}

template <int R0, int R1, int R2, int LOGNB, int NTH, bool TWLDS>
int launch_fixed(pb_ctx *ctx, const float *gray, const float *gx, int P, int W, unsigned *mags, bool sat, const FftPlan *pl,
                 const AngleTable7 &ang, void *gy_out = nullptr, int gy_dtype = PB_F32) {
    constexpr int N = R0 * R1 * R2, NB = 1 << LOGNB;
    constexpr size_t lds = (size_t)((N * NB * 8 + 1023) / 1024) * 1024 + (TWLDS ? (size_t)((N * 8 + 1023) / 1024) * 1024 : 0);
    static_assert(lds <= 160 * 1024, "tile + twiddle table beyond LDS");
    const long blocks = (long)P * (W / (2 * NB));
    if (blocks > 0x7fffffffL) return PB_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((blocks + 7) / 8 * 8);
#define PB_LAUNCH_FIXED(SAT, TGY)                                                                                                \
    do {                                                                                                                         \
        auto k = cols_fixed_kernel<R0, R1, R2, LOGNB, NTH, TWLDS, SAT, TGY>;                                                     \
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
        hipLaunchKernelGGL(k, dim3(grid), dim3(NTH), lds, ctx->stream, gray, gx, W, mags, static_cast<TGY *>(gy_out), (int)blocks, \
                           pl->tw, pl->drev, ang, 0.99f);                                                                        \
    } while (0)
    if (gy_out && gy_dtype == PB_F16) PB_LAUNCH_FIXED(false, __half);
    else if (gy_out) PB_LAUNCH_FIXED(false, float);
    else if (sat) PB_LAUNCH_FIXED(true, void);
    else PB_LAUNCH_FIXED(false, void);
#undef PB_LAUNCH_FIXED
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// PB_ERR_UNSUPPORTED: not one of the compiled plans / tile widths (the caller then takes grad_cols_kernel).
// Which kernel transforms a line must not depend on the batch it arrives in (an image gets the same bits alone and in a
// batch): every tile width estimate.hip:pick_lognb can give one of these line lengths is compiled -- the wide tile of a grid
// that fills the chip (1024 threads, one workgroup per CU) and the narrow one of a lone image (512 threads, two per CU) -- and
// what decides between this file and grad_cols_kernel is the image's own shape and the options alone.
int pb_launch_cols_fixed(pb_ctx *ctx, const float *gray, const float *gx, int P, int H, int W, int lognb, unsigned *mags,
                         int n_angles, int discard_sat, const FftPlan *pl, void *gy_out, int gy_dtype) {
    if ((!gy_out && n_angles != 6) || !pl || pl->bluestein_m || pl->nstage != 3 || pl->n != H || lognb < 1 || (W % (2 << lognb)) != 0)
        return PB_ERR_UNSUPPORTED;
    AngleTable7 ang;
    for (int k = 0; k < 7; ++k) {
        const float t = n_angles > 0 ? 3.14159265358979323846f * (float)k / (float)n_angles : 0.f;       // (as launch_cols)
        ang.cs[k] = std::cos(t);
        ang.sn[k] = std::sin(t);
    }
    const int r0 = pl->radix[0], r1 = pl->radix[1], r2 = pl->radix[2];
    const bool sat = discard_sat != 0;
    // The compiled plans: line length = R0 x R1 x R2 in the order launch_cols gives it (smallest radix from 5 up -- or the one whose
    // butterflies are one trip -- first), with the two tile widths pick_lognb can give that length: NARROW on 512 threads (two
    // workgroups per CU) and WIDE on 1024 (one), 0 = that width does not occur.  The BASELINE lengths first; then the common
    // picture sizes (720p, 1440p, 2K ...), where the same kernels run 1.3 - 2 x faster than the run-time-plan ones.
    // The twiddle table joins the tile in LDS where both fit (cols_twlds).
#define PB_COLS_PLANS(X)                                                                          \
    X(2160, 9, 16, 15, 2, 3) X(1080, 6, 15, 12, 3, 4) X(4320, 15, 16, 18, 1, 2)                    \
    X(512, 2, 16, 16, 3, 4) X(640, 4, 16, 10, 3, 4) X(720, 3, 16, 15, 3, 4) X(768, 3, 16, 16, 3, 4) \
    X(800, 5, 16, 10, 3, 4) X(960, 4, 16, 15, 3, 4) X(1024, 4, 16, 16, 3, 4) X(1200, 5, 16, 15, 3, 0) \
    X(1280, 5, 16, 16, 3, 0) X(1440, 6, 16, 15, 2, 3) X(1536, 6, 16, 16, 2, 3) X(1600, 10, 16, 10, 2, 3) \
    X(2048, 8, 16, 16, 2, 3) X(2560, 10, 16, 16, 2, 0) X(3072, 12, 16, 16, 1, 2) X(4096, 16, 16, 16, 1, 2)
#define PB_COLS_CASE(N_, R0, R1, R2, LN, LW)                                                                                   \
    if (H == N_ && r0 == R0 && r1 == R1 && r2 == R2) {                                                                         \
        if (LW != 0 && lognb == LW) return launch_fixed<R0, R1, R2, (LW ? LW : 1), 1024, cols_twlds(N_, (LW ? LW : 1), 1024)>(ctx, gray, gx, P, W, mags, sat, pl, ang, gy_out, gy_dtype); \
        if (lognb == LN) return launch_fixed<R0, R1, R2, LN, 512, cols_twlds(N_, LN, 512)>(ctx, gray, gx, P, W, mags, sat, pl, ang, gy_out, gy_dtype); \
    }
    PB_COLS_PLANS(PB_COLS_CASE)
#undef PB_COLS_CASE
    return PB_ERR_UNSUPPORTED;
}

// The row transform of W-sample lines with the plan compiled in; C == 0: a float plane as it is (in = P planes of H x W),
// else gray + range partials + transform from the image's C channels (in = B images; part: B x pairs partials).
// nth: the thread count estimate.hip:rows_threads gives the launch.  PB_ERR_UNSUPPORTED: not compiled -- the caller runs its own.
int pb_launch_rows_fixed(pb_ctx *ctx, const float *in, int C, float *gray, void *gx, float2 *part, long images, int H, int W, int nth,
                         const FftPlan *pl, int gx_dtype) {
    if (gx_dtype != PB_F32 && !(gx_dtype == PB_F16 && C == 0)) return PB_ERR_UNSUPPORTED;
    if (!pl || pl->bluestein_m || pl->nstage != 3 || pl->n != W || (C != 0 && C != 1 && C != 3)) return PB_ERR_UNSUPPORTED;
    const long blocks = images * ((H + 1) / 2);
    if (blocks > 0x7fffffffL || blocks < 1) return PB_ERR_UNSUPPORTED;
    const int r0 = pl->radix[0], r1 = pl->radix[1], r2 = pl->radix[2];
#define PB_ROWS_FIXED(R0, R1, R2, NTH)                                                                                              \
    do {                                                                                                                            \
        constexpr size_t lds = rows_lds_bytes(R0 * R1 * R2);                                                                        \
        if (C == 0 && gx_dtype == PB_F16) {                                                                                         \
            auto k = grad_rows_fixed_kernel<R0, R1, R2, NTH, __half>;                                                               \
            if (lds > 48 * 1024) PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream, in, static_cast<__half *>(gx), H, pl->tw, pl->drev); \
        } else if (C == 0) {                                                                                                        \
            auto k = grad_rows_fixed_kernel<R0, R1, R2, NTH, float>;                                                                \
            if (lds > 48 * 1024) PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream, in, static_cast<float *>(gx), H, pl->tw, pl->drev); \
        } else if (C == 3) {                                                                                                        \
            auto k = gray_rows_fixed_kernel<R0, R1, R2, NTH, 3>;                                                                    \
            if (lds > 48 * 1024) PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream, in, gray, static_cast<float *>(gx), part, H, pl->tw, pl->drev);    \
        } else {                                                                                                                    \
            auto k = gray_rows_fixed_kernel<R0, R1, R2, NTH, 1>;                                                                    \
            if (lds > 48 * 1024) PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream, in, gray, static_cast<float *>(gx), part, H, pl->tw, pl->drev);    \
        }                                                                                                                           \
        PB_LAUNCH_CHECK();                                                                                                          \
        return PB_OK;                                                                                                               \
    } while (0)
    // (the compiled plans, in the order rows_plan gives them: the BASELINE widths, then the common picture widths)
#define PB_ROWS_PLANS(X)                                                                         \
    X(3840, 15, 16, 16) X(1920, 8, 16, 15)                                                       \
    X(512, 2, 16, 16) X(640, 4, 16, 10) X(720, 3, 16, 15) X(768, 3, 16, 16) X(800, 5, 16, 10) X(960, 4, 16, 15) \
    X(1024, 4, 16, 16) X(1200, 5, 16, 15) X(1280, 5, 16, 16) X(1440, 6, 16, 15) X(1536, 6, 16, 16) X(1600, 10, 16, 10) \
    X(2048, 8, 16, 16) X(2560, 10, 16, 16) X(3072, 12, 16, 16) X(4096, 16, 16, 16)
#define PB_ROWS_CASE(N_, R0, R1, R2)                                   \
    if (W == N_ && r0 == R0 && r1 == R1 && r2 == R2) {                 \
        if (nth == 256) PB_ROWS_FIXED(R0, R1, R2, 256);                \
        if (nth == 128) PB_ROWS_FIXED(R0, R1, R2, 128);                \
    }
    PB_ROWS_PLANS(PB_ROWS_CASE)
#undef PB_ROWS_CASE
    if (W == 7680 && r0 == 16 && r1 == 20 && r2 == 24 && nth == 512) PB_ROWS_FIXED(16, 20, 24, 512);
#undef PB_ROWS_FIXED
    return PB_ERR_UNSUPPORTED;
}

bool pb_lines_fixed_shape(int H, int W) {
    bool h = false, w = W == 7680;
#define PB_H_CASE(N_, R0, R1, R2, LN, LW) h = h || H == N_;
#define PB_W_CASE(N_, R0, R1, R2) w = w || W == N_;
    PB_COLS_PLANS(PB_H_CASE)
    PB_ROWS_PLANS(PB_W_CASE)
#undef PB_H_CASE
#undef PB_W_CASE
    return h && w && (W % 32) == 0;
}
