// Tile-spectrum body of the reblurring pass, one WAVE per window pair.
//
// Same pass, same operands, same result as conv_fft.hip (one Horner step  t <- K*t + coef*x  of the polynomial
// deconvolution, reference deblurring.py:122-138 / :141-169, or one edgetaper blend, edgetaper.py:30-32, with the
// boundary models of filters.py:14-49): the exact 2-D stencil of a 64 x 64 window is evaluated as a circular correlation
// -- forward 2-D DFT, product with the kernel's real 64 x 64 spectrum (khat_kernel, conv_fft.hip), inverse DFT, of which
// the samples at least R from the window's edge are kept (overlap-save); two horizontally adjacent real windows ride
// one complex transform, z = A + iB.
//
// What differs is who does it.  conv_fft.hip spreads a window pair over 512 threads: eight 8-point butterfly stages,
// six LDS round trips and seven workgroup barriers per pair, every stage exposing an LDS or barrier latency.  Here a
// window pair belongs to ONE wave and a whole 64-point line to one lane (64 complex values = 128 registers):
//
//   load      lane = window column x, register y = window row: 2 x 64 row-segment loads of 256 bytes per wave
//   columns   fft64 in registers (two radix-8 stages, compile-time twiddles in scalar registers), no LDS, no barrier
//   transpose through the wave's private LDS tile: lane = transformed row, register = column
//   rows      fft64, x real spectrum (64 values per lane, [x position][y position] layout: coalesced), inverse fft64
//   transpose back
//   columns   inverse fft64; lane = window column again: the epilogue (scale, + coef * x, clamp) and the stores walk
//             rows with 256-byte segments per wave instruction
//
// Two LDS round trips per pair instead of six, no workgroup barrier at all (a wave's DS operations execute in order;
// only the compiler has to be told, wave_lds_fence), rows wave-uniform -- every row offset, boundary mapping and
// validity test is scalar work -- and columns one map per lane and window.  Border windows, the taper epilogue and
// ragged ends therefore run the same code as interior ones with different offsets (an out-of-range buffer offset
// loads 0 / drops the store).
//
// The transposes keep only HALF a window pair in LDS at a time (16.6 KB per wave, eight waves per CU): a 64 x 64
// transpose is the swap of the two off-diagonal 32 x 32 quadrants -- v_permlane32_swap between the wave's halves --
// followed by a transpose inside each quadrant, and the quadrant pairs go through the same 32 x 64 LDS tile one after
// the other.
//
// Work items: window pairs of the images whose record selects this body, every image with its own tile size (prefix sum
// over the batch's pb_fft_sel records inside the kernel: no idle slots for images with larger tiles, no host read-back);
// workgroup b runs on XCD b % 8 (observed, used for speed only) and every XCD takes the same contiguous eighth of every
// plane's pairs, so neighbouring windows share their halos in that XCD's L2.
// No MFMA, no library FFT.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_fft_common.h"

namespace {

// timing-only ablations (wrong results; PB_EXTRA_FLAGS=-DPB_ABL=bits): 1 no window loads, 2 no spectrum DMA, 4 no x loads /
// stores, 8 no transposes, 16 no butterflies
#ifndef PB_ABL
#define PB_ABL 0
#endif

constexpr int WF_ROWS = 32;                          // LDS tile rows per wave (half a window pair)
constexpr size_t kWfLdsWave = sizeof(float2) * WF_ROWS * FT_P;

// cos / sin (2 pi m / 64): indexed with compile-time constants only (the values fold into scalar moves)
static __device__ const float kC64[64] = {
    1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f,
    0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f,
    0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f,
    -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f,
    -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f,
    -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f,
    -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f,
    -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f,
    -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f, 0.0f,
    0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f,
    0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f,
    0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f};
static __device__ const float kS64[64] = {
    0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f,
    0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f,
    0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f, 1.0f,
    0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f,
    0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f,
    0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f,
    -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f,
    -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f,
    -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f,
    -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f,
    -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f,
    -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f};

// ---------------------------------------------------------------------------------------------
// 64-point DFT of a line held in registers.  Index split n = 8 n1 + n2 -> k = k1 + 8 k2 (decimation in frequency):
// register 8 k1 + k2 of the transformed line holds frequency k1 + 8 k2 -- the order khat_kernel lays the spectrum out
// in -- and the inverse runs the mirrored stages, so nothing is ever reordered.  Unnormalised (khat carries 1/4096).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf8(cf (&v)[8]) { if constexpr (!(PB_ABL & 16)) pbfft::dft_small<8>(v); }
__device__ __forceinline__ void ibf8(cf (&v)[8]) { if constexpr (!(PB_ABL & 16)) idft8(v); }
// forward stage 1 of group n2: registers 8 n1 + n2 over n1, then x W64^(n2 k1)
template <int N2> __device__ __forceinline__ void fwd_stage1(cf (&v)[64]) {
    cf a[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) a[n1] = v[8 * n1 + N2];
    bf8(a);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        const int m = (N2 * k1) & 63;
        v[8 * k1 + N2] = m ? cmul_s(a[k1], (cf){kC64[m], -kS64[m]}) : a[k1];
    }
}
// forward stage 2 of group k1: registers 8 k1 + n2 over n2
template <int K1> __device__ __forceinline__ void fwd_stage2(cf (&v)[64]) {
    cf b[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b[n2] = v[8 * K1 + n2];
    bf8(b);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[8 * K1 + k2] = b[k2];
}
// inverse stage 2 of group k1, then x conj W64^(n2 k1)
template <int K1> __device__ __forceinline__ void inv_stage2(cf (&v)[64]) {
    cf b[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) b[k2] = v[8 * K1 + k2];
    ibf8(b);
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        const int m = (n2 * K1) & 63;
        v[8 * K1 + n2] = m ? cmul_conj_s(b[n2], (cf){kC64[m], -kS64[m]}) : b[n2];
    }
}
// the two of them around the product with the real spectrum (group k1 of a transformed row)
template <int K1> __device__ __forceinline__ void centre_stage(cf (&v)[64], const float (&kh)[64]) {
    cf b[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b[n2] = v[8 * K1 + n2];
    bf8(b);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) b[k2] = b[k2] * kh[8 * K1 + k2];
    ibf8(b);
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        const int m = (n2 * K1) & 63;
        v[8 * K1 + n2] = m ? cmul_conj_s(b[n2], (cf){kC64[m], -kS64[m]}) : b[n2];
    }
}
// inverse stage 1 of group n2: its eight outputs are registers (window rows, in the last pass) 8 n1 + n2
template <int N2> __device__ __forceinline__ void inv_stage1(cf (&v)[64]) {
    cf a[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) a[k1] = v[8 * k1 + N2];
    ibf8(a);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[8 * n1 + N2] = a[n1];
}
#define PB_EACH8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
__device__ __forceinline__ void fft64_fwd(cf (&v)[64]) {
#define PB_S(i) fwd_stage1<i>(v);
    PB_EACH8(PB_S)
#undef PB_S
#define PB_S(i) fwd_stage2<i>(v);
    PB_EACH8(PB_S)
#undef PB_S
}
__device__ __forceinline__ void fft64_fwd_stage1(cf (&v)[64]) {
#define PB_S(i) fwd_stage1<i>(v);
    PB_EACH8(PB_S)
#undef PB_S
}
__device__ __forceinline__ void fft64_centre(cf (&v)[64], const float (&kh)[64]) {
#define PB_S(i) centre_stage<i>(v, kh);
    PB_EACH8(PB_S)
#undef PB_S
}
__device__ __forceinline__ void fft64_inv_stage1(cf (&v)[64]) {
#define PB_S(i) inv_stage1<i>(v);
    PB_EACH8(PB_S)
#undef PB_S
}
__device__ __forceinline__ void fft64_inv_stage2(cf (&v)[64]) {
#define PB_S(i) inv_stage2<i>(v);
    PB_EACH8(PB_S)
#undef PB_S
}

// v_permlane32_swap: lanes 32..63 of `hi_part` <-> lanes 0..31 of `lo_part`
__device__ __forceinline__ void swap_halves(cf &hi_part, cf &lo_part) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(hi_part.x), __float_as_uint(lo_part.x), false, false);
    const u2 y = __builtin_amdgcn_permlane32_swap(__float_as_uint(hi_part.y), __float_as_uint(lo_part.y), false, false);
    hi_part = (cf){__uint_as_float(x[0]), __uint_as_float(y[0])};
    lo_part = (cf){__uint_as_float(x[1]), __uint_as_float(y[1])};
}

// Transpose of the 64 x 64 matrix whose column `lane` sits in lane `lane`'s registers: afterwards lane l holds row l
// (register c = column c).  The off-diagonal 32 x 32 quadrants swap between the wave's halves, then each half
// transposes its two quadrants through the 32 x 64 LDS tile, one quadrant pair after the other.  The same routine
// takes the matrix back.  STRIDED: the reads of a quadrant pair are issued in the order the next stage consumes them
// (register 8 n1 + n2, n2-major).  Row pitch 65 complex values: the writes (consecutive lanes, consecutive 8-byte
// words) and the reads (lane i reads word 65 i + c: 32 different banks pairs per half wave) are conflict-free.
__device__ __forceinline__ void transpose64(cf (&v)[64], float2 *Z, int lane) {
    if constexpr (PB_ABL & 8) return;
#pragma unroll
    for (int r = 0; r < 32; ++r) swap_halves(v[r], v[32 + r]);
    const float2 *rd = Z + (lane & 31) * FT_P + (lane & 32);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 32; ++r) Z[r * FT_P + lane] = pbfft::to_f2(v[32 * h + r]);
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 32; ++c) v[32 * h + c] = pbfft::to_cf(rd[c]);
        wave_lds_fence();
    }
}

// Geometry of a pass, per window halo class (index R / 4 - 1), computed on the host.
struct WGeom {
    int pairs_x[3], njobs[3], per[3];       // window pairs per row, per plane, per plane and XCD
    float inv_pairs_x[3], inv_per[3];
};

// inclusive prefix sum over the wave
__device__ __forceinline__ int wave_scan(int x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(x, d, 64);
        if (lane >= d) x += t;
    }
    return x;
}

#ifdef PB_WF_TRACE
// Debug build only (python -m polyblur_amd.build with PB_EXTRA_FLAGS=-DPB_WF_TRACE): shader-clock stamps of the first
// waves' phases, read back with pb_debug_wf_trace (tools/wf_trace.py).
constexpr int kTraceWaves = 8192, kTraceStamps = 14;
__device__ unsigned long long g_wf_trace[kTraceWaves * kTraceStamps];
#define PB_T(i) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PB_TWAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PB_TRT(i) do { if (tr) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per wave: up to kJobSlots jobs x {taken, done (100 MHz ticks), step << 28 | pair}
constexpr int kJobWaves = 2048, kJobSlots = 40;
__device__ unsigned long long g_wf_jobs[kJobWaves * kJobSlots * 3];
#else
#define PB_T(i)
#define PB_TWAIT()
#define PB_TRT(i)
#endif

// 16 bytes per lane from a buffer straight into LDS (1 KiB per wave instruction, no staging registers): lane i's bytes
// land at lds + 16 i.  An offset at or beyond the descriptor's size writes zeros.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ lds_char *lds_ptr(void *p) { return (lds_char *)p; }
template <int IMM> __device__ __forceinline__ void dma16(brsrc r, lds_char *dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 16, (int)voffset, soffset, IMM, 0);
}
#pragma clang diagnostic pop
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v ld_b128(brsrc r, unsigned voffset, int soffset) {
    return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, soffset, 0));
}
// 16-byte store.  The whole offset travels in the vector register and the scalar-offset field stays 0: a VALU write to
// the data registers right behind a store of more than 8 bytes reads as a hazard to the compiler only in that form (it
// assumes a register in the scalar-offset field buys the wait state; on gfx950 it does not -- the first data dword of the
// last lanes was sporadically replaced by the next instruction's result).
__device__ __forceinline__ void st_b128(brsrc r, unsigned voffset, int soffset, f4v v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, (int)(voffset + (unsigned)soffset), 0, 0);
}

// four horizontally adjacent samples of the x operand / of the output as one piece: 16 bytes of fp32, 8 bytes of fp16
template <typename T> struct Piece4;
template <> struct Piece4<float> {
    typedef f4v raw;
    static __device__ __forceinline__ raw ld(brsrc r, unsigned vo, int so) { return ld_b128(r, vo, so); }
    static __device__ __forceinline__ f4v to_f(raw v) { return v; }
    static __device__ __forceinline__ void st(brsrc r, unsigned vo, int so, f4v v) { st_b128(r, vo, so, v); }
};
template <> struct Piece4<__half> {
    typedef uint2 raw;
    static __device__ __forceinline__ raw ld(brsrc r, unsigned vo, int so) {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        const u2v t = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, (int)vo, so, 0));
        return make_uint2(t[0], t[1]);
    }
    static __device__ __forceinline__ f4v to_f(raw v) {
        const float2 a = __half22float2(__builtin_bit_cast(__half2, v.x)), b = __half22float2(__builtin_bit_cast(__half2, v.y));
        return (f4v){a.x, a.y, b.x, b.y};
    }
    static __device__ __forceinline__ void st(brsrc r, unsigned vo, int so, f4v v) {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
        const u2v t = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
        __builtin_amdgcn_raw_buffer_store_b64(t, r, (int)vo, so, 0);
    }
};

// One window pair.  zb: the wave's LDS region (kWfLdsWave bytes); kp: the image's spectrum, [x position][y position].
template <int R, bool FAST, typename TIn, typename TX, typename TOut>
__device__ __forceinline__ void wave_pair(const ConvPass &a, const pb_blur_info *info, int plane, int ty, int pxi, char *zb,
                                          const float *kp, unsigned long long *tr) {
    constexpr int T = FT_N - 2 * R;
    PB_T(1);
    float2 *Z = reinterpret_cast<float2 *>(zb);
    float *Zf = reinterpret_cast<float *>(zb);
    const OutRegion rg = out_region(a);
    const int wy0 = rg.y_lo + ty * T - R;                       // window origin, padded coordinates
    const int wxA = rg.x_lo + 2 * pxi * T - R, wxB = wxA + T;
    const bool hasB = wxB + R < rg.x_hi;
    const int lane = threadIdx.x & 63;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    cf v[64];

    // ---- the window: lane = column, register = row ----
    {
        const brsrc rin = plane_rsrc(ipl, a.in_plane);
        const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
        const unsigned pitchb = (unsigned)a.in_pitch * (unsigned)sizeof(TIn);
        if constexpr (FAST && (PB_ABL & 1)) {
#pragma unroll
            for (int y = 0; y < 64; ++y) v[y] = (cf){(float)(lane + y), (float)(lane - y)};
        } else if constexpr (FAST) {
            // Interior fp32 pair on 16-byte boundaries: the union of the two windows (64 + T columns) goes global -> LDS in
            // 16-byte pieces (four-byte loads straight into the registers cost one vector memory instruction per row and
            // tile, 128 per pair instead of 32: the pass was bound by their issue), LDS rows of 128 floats -- one wave
            // instruction fills two of them --, sixteen rows at a time through two LDS buffers: chunk k holds the rows
            // 8 n1 + 2k, 8 n1 + 2k + 1 of the first-stage groups n2 = 2k, 2k + 1, so the column transform starts on what
            // has arrived while the rest is in flight, and every lane picks its column's two samples per row with one
            // ds_read2_b32.  Two chunks are requested before the first is waited for and the next as soon as a buffer has
            // been read: one memory latency per pair.
            constexpr int C4 = (FT_N + T) / 4;
            const int c = lane & 31;
            // (lanes past the union's last piece repeat it rather than go out of range: see the note at dma16)
            const unsigned vo = (unsigned)(lane >> 5) * pitchb + (unsigned)(wxA - lo + 4 * min(c, C4 - 1)) * 4u;
            const unsigned row0 = (unsigned)(wy0 - lo) * pitchb;
            lds_char *zl = lds_ptr(zb);
            auto request = [&](int k, int buf) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dma16<0>(rin, zl + buf * 8192 + j * 1024, vo, (int)(row0 + (unsigned)(8 * j + 2 * k) * pitchb));
            };
            // (the LDS reads are issued behind the compiler's back: it would make every read of either buffer wait for ALL
            // outstanding LDS-DMA; the waits for the right chunk are placed by hand, and the wait that follows a chunk's
            // reads names their destinations, so that nothing using them can be scheduled above it)
            const unsigned la = lds_addr(zb) + (unsigned)lane * 4u;
            auto pick = [&](int k, int buf) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned ad = la + (unsigned)(buf * 8192 + j * 1024);
                    asm volatile("ds_read2_b32 %0, %1 offset1:%2" : "=v"(v[8 * j + 2 * k]) : "v"(ad), "n"(T));
                    asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:%2" : "=v"(v[8 * j + 2 * k + 1]) : "v"(ad), "n"(128 + T));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[2 * k]), "+v"(v[2 * k + 1]), "+v"(v[8 + 2 * k]), "+v"(v[9 + 2 * k]), "+v"(v[16 + 2 * k]),
                             "+v"(v[17 + 2 * k]), "+v"(v[24 + 2 * k]), "+v"(v[25 + 2 * k]), "+v"(v[32 + 2 * k]), "+v"(v[33 + 2 * k]), "+v"(v[40 + 2 * k]),
                             "+v"(v[41 + 2 * k]), "+v"(v[48 + 2 * k]), "+v"(v[49 + 2 * k]), "+v"(v[56 + 2 * k]), "+v"(v[57 + 2 * k]) :: "memory");
            };
            request(0, 0); request(1, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            pick(0, 0);
            request(2, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            pick(1, 1);
            request(3, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            pick(2, 0);
            wait_vm0();
            pick(3, 1);
        } else if (wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && hasB) {
            const unsigned colA = (unsigned)(wxA - lo + lane) * (unsigned)sizeof(TIn), colB = colA + T * (unsigned)sizeof(TIn);
            const unsigned row0 = (unsigned)(wy0 - lo) * pitchb;
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const int y = 8 * (q & 7) + (q >> 3);
                const int so = (int)(row0 + (unsigned)y * pitchb);
                v[y] = (cf){BufIO<TIn>::ld(rin, colA, so), BufIO<TIn>::ld(rin, colB, so)};
            }
        } else {
            // border window: columns mapped through the boundary model once per lane, rows on the scalar side
            const int ixa = map_axis(wxA + lane, a.W, a.in_kind, a.boundary, a.pad);
            const int ixb = hasB ? map_axis(wxB + lane, a.W, a.in_kind, a.boundary, a.pad) : -1;
            const unsigned colA = ixa >= 0 ? (unsigned)ixa * (unsigned)sizeof(TIn) : kNoAccess;
            const unsigned colB = ixb >= 0 ? (unsigned)ixb * (unsigned)sizeof(TIn) : kNoAccess;
            const bool wrap = a.boundary == PB_WRAP;
            const int base = wrap ? __builtin_amdgcn_readfirstlane(wrap_idx(wy0, Hp)) : wy0;
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const int y = 8 * (q & 7) + (q >> 3);
                int p = base + y;
                if (wrap) { while (p >= Hp) p -= Hp; }
                const bool ok = wrap || (p >= 0 && p < Hp);
                const int iy = ok ? (a.in_kind == SRC_VIRTUAL ? min(max(p - a.pad, 0), a.H - 1) : p) : 0;
                const int so = (int)((unsigned)iy * pitchb);
                v[y] = (cf){BufIO<TIn>::ld(rin, ok ? colA : kNoAccess, so), BufIO<TIn>::ld(rin, ok ? colB : kNoAccess, so)};
            }
        }
    }
    PB_T(2);
    fft64_fwd(v);                                               // columns
    PB_T(3);
    transpose64(v, Z, lane);
    PB_T(4);
    {
        // the image's spectrum (16 KB, resident in L2), [x position][y position]: lane = transformed row py reads kh[px][py]
        // for every px -- 64 coalesced requests that travel while the first row stage runs (through LDS they cost a
        // DMA pass, 64 LDS reads and the wait for both)
        const brsrc rk = plane_rsrc(kp, (long)FT_N * FT_N);
        float kh[64];
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            if constexpr (PB_ABL & 2) kh[p] = 1.0f; else kh[p] = BufIO<float>::ld(rk, (unsigned)lane * 4u, p * (FT_N * 4));
        }
        fft64_fwd_stage1(v);                                    // rows
        PB_T(5);
        fft64_centre(v, kh);                                    // stage 2, x spectrum, inverse stage 2
    }
    fft64_inv_stage1(v);
    PB_T(6);
    transpose64(v, Z, lane);
    PB_T(7);

    // ---- epilogue: lane = window column again ----
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const int xmax = virt ? a.W - 1 : Wp - 1, ymax = virt ? a.H - 1 : Hp - 1, xsh = virt ? a.pad : 0;
    const unsigned xpitchb = (unsigned)a.x_pitch * (unsigned)sizeof(TX), opitchb = (unsigned)a.out_pitch * (unsigned)sizeof(TOut);
    const brsrc rx = plane_rsrc(xpl, a.x_plane);
    const brsrc ro = plane_rsrc(opl, a.out_plane);
    const bool colin = lane >= R && lane < FT_N - R;
    const float sc = a.scale, cfx = a.coef;
    const bool cl = a.clamp01 != 0;
    const int oy0 = wy0 + R, oxA = wxA + R;
    if constexpr (FAST) {
        // Complete interior pair, plain Horner epilogue, everything on 16-byte boundaries: the 2T-wide block of outputs goes
        // through an LDS tile (written by columns, read back as 16-byte row pieces), NR rows at a time, so that the
        // x operand arrives and the result leaves in 16-byte accesses: 2 NK vector memory instructions per NR rows instead
        // of 4 NR.  The x operand of the round after next is requested when a round has been stored.
                constexpr int NRND = R == 12 ? 2 : (R == 8 ? 3 : 4);
        constexpr int NR = T / NRND, C = 2 * T / 4, NK = (NR * C + 63) / 64;
        constexpr int PT = 2 * T;
        static_assert(NR * NRND == T, "rounds must tile the rows");
        typename Piece4<TX>::raw xq[2][NK];
        constexpr unsigned XP = 4 * sizeof(TX), OP = 4 * sizeof(TOut);                 // bytes per piece of x / of the output
        const int xso = (int)((unsigned)(oy0 - xsh) * xpitchb + (unsigned)(oxA - xsh) * (unsigned)sizeof(TX));
        const int oso = (int)((unsigned)(oy0 - oo) * opitchb + (unsigned)(oxA - oo) * (unsigned)sizeof(TOut));
        auto request = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NRND) {
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int e = 64 * k + lane, rl = e / C, ch = e - rl * C;
                    const unsigned vo = rl < NR ? (unsigned)(rl + q * NR) * xpitchb + (unsigned)ch * XP : kNoAccess;
                    xq[q & 1][k] = Piece4<TX>::ld(rx, vo, xso);
                }
            }
        };
        auto round = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NRND) {
                {
                    // (halo lanes write to a scratch copy of the tile behind it rather than sit out under an exec mask)
                    float *zt = Zf + (colin ? lane - R : NR * PT + lane);
#pragma unroll
                    for (int i = 0; i < NR; ++i) { zt[i * PT] = v[R + q * NR + i].x; zt[i * PT + T] = v[R + q * NR + i].y; }
                }
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int e = 64 * k + lane, rl = e / C, ch = e - rl * C;
                    const f4v acc = *reinterpret_cast<const f4v *>(Zf + min(rl, NR - 1) * PT + 4 * ch);
                    const f4v x4 = Piece4<TX>::to_f(xq[q & 1][k]);
                    f4v o;
                    o.x = fmaf(sc, acc.x, cfx * x4.x); o.y = fmaf(sc, acc.y, cfx * x4.y);
                    o.z = fmaf(sc, acc.z, cfx * x4.z); o.w = fmaf(sc, acc.w, cfx * x4.w);
                    if (cl) {
                        o.x = fminf(fmaxf(o.x, 0.f), 1.f); o.y = fminf(fmaxf(o.y, 0.f), 1.f);
                        o.z = fminf(fmaxf(o.z, 0.f), 1.f); o.w = fminf(fmaxf(o.w, 0.f), 1.f);
                    }
                    const unsigned vo = rl < NR ? (unsigned)(rl + q * NR) * opitchb + (unsigned)ch * OP : kNoAccess;
                    Piece4<TOut>::st(ro, vo, oso, o);
                }
                wave_lds_fence();
            }
        };
        typedef std::integral_constant<int, 0> Q0; typedef std::integral_constant<int, 1> Q1; typedef std::integral_constant<int, 2> Q2;
        typedef std::integral_constant<int, 3> Q3; typedef std::integral_constant<int, 4> Q4; typedef std::integral_constant<int, 5> Q5;
        fft64_inv_stage2(v);                                    // columns
        request(Q0{});
        fft64_inv_stage1(v);
        PB_T(8);
        request(Q1{});
        round(Q0{}); request(Q2{});
        round(Q1{}); request(Q3{});
        round(Q2{}); request(Q4{});
        round(Q3{}); request(Q5{});
        PB_T(9);
        PB_TWAIT();
        PB_T(10);
        PB_TRT(13);
        return;
    }
    (void)oxA;
    const int pxA = wxA + lane, pxB = wxB + lane;
    const bool okA = colin && pxA < rg.x_hi, okB = colin && hasB && pxB < rg.x_hi;
    const unsigned xoffA = okA ? (unsigned)min(max(pxA - xsh, 0), xmax) * (unsigned)sizeof(TX) : kNoAccess;
    const unsigned xoffB = okB ? (unsigned)min(max(pxB - xsh, 0), xmax) * (unsigned)sizeof(TX) : kNoAccess;
    const unsigned ooffA = okA ? (unsigned)(pxA - oo) * (unsigned)sizeof(TOut) : kNoAccess;
    const unsigned ooffB = okB ? (unsigned)(pxB - oo) * (unsigned)sizeof(TOut) : kNoAccess;
    const bool taper = a.epilogue == EPI_TAPER;
    float txa = 0.f, txb = 0.f;
    if (taper) {
        txa = taper_weight(info->acorr_x, min(max(pxA, 0), Wp - 1), Wp);
        txb = taper_weight(info->acorr_x, min(max(pxB, 0), Wp - 1), Wp);
    }
    // Border pairs, the taper blend, narrower types: the last transform's second stage finishes the window rows 8 n1 + n2
    // group by group (n2 = 0 .. 7); each group's rows go through the epilogue and to memory at once, and the x operand
    // travels in a ring four groups deep -- the loads of group n2 + 4 are issued when group n2 has been stored.
    float xa[8][8], xb[8][8];
    auto request = [&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int y = 8 * n1 + n2;
            if (y >= R && y < FT_N - R) {
                const int xr = min(max(wy0 + y - xsh, 0), ymax);
                const int so = (int)((unsigned)xr * xpitchb);
                xa[n2][n1] = BufIO<TX>::ld(rx, xoffA, so); xb[n2][n1] = BufIO<TX>::ld(rx, xoffB, so);
            }
        }
    };
    auto finish = [&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
        inv_stage1<n2>(v);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int y = 8 * n1 + n2;
            if (y >= R && y < FT_N - R) {
                const int py = wy0 + y;
                if (py < rg.y_hi) {
                    float ra, rb;
                    if (taper) {
                        const float tyw = taper_weight(info->acorr_y, py, Hp);
                        const float ala = tyw * txa, alb = tyw * txb;
                        ra = ala * xa[n2][n1] + (1.f - ala) * v[y].x; rb = alb * xb[n2][n1] + (1.f - alb) * v[y].y;
                    } else {
                        ra = fmaf(sc, v[y].x, cfx * xa[n2][n1]); rb = fmaf(sc, v[y].y, cfx * xb[n2][n1]);
                    }
                    if (cl) { ra = fminf(fmaxf(ra, 0.f), 1.f); rb = fminf(fmaxf(rb, 0.f), 1.f); }
                    const int so = (int)((unsigned)(py - oo) * opitchb);
                    BufIO<TOut>::st(ro, ooffA, so, ra); BufIO<TOut>::st(ro, ooffB, so, rb);
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
    typedef std::integral_constant<int, 6> I6; typedef std::integral_constant<int, 7> I7;
    request(I0{}); request(I1{}); request(I2{}); request(I3{});
    fft64_inv_stage2(v);                                        // columns
    PB_T(8);
    finish(I0{}); request(I4{});
    finish(I1{}); request(I5{});
    finish(I2{}); request(I6{});
    finish(I3{}); request(I7{});
    finish(I4{}); finish(I5{}); finish(I6{}); finish(I7{});
    PB_T(9);
    PB_TWAIT();
    PB_T(10);
    PB_TRT(13);
}

// Whether a pair takes the all-16-byte path: fp32 everywhere, plain Horner epilogue, both windows inside the source
// without boundary mapping, both tiles complete inside the output region, the x operand addressed without clamping, and
// rows / origins on 16-byte boundaries.
template <int R, typename TIn, typename TX, typename TOut>
__device__ __forceinline__ bool pair_is_fast(const ConvPass &a, int ty, int pxi) {
    // (the window must be fp32 -- the planes between the steps always are; the x operand and the output may be fp16: four
    // samples are then an 8-byte piece)
    if (sizeof(TIn) != 4 || sizeof(TX) < 2 || sizeof(TOut) < 2 || a.epilogue != EPI_HORNER) return false;
    constexpr int T = FT_N - 2 * R;
    const OutRegion rg = out_region(a);
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const int wy0 = rg.y_lo + ty * T - R, wxA = rg.x_lo + 2 * pxi * T - R, wxB = wxA + T;
    const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0, xsh = virt ? a.pad : 0;
    const int xw = virt ? a.W : Wp, xh = virt ? a.H : Hp;
    const int oy0 = wy0 + R, oxA = wxA + R;
    return wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && oy0 + T <= rg.y_hi && oxA + 2 * T <= rg.x_hi &&
           oy0 - xsh >= 0 && oxA - xsh >= 0 && oy0 + T - xsh <= xh && oxA + 2 * T - xsh <= xw &&
           ((a.in_pitch | a.x_pitch | a.out_pitch | (wxA - lo) | (oxA - xsh) | (oxA - oo)) & 3) == 0;
}

// One wave (= one workgroup) per window pair; the GRID is the job list.  The jobs are the window pairs of the images whose
// record selects this body, every image with its own tile size (prefix sum over the batch's pb_fft_sel records: no host
// read-back; the grid is sized for the smallest tile and the surplus workgroups leave at once).  Workgroup b belongs to
// list b % 8 (the XCD it is observed to run on -- used for speed only) at position b / 8; every list owns the same eighth
// of EVERY plane's pairs -- a contiguous run, so neighbouring windows share their halos in that XCD's L2 -- in the order
// image, plane, pair.
//
// (Measured and dropped: the three Horner steps of a polynomial in ONE launch -- step-major lists, per-plane completion
// counters, write-through stores for the planes a later step reads, one agent-scope acquire per pair -- once with
// persistent waves taking pairs from queue heads in memory, once with the grid as the queue.  With every
// synchronisation compiled out the single launch takes exactly what the three launches take, 254 us per 4K polynomial:
// the launch tails it removes were not idle time, the waves that remain in a tail run faster; with the acquire and the
// write-through stores in place 282 us.  The persistent form also cost 50 more spilled registers.
//  Three waves per SIMD instead of two -- 168 registers per lane and 13 KiB of LDS per wave: the spectrum streamed from L2
// eight values ahead, the window staged through a ring of three 8-row LDS buffers, the transposes through a 16-row tile
// after a v_permlane16_swap level, the x operand three pieces per lane ahead -- was built, passed every test and ran at
// 119 us per 4K pass against 85: with 40 registers beside the window pair nothing can be requested far enough ahead
// (epilogue 30 k cycles per pair instead of 6 k, the centre stage 15.5 k instead of 4.8 k), twelve waves per CU queue
// on the LDS (transposes 9 k cycles instead of 3.5 k) and the compiler still spilled 28 window rows per pair.  The 128
// registers a wave has beside its window pair at two waves per SIMD are what hides this kernel's latencies.)
template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(64, 2) void conv_wfft_kernel(const ConvPass a, const WGeom g) {
    extern __shared__ __attribute__((aligned(16))) char zb[];
    const int lane = threadIdx.x & 63;
    unsigned long long *tr = nullptr;
#ifdef PB_WF_TRACE
    if (blockIdx.x < kTraceWaves) tr = g_wf_trace + (long)blockIdx.x * kTraceStamps;
    PB_T(0);
    PB_TRT(12);
#endif
    const int C = a.C, B = a.P / C;
    const int q = (int)(blockIdx.x & 7u);
    int rem = (int)(blockIdx.x >> 3);                  // position in the list
    int img = 0, R = 0;
    bool fold = false;
    if (B == 1) {
        // (one image: its record is read on the scalar side -- no trip through the vector memory queue)
        const PB_CONSTANT pb_fft_sel *s0 = as_constant(a.fsel);
        if (!s0->use_fft || !poly_match(a.poly, s0->poly)) return;
        R = s0->rf;
        fold = a.poly == 2 && s0->poly != 0;
    } else {
        // list entries of image i: its share of every plane
        auto share_of = [&](int i) -> int {
            if (i >= B) return 0;
            const pb_fft_sel s = a.fsel[i];
            if (!s.use_fft || !poly_match(a.poly, s.poly)) return 0;
            return (s.rf <= 4 ? g.per[0] : (s.rf <= 8 ? g.per[1] : g.per[2])) * C;
        };
        bool work = false;
        int base = 0;
        for (int c0 = 0; c0 < B; c0 += 64) {
            const int n = share_of(c0 + lane), incl = wave_scan(n, lane);
            const unsigned long long m = __ballot(base + incl > rem);
            if (m) {
                const int l = __builtin_ctzll(m);
                img = c0 + l;
                rem -= base + (__builtin_amdgcn_readlane(incl, l) - __builtin_amdgcn_readlane(n, l));
                work = true;
                break;
            }
            base += __builtin_amdgcn_readlane(incl, 63);
        }
        if (!work) return;
        img = __builtin_amdgcn_readfirstlane(img); rem = __builtin_amdgcn_readfirstlane(rem);
        R = as_constant(a.fsel + img)->rf;
        fold = a.poly == 2 && as_constant(a.fsel + img)->poly != 0;
    }
    const int c = R <= 4 ? 0 : (R <= 8 ? 1 : 2);
    const int pl = __builtin_amdgcn_readfirstlane(div_small(rem, g.inv_per[c]));
    if (pl >= C) return;                               // (one image: positions beyond its planes)
    const int pair = q * g.per[c] + (rem - pl * g.per[c]);
    if (pair >= g.njobs[c]) return;                    // (the ragged end of the last list's run)
    const int ty = __builtin_amdgcn_readfirstlane(div_small(pair, g.inv_pairs_x[c])), pxi = pair - ty * g.pairs_x[c];
    const int plane = img * C + pl;
    const float *kp = a.khat + (long)img * (FT_N * FT_N);
    const pb_blur_info *info = a.info + img;
    const ConvPass af = fold_pass(a, fold);
#define PB_RUN(RR)                                                                                  \
    if (pair_is_fast<RR, TIn, TX, TOut>(af, ty, pxi)) wave_pair<RR, true, TIn, TX, TOut>(af, info, plane, ty, pxi, zb, kp, tr); \
    else wave_pair<RR, false, TIn, TX, TOut>(af, info, plane, ty, pxi, zb, kp, tr);
    if (c == 2) { PB_RUN(12) } else if (c == 1) { PB_RUN(8) } else { PB_RUN(4) }
#undef PB_RUN
}

bool wfft_geometry(const ConvPass &p, WGeom &g, long &total_max) {
    FftGeom f;
    if (!fft_geometry(p, f)) return false;
    for (int c = 0; c < 3; ++c) {
        g.pairs_x[c] = f.pairs_x[c]; g.njobs[c] = f.njobs[c]; g.per[c] = (f.njobs[c] + 7) / 8;
        g.inv_pairs_x[c] = f.inv_pairs_x[c]; g.inv_per[c] = 1.0f / (float)g.per[c];
    }
    total_max = (long)f.njobs[2] * p.P;
    return total_max > 0 && total_max <= (1L << 22);
}

template <typename TIn, typename TX, typename TOut>
int launch_wfft_typed(pb_ctx *ctx, const ConvPass &p) {
    WGeom g;
    long total_max = 0;
    if (!wfft_geometry(p, g, total_max)) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: too many windows for the tile-spectrum body");
    const long groups = 8L * g.per[2] * p.P;             // list entries if every image had the smallest tiles
    hipLaunchKernelGGL((conv_wfft_kernel<TIn, TX, TOut>), dim3((unsigned)groups), dim3(64), kWfLdsWave, ctx->stream, p, g);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

#ifdef PB_WF_TRACE
extern "C" int pb_debug_wf_trace_clear(void) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wf_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * kTraceStamps * kTraceWaves);
}
extern "C" int pb_debug_wf_jobs(unsigned long long *host, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wf_jobs)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * kJobWaves * kJobSlots * 3);
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wf_jobs), sizeof(unsigned long long) * kJobWaves * kJobSlots * 3);
}
extern "C" int pb_debug_wf_trace(unsigned long long *host, int n_waves) {
    if (n_waves > kTraceWaves) n_waves = kTraceWaves;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wf_trace), sizeof(unsigned long long) * kTraceStamps * n_waves);
}
#endif

// PB_ERR_UNSUPPORTED: dtype combination not built, or a pass so small that the workgroup form is the faster one (the caller
// falls back to it).  A wave takes ~19 us for its pair whatever the size of the launch, a 512-thread workgroup ~8 us: a pass
// whose pairs do not even fill the chip's 2048 wave slots once is a race of single pairs (700 x 500: 420 pairs, 0.31 ms per
// call through the workgroup form against 0.38 ms), from about one and a half rounds on the wave form wins.
int pb_launch_conv_wfft(pb_ctx *ctx, const ConvPass &p) {
    static const long min_jobs = [] { const char *e = getenv("PB_WAVE_MIN_JOBS"); return e ? atol(e) : 3000L; }();
    {
        WGeom g;
        long total_max = 0;
        if (wfft_geometry(p, g, total_max) && total_max < min_jobs) return PB_ERR_UNSUPPORTED;
    }
    ProfScope prof(ctx, PB_PROF_CONV_FFT);
    // fp32 planes, and the second and third Horner step of fp16 images (fp32 temporaries in, fp16 x operand, fp32 or fp16
    // out); the first step of an fp16 image -- its window is fp16 -- stays with the workgroup form (measured through this
    // body's element-wise loader: 64 x 1080p fp16 37.96 / 37.89 ms per step against 38.03 / 38.13, 8K fp16 7.05 / 7.11
    // against 7.03 / 7.11 -- equal, so the instantiation is not built)
    switch (p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype) {
        case 0: return launch_wfft_typed<float, float, float>(ctx, p);
        case 3: return launch_wfft_typed<float, __half, float>(ctx, p);
        case 4: return launch_wfft_typed<float, __half, __half>(ctx, p);
        default: return PB_ERR_UNSUPPORTED;
    }
}
