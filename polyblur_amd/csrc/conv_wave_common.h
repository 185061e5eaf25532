// Pieces shared by the wave-private tile-spectrum kernels (conv_wfft.hip: 64 x 64 windows, one wave per window pair;
// conv_w128.hip: 128 x 128 windows, four waves per pair): the 64-point transform of a line held in one lane's registers,
// the exchange between a wave's halves, LDS-DMA and 16-byte buffer accesses.
#pragma once
#include "conv_fft_common.h"

namespace {

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
__device__ __forceinline__ void bf8(cf (&v)[8]) { pbfft::dft_small<8>(v); }
__device__ __forceinline__ void ibf8(cf (&v)[8]) { idft8(v); }
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
template <int K1> __device__ __forceinline__ void centre_stage(cf (&v)[64], const float (&kh)[8]) {
    cf b[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b[n2] = v[8 * K1 + n2];
    bf8(b);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) b[k2] = b[k2] * kh[k2];
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

// inclusive prefix sum over the wave
__device__ __forceinline__ int wave_scan(int x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(x, d, 64);
        if (lane >= d) x += t;
    }
    return x;
}


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
// (four bytes per lane: lane l's sample lands at dst + 4 l -- a gather of one 256-byte LDS row from 64 arbitrary addresses)
template <int IMM> __device__ __forceinline__ void dma4(brsrc r, lds_char *dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 4, (int)voffset, soffset, IMM, 0);
}
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

template <> struct Piece4<unsigned char> {
    typedef unsigned raw;
    static __device__ __forceinline__ raw ld(brsrc r, unsigned vo, int so) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)vo, so, 0); }
    static __device__ __forceinline__ f4v to_f(raw u) {
        return (f4v){pb_from_ubyte(u & 255u), pb_from_ubyte((u >> 8) & 255u), pb_from_ubyte((u >> 16) & 255u), pb_from_ubyte(u >> 24)};
    }
    static __device__ __forceinline__ void st(brsrc r, unsigned vo, int so, f4v v) {
        const unsigned u = pb_to_ubyte(v.x) | (pb_to_ubyte(v.y) << 8) | (pb_to_ubyte(v.z) << 16) | (pb_to_ubyte(v.w) << 24);
        __builtin_amdgcn_raw_buffer_store_b32(u, r, (int)vo, so, 0);
    }
};

}  // namespace
