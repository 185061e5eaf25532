// Pieces shared by the two tile-spectrum bodies of the reblurring pass (conv_fft.hip: one workgroup per window pair;
// conv_wfft.hip: one wave per window pair): the 64-point twiddle table, packed complex products, buffer-descriptor I/O,
// single-instruction LDS reads, and the pass geometry.
#pragma once
#include "common.h"
#include "conv_common.h"
#include "fft.h"

namespace {

using pbfft::cf;

constexpr int FT_N = 64;          // window side
constexpr int FT_P = 65;          // LDS row pitch in complex values
constexpr int FT_NT = 512;         // one radix-8 butterfly per thread and stage
constexpr int KH_NT = 256;
constexpr size_t kFftLds = sizeof(float2) * FT_N * FT_P;

// W64^m = exp(-2 pi i m / 64)
static __device__ const float2 kW64[64] = {
    {1.0f, 0.0f}, {0.99518472f, -0.0980171412f}, {0.980785251f, -0.195090324f}, {0.956940353f, -0.290284663f},
    {0.923879504f, -0.382683426f}, {0.881921291f, -0.471396744f}, {0.831469595f, -0.555570245f}, {0.773010433f, -0.634393275f},
    {0.707106769f, -0.707106769f}, {0.634393275f, -0.773010433f}, {0.555570245f, -0.831469595f}, {0.471396744f, -0.881921291f},
    {0.382683426f, -0.923879504f}, {0.290284663f, -0.956940353f}, {0.195090324f, -0.980785251f}, {0.0980171412f, -0.99518472f},
    {0.0f, -1.0f}, {-0.0980171412f, -0.99518472f}, {-0.195090324f, -0.980785251f}, {-0.290284663f, -0.956940353f},
    {-0.382683426f, -0.923879504f}, {-0.471396744f, -0.881921291f}, {-0.555570245f, -0.831469595f}, {-0.634393275f, -0.773010433f},
    {-0.707106769f, -0.707106769f}, {-0.773010433f, -0.634393275f}, {-0.831469595f, -0.555570245f}, {-0.881921291f, -0.471396744f},
    {-0.923879504f, -0.382683426f}, {-0.956940353f, -0.290284663f}, {-0.980785251f, -0.195090324f}, {-0.99518472f, -0.0980171412f},
    {-1.0f, 0.0f}, {-0.99518472f, 0.0980171412f}, {-0.980785251f, 0.195090324f}, {-0.956940353f, 0.290284663f},
    {-0.923879504f, 0.382683426f}, {-0.881921291f, 0.471396744f}, {-0.831469595f, 0.555570245f}, {-0.773010433f, 0.634393275f},
    {-0.707106769f, 0.707106769f}, {-0.634393275f, 0.773010433f}, {-0.555570245f, 0.831469595f}, {-0.471396744f, 0.881921291f},
    {-0.382683426f, 0.923879504f}, {-0.290284663f, 0.956940353f}, {-0.195090324f, 0.980785251f}, {-0.0980171412f, 0.99518472f},
    {0.0f, 1.0f}, {0.0980171412f, 0.99518472f}, {0.195090324f, 0.980785251f}, {0.290284663f, 0.956940353f},
    {0.382683426f, 0.923879504f}, {0.471396744f, 0.881921291f}, {0.555570245f, 0.831469595f}, {0.634393275f, 0.773010433f},
    {0.707106769f, 0.707106769f}, {0.773010433f, 0.634393275f}, {0.831469595f, 0.555570245f}, {0.881921291f, 0.471396744f},
    {0.923879504f, 0.382683426f}, {0.956940353f, 0.290284663f}, {0.980785251f, 0.195090324f}, {0.99518472f, 0.0980171412f}
};

// a * conj(w)
__device__ __forceinline__ cf cmul_conj(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                      // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;                                                                                                    // (.. + a.y w.y, a.y w.x - ..)
}

// a * w and a * conj(w) for a wave-uniform w held in a scalar register pair
__device__ __forceinline__ cf cmul_s(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
__device__ __forceinline__ cf cmul_conj_s(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}

// a * h.x  and  a * h.y  for real h (two spectrum values share a register pair)
__device__ __forceinline__ cf scale_lo(cf a, cf h) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(h));
    return r;
}
__device__ __forceinline__ cf scale_hi(cf a, cf h) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(h));
    return r;
}

// Plane-relative accesses of the interior path go through buffer descriptors: a 32-bit byte offset per lane plus a scalar
// one (tile B sits T samples to the right of tile A), no 64-bit address arithmetic, and an offset at or beyond the plane's size makes a load return 0 and a store vanish -- the halo
// rows and columns of a window, which produce no output, need no branch.
typedef __amdgpu_buffer_rsrc_t brsrc;
constexpr unsigned kNoAccess = 0x80000000u;     // planes are smaller than 2 GiB (checked on the host)
template <typename T> __device__ __forceinline__ brsrc plane_rsrc(const T *plane, long elems) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(plane), 0, (int)(elems * (long)sizeof(T)), 0x00020000);
}
template <typename T> struct BufIO;
template <> struct BufIO<float> {
    static __device__ __forceinline__ float ld(brsrc r, unsigned b, int sb) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)b, sb, 0)); }
    static __device__ __forceinline__ void st(brsrc r, unsigned b, int sb, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)b, sb, 0); }
};
template <> struct BufIO<__half> {
    static __device__ __forceinline__ float ld(brsrc r, unsigned b, int sb) { return __half2float(__builtin_bit_cast(__half, __builtin_amdgcn_raw_buffer_load_b16(r, (int)b, sb, 0))); }
    static __device__ __forceinline__ void st(brsrc r, unsigned b, int sb, float v) { __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, __float2half_rn(v)), r, (int)b, sb, 0); }
};
template <> struct BufIO<unsigned char> {
    static __device__ __forceinline__ float ld(brsrc r, unsigned b, int sb) { return pb_from_ubyte(__builtin_amdgcn_raw_buffer_load_b8(r, (int)b, sb, 0)); }
    static __device__ __forceinline__ void st(brsrc r, unsigned b, int sb, float v) { __builtin_amdgcn_raw_buffer_store_b8((unsigned char)pb_to_ubyte(v), r, (int)b, sb, 0); }
};

// Eight complex values SB bytes apart from the LDS tile as eight ds_read_b64 (256 bytes per clock).  Left to the compiler
// neighbouring reads are paired into ds_read2_b64, which moves 128 bytes per clock (MI355X_MICROARCH.md, LDS table).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)p;
}
#pragma clang diagnostic pop
template <int SB> __device__ __forceinline__ void lds_read8(cf (&v)[8], const float2 *p) {
    const unsigned a = lds_addr(p);
    asm volatile("ds_read_b64 %0, %8\n\t"
                 "ds_read_b64 %1, %8 offset:%9\n\t"
                 "ds_read_b64 %2, %8 offset:%10\n\t"
                 "ds_read_b64 %3, %8 offset:%11\n\t"
                 "ds_read_b64 %4, %8 offset:%12\n\t"
                 "ds_read_b64 %5, %8 offset:%13\n\t"
                 "ds_read_b64 %6, %8 offset:%14\n\t"
                 "ds_read_b64 %7, %8 offset:%15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(a), "n"(SB), "n"(2 * SB), "n"(3 * SB), "n"(4 * SB), "n"(5 * SB), "n"(6 * SB), "n"(7 * SB)
                 : "memory");
}

// inverse 8-point DFT (unnormalised): the forward one with its outputs read in mirrored order
__device__ __forceinline__ void idft8(cf (&v)[8]) {
    pbfft::dft_small<8>(v);
    cf t;
    t = v[1]; v[1] = v[7]; v[7] = t;
    t = v[2]; v[2] = v[6]; v[6] = t;
    t = v[3]; v[3] = v[5]; v[5] = t;
}

// ConvPass.poly == 2: what the first step's launch does for an image whose whole polynomial is one pass
__device__ __forceinline__ ConvPass fold_pass(const ConvPass &a, bool fold) {
    ConvPass f = a;
    if (fold) {
        f.out = a.out2; f.out_kind = a.out2_kind; f.out_pitch = a.out2_pitch; f.out_plane = a.out2_plane;
        f.scale = 1.f; f.coef = 0.f; f.clamp01 = a.clamp2;
    }
    return f;
}
// whether a launch of the tile-spectrum body takes an image (per its pb_fft_sel.poly; see ConvPass.poly)
__device__ __forceinline__ bool poly_match(int pass_poly, int sel_poly) { return pass_poly == 2 || (sel_poly != 0) == (pass_poly != 0); }

// Geometry of a pass for the three window halo classes (index R / 4 - 1), computed on the host: window pairs per row, pairs
// per plane, pairs per plane and XCD; the reciprocals turn the kernel's divisions of small integers into one multiply.
struct FftGeom {
    int pairs_x[3], njobs[3], per[3];
    float inv_pairs_x[3];
    int slots;                    // per[2]: pairs per plane and XCD for the smallest tile (the job grid is sized for it)
    float inv_slots;
};
__device__ __forceinline__ int div_small(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }   // n < 2^22, exact

// Geometry of the pass; false when a plane or the batch has more window pairs than the kernel's index arithmetic takes
// (2^22: the caller then keeps the stencil bodies).
inline bool fft_geometry(const ConvPass &p, FftGeom &g) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    for (int c = 0; c < 3; ++c) {
        const int T = FT_N - 8 * (c + 1);
        const long tiles_x = (ow + T - 1) / T, tiles_y = (oh + T - 1) / T;
        const long px = (tiles_x + 1) / 2, nj = px * tiles_y;
        if (nj > (1L << 22)) return false;
        g.pairs_x[c] = (int)px; g.njobs[c] = (int)nj; g.per[c] = (int)((nj + 7) / 8);
        g.inv_pairs_x[c] = 1.0f / (float)px;
    }
    g.slots = g.per[2];
    g.inv_slots = 1.0f / (float)g.slots;
    const long total = (long)g.slots * p.P;
    return total > 0 && total <= (1L << 22);
}

}  // namespace
