// Device helpers shared by the two bodies of the stencil pass (conv.hip).
#pragma once
#include "common.h"

namespace {

#define PB_CONSTANT __attribute__((address_space(4)))

// loads through the constant address space become scalar (s_load) when the address is uniform
template <typename T> __device__ __forceinline__ const PB_CONSTANT T *as_constant(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const PB_CONSTANT T *)p;
#pragma clang diagnostic pop
}

// compiler-only ordering of LDS traffic inside one wavefront (no instruction is emitted)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wrap_idx(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// padded coordinate -> source index along one axis, or -1 for "reads as zero"
__device__ __forceinline__ int map_axis(int p, int n_unpadded, int kind, int boundary, int pad) {
    const int np = n_unpadded + 2 * pad;
    if (boundary == PB_WRAP) p = wrap_idx(p, np);
    else if (p < 0 || p >= np) return -1;
    if (kind == SRC_VIRTUAL) return min(max(p - pad, 0), n_unpadded - 1);
    return p;
}

__device__ __forceinline__ float taper_weight(const float *ac, int p, int n) {
    // v[p] = 1 - z[p]/z[0], z = circular autocorrelation with period n-1, z[n-1] := z[0] (edgetaper.py:11-15):
    // z[p] = ac[p] + ac[n-1-p] with ac = 0 beyond lag 24 -- non-zero only within 24 samples of either end, and
    // both terms count at once only when the padded axis is shorter than 49 samples (n-1 >= 25 always: z[0] = ac[0]).
    const int q = n - 1 - p;
    const float z = ((p < PB_KSIZE) ? ac[p] : 0.f) + ((q < PB_KSIZE) ? ac[q] : 0.f);
    return 1.f - z * __frcp_rn(ac[0]);
}

template <typename T> __device__ __forceinline__ float4 ld4(const T *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<__half>(const __half *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
template <> __device__ __forceinline__ float4 ld4<unsigned char>(const unsigned char *p) {
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return make_float4(pb_from_ubyte(u & 255u), pb_from_ubyte((u >> 8) & 255u), pb_from_ubyte((u >> 16) & 255u), pb_from_ubyte(u >> 24));
}
template <typename T> __device__ __forceinline__ void st4(T *p, float4 v);
template <> __device__ __forceinline__ void st4<unsigned char>(unsigned char *p, float4 v) {
    *reinterpret_cast<unsigned *>(p) = pb_to_ubyte(v.x) | (pb_to_ubyte(v.y) << 8) | (pb_to_ubyte(v.z) << 16) | (pb_to_ubyte(v.w) << 24);
}
template <> __device__ __forceinline__ void st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<__half>(__half *p, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned *>(&a);
    u.y = *reinterpret_cast<const unsigned *>(&b);
    *reinterpret_cast<uint2 *>(p) = u;
}

// Where the outputs of a pass live, in padded coordinates.
struct OutRegion { int y_lo, y_hi, x_lo, x_hi; };
__device__ __forceinline__ OutRegion out_region(const ConvPass &a) {
    if (a.out_kind == OUT_INTERIOR) return OutRegion{a.pad, a.pad + a.H, a.pad, a.pad + a.W};
    return OutRegion{0, a.H + 2 * a.pad, 0, a.W + 2 * a.pad};
}

// The x operand of 4 horizontally adjacent outputs at padded (py, px..px+3): rows and columns of a virtual
// (un-padded) source clamp (replicate pad); a 16-byte load when the four columns are contiguous in the source.
template <typename TX>
__device__ __forceinline__ float4 load_x4(const ConvPass &a, const TX *xpl, int py, int px) {
    const int H = a.H, W = a.W;
    const int xr = (a.x_kind == SRC_VIRTUAL) ? min(max(py - a.pad, 0), H - 1) : py;
    const int xc0 = (a.x_kind == SRC_VIRTUAL) ? px - a.pad : px;
    const int xcmax = (a.x_kind == SRC_VIRTUAL) ? W : W + 2 * a.pad;
    const TX *xrow = xpl + (long)xr * a.x_pitch;
    if (xc0 >= 0 && xc0 + 3 < xcmax && ((a.x_pitch | xc0) & 3) == 0) return ld4<TX>(xrow + xc0);
    float4 t;
    t.x = pb_ld(xrow + min(max(xc0, 0), xcmax - 1));
    t.y = pb_ld(xrow + min(max(xc0 + 1, 0), xcmax - 1));
    t.z = pb_ld(xrow + min(max(xc0 + 2, 0), xcmax - 1));
    t.w = pb_ld(xrow + min(max(xc0 + 3, 0), xcmax - 1));
    return t;
}

// Epilogue + store of 4 horizontally adjacent outputs at padded (py, px..px+3); xq = their x operand.
template <typename TOut>
__device__ __forceinline__ void finish4(const ConvPass &a, const pb_blur_info *info, TOut *opl, const OutRegion &rg, int py,
                                        int px, float4 acc, float4 xq) {
    if (py < rg.y_lo || py >= rg.y_hi || px >= rg.x_hi || px + 3 < rg.x_lo) return;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    float av[4] = {acc.x, acc.y, acc.z, acc.w};
    const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
    const bool full = px >= rg.x_lo && px + 3 < rg.x_hi;
    float ty = 1.f;
    if (a.epilogue == EPI_TAPER) ty = taper_weight(info->acorr_y, py, Hp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if (a.epilogue == EPI_TAPER) {
            const float al = ty * taper_weight(info->acorr_x, px + i, Wp);
            v = al * xv[i] + (1.f - al) * av[i];
        } else {
            v = a.scale * av[i] + a.coef * xv[i];
        }
        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        av[i] = v;
    }
    const int orow = (a.out_kind == OUT_INTERIOR) ? py - a.pad : py;
    const int oc0 = (a.out_kind == OUT_INTERIOR) ? px - a.pad : px;
    TOut *orow_p = opl + (long)orow * a.out_pitch;
    if (full && ((a.out_pitch | oc0) & 3) == 0) {
        st4<TOut>(orow_p + oc0, make_float4(av[0], av[1], av[2], av[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px + i >= rg.x_lo && px + i < rg.x_hi) pb_st(orow_p + oc0 + i, av[i]);
    }
}


typedef float f2 __attribute__((ext_vector_type(2)));

// acc.lo += taps.(SWAP ? hi : lo) * d,  acc.hi += taps.(SWAP ? lo : hi) * d,  d = data.(HI ? hi : lo)
// One v_pk_fma_f32 = two FMAs; the tap pair sits in an aligned SGPR pair, op_sel does the rest.
template <int SWAP, int HI> __device__ __forceinline__ void pk_bcast_data(f2 &acc, f2 taps, f2 data) {
    if (!SWAP && !HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(taps), "v"(data));
    if (!SWAP && HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(taps), "v"(data));
    if (SWAP && !HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "s"(taps), "v"(data));
    if (SWAP && HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(taps), "v"(data));
}
// acc += taps.(HI ? hi : lo) * data   (both halves use the same tap)
template <int HI> __device__ __forceinline__ void pk_bcast_tap(f2 &acc, f2 taps, f2 data) {
    if (!HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(taps), "v"(data));
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(taps), "v"(data));
}


}  // namespace
