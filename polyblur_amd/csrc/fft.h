// In-LDS mixed-radix FFT used for the spectral derivative (reference filters.py:159-186).
//
// The reference computes image gradients by Fourier interpolation: multiply the 2-D spectrum
// by 2*pi*i*f and transform back.  The multiplier depends on one frequency axis only, so the
// x-gradient is a 1-D periodic spectral derivative of every row and the y-gradient of every
// column.  Each workgroup keeps whole lines resident in LDS:
//
//   natural order --DIF stages--> digit-reversed spectrum --(x i*d[k]/N, conj)-->
//   --transposed (DIT) stages--> natural order, conj  ==  derivative of the line
//
// Two real lines are packed as one complex line (z = a + i b): the multiplier is Hermitian
// (its Nyquist bin is zero, exactly what `real()` drops in filters.py:180,183), so the real
// and imaginary parts stay independent.  All stages are in place (one LDS buffer, one
// barrier per stage); NB interleaved lines are transformed together (element (p, j) lives at
// s[p*NB + j]) so that consecutive lanes touch consecutive LDS words.
//
// Direct plans of two or more stages run the fused form (spectral_derivative_fused): the first stage reads global memory,
// the innermost DIF stage + multiplier + innermost DIT stage run in registers, the last stage feeds the caller's epilogue
// -- four LDS round trips and barriers for a 3840-point line instead of eight.
//
// Lengths whose prime factors are all <= 7 use radices up to 16 (composites 16/15/12/10/9/8/6 are evaluated
// in registers as two-level Cooley-Tukey, so a 3840-point line needs 3 LDS passes, not 6) -- and 18 / 20 / 24 where
// that saves a stage (4320 = 18 x 16 x 15, 7680 = 24 x 20 x 16: the 8K sides); any other length
// goes through Bluestein's chirp-z with a power-of-two inner length (plan.bluestein_m).
#pragma once
#include "common.h"

// (lab builds: PB_FT(i) stamps the stages of spectral_derivative_fused -- estimate.hip defines it under
// -DPB_EXPERIMENTAL -DPB_LINES_TRACE, tools/lines_trace.py reads the stamps)
#ifndef PB_FT
#define PB_FT(i)
#endif

namespace pbfft {


// (float2 helpers: the Bluestein path and the callers' I/O)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---- butterfly arithmetic on complex values held as one aligned register pair ------------------------------------------
// A complex add is one v_pk_add_f32, a complex product two packed instructions (op_sel picks the halves, neg_lo / neg_hi
// the signs), a multiplication by -i is folded into the add that consumes it.  Written out because the compiler, given
// the same formulas on scalars, packs them after the fact and pays one register move for every two packed operations
// (343 vector instructions per radix-16 butterfly with its 15 twiddles; ~190 this way).
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf to_cf(float2 v) { return (cf){v.x, v.y}; }
__device__ __forceinline__ float2 to_f2(cf v) { return make_float2(v.x, v.y); }

// a * w
__device__ __forceinline__ cf cmul(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                      // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;                                                                                                    // (.. - a.y w.y, .. + a.y w.x)
}
// a * w for a compile-time constant w (kept in a scalar register pair)
__device__ __forceinline__ cf cmulc(cf a, cf w) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// a + (-i) b  and  a - (-i) b          ((-i) b = (b.y, -b.x))
__device__ __forceinline__ cf cadd_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf csub_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// m + k (-i) d  and  m - k (-i) d  for a real constant k (both halves of kk hold it)
__device__ __forceinline__ cf cfma_mi(cf m, cf kk, cf d) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(d), "s"(kk), "v"(m));
    return r;
}
__device__ __forceinline__ cf cfms_mi(cf m, cf kk, cf d) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(d), "s"(kk), "v"(m));
    return r;
}

static __device__ const float kCos7[7] = {1.f, 0.62348980185873353053f, -0.22252093395631440429f,
                                          -0.90096886790241912624f, -0.90096886790241912624f,
                                          -0.22252093395631440429f, 0.62348980185873353053f};
static __device__ const float kSin7[7] = {0.f, 0.78183148246802980871f, 0.97492791218182360702f,
                                          0.43388373911755812048f, -0.43388373911755812048f,
                                          -0.97492791218182360702f, -0.78183148246802980871f};

// forward DFT of R points held in registers: v[k] <- sum_q v[q] exp(-2 pi i q k / R)
template <int R> __device__ __forceinline__ void dft_small(cf (&v)[R]);

template <> __device__ __forceinline__ void dft_small<2>(cf (&v)[2]) {
    const cf a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
}
template <> __device__ __forceinline__ void dft_small<4>(cf (&v)[4]) {
    const cf s02 = v[0] + v[2], d02 = v[0] - v[2];
    const cf s13 = v[1] + v[3], d13 = v[1] - v[3];
    v[0] = s02 + s13;
    v[1] = cadd_mi(d02, d13);
    v[2] = s02 - s13;
    v[3] = csub_mi(d02, d13);
}
// 3 and 5 points in the Rader/Winograd form (sums and differences of the mirrored pairs first): 6 and 19 packed
// operations -- radices 3, 5, 6, 9, 10, 12, 15 are built on them
// (Every multiply-add below is an EXPLICIT fused one: left to -ffp-contract the compiler chooses which product of a sum
// of two products is fused, and the choice depends on the code around the butterfly -- the same source gave results one unit
// in the last place apart in two kernels.  Written out, a line gets the same bits from every kernel that transforms it.)
__device__ __forceinline__ cf cfma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cf cmul_only(cf a, cf b) {           // a product that stays a product (never fused into its consumer)
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <> __device__ __forceinline__ void dft_small<3>(cf (&v)[3]) {
    const cf t = v[1] + v[2], d = v[1] - v[2];
    const cf m = cfma((cf){-0.5f, -0.5f}, t, v[0]);
    const float s = 0.86602540378443864676f;
    v[0] = v[0] + t;
    v[1] = cfma_mi(m, (cf){s, s}, d);                        // m - i s d
    v[2] = cfms_mi(m, (cf){s, s}, d);
}
template <> __device__ __forceinline__ void dft_small<5>(cf (&v)[5]) {
    const cf t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
    const cf t5 = t1 + t2;
    const float c = 0.55901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const cf m1 = cfma((cf){-0.25f, -0.25f}, t5, v[0]);
    const cf d12 = t1 - t2;
    const cf a1 = cfma((cf){c, c}, d12, m1), a2 = cfma((cf){-c, -c}, d12, m1);
    const cf b1 = cfma((cf){s1, s1}, t3, cmul_only((cf){s2, s2}, t4));
    const cf b2 = cfma((cf){s2, s2}, t3, -cmul_only((cf){s1, s1}, t4));
    v[0] = v[0] + t5;
    v[1] = cadd_mi(a1, b1);                                  // a1 - i b1
    v[4] = csub_mi(a1, b1);
    v[2] = cadd_mi(a2, b2);
    v[3] = csub_mi(a2, b2);
}
template <> __device__ __forceinline__ void dft_small<7>(cf (&v)[7]) {        // matrix form (rare)
    cf o[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        cf acc = v[0];
#pragma unroll
        for (int q = 1; q < 7; ++q) {
            const int m = (q * k) % 7;
            const float c = kCos7[m], s = -kSin7[m];         // exp(-i phi)
            acc.x += v[q].x * c - v[q].y * s;
            acc.y += v[q].x * s + v[q].y * c;
        }
        o[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = o[k];
}

// |cos(t) gx - sin(t) gy| of the directional maxima (blur_estimation.py:129-133) with its roundings written out (as above: the
// second product rounded, the first fused into the difference), so that every kernel that folds a sample gets the same bits
__device__ __forceinline__ float dir_abs(float cs, float sn, float dx, float dy) { return fabsf(fmaf(cs, dx, -__fmul_rn(sn, dy))); }

// ---- composite radices, evaluated in registers as A x B Cooley-Tukey with constant twiddles -------------
// X[k1 + A k2] = sum_{n2} W_N^{n2 k1} ( sum_{n1} x[n1 B + n2] W_A^{n1 k1} ) W_B^{n2 k2},  N = A B
static __device__ const float kCos6[6] = {1.0f, 0.5f, -0.5f, -1.0f, -0.5f, 0.5f};
static __device__ const float kSin6[6] = {0.0f, 0.8660254037844386f, 0.8660254037844387f, 0.0f, -0.8660254037844384f, -0.8660254037844386f};
static __device__ const float kCos8[8] = {1.0f, 0.7071067811865476f, 0.0f, -0.7071067811865475f, -1.0f, -0.7071067811865477f, 0.0f, 0.7071067811865474f};
static __device__ const float kSin8[8] = {0.0f, 0.7071067811865475f, 1.0f, 0.7071067811865476f, 0.0f, -0.7071067811865475f, -1.0f, -0.7071067811865477f};
static __device__ const float kCos9[9] = {1.0f, 0.766044443118978f, 0.17364817766693041f, -0.5f, -0.9396926207859083f, -0.9396926207859084f, -0.5f, 0.17364817766692997f, 0.7660444431189778f};
static __device__ const float kSin9[9] = {0.0f, 0.6427876096865393f, 0.984807753012208f, 0.8660254037844387f, 0.3420201433256689f, -0.34202014332566866f, -0.8660254037844384f, -0.9848077530122081f, -0.6427876096865396f};
static __device__ const float kCos10[10] = {1.0f, 0.8090169943749475f, 0.30901699437494745f, -0.30901699437494734f, -0.8090169943749473f, -1.0f, -0.8090169943749476f, -0.30901699437494756f, 0.30901699437494723f, 0.8090169943749473f};
static __device__ const float kSin10[10] = {0.0f, 0.5877852522924731f, 0.9510565162951535f, 0.9510565162951536f, 0.5877852522924732f, 0.0f, -0.587785252292473f, -0.9510565162951535f, -0.9510565162951536f, -0.5877852522924734f};
static __device__ const float kCos12[12] = {1.0f, 0.8660254037844387f, 0.5f, 0.0f, -0.5f, -0.8660254037844387f, -1.0f, -0.8660254037844388f, -0.5f, 0.0f, 0.5f, 0.8660254037844384f};
static __device__ const float kSin12[12] = {0.0f, 0.5f, 0.8660254037844386f, 1.0f, 0.8660254037844387f, 0.5f, 0.0f, -0.5f, -0.8660254037844384f, -1.0f, -0.8660254037844386f, -0.5f};
static __device__ const float kCos15[15] = {1.0f, 0.9135454576426009f, 0.6691306063588582f, 0.30901699437494745f, -0.10452846326765333f, -0.5f, -0.8090169943749473f, -0.9781476007338057f, -0.9781476007338057f, -0.8090169943749476f, -0.5f, -0.10452846326765423f, 0.30901699437494723f, 0.6691306063588585f, 0.913545457642601f};
static __device__ const float kSin15[15] = {0.0f, 0.40673664307580015f, 0.7431448254773941f, 0.9510565162951535f, 0.9945218953682734f, 0.8660254037844387f, 0.5877852522924732f, 0.20791169081775931f, -0.20791169081775907f, -0.587785252292473f, -0.8660254037844384f, -0.9945218953682733f, -0.9510565162951536f, -0.743144825477394f, -0.40673664307580015f};
static __device__ const float kCos16[16] = {1.0f, 0.9238795325112867f, 0.7071067811865476f, 0.38268343236508984f, 0.0f, -0.3826834323650897f, -0.7071067811865475f, -0.9238795325112867f, -1.0f, -0.9238795325112868f, -0.7071067811865477f, -0.38268343236509034f, 0.0f, 0.38268343236509f, 0.7071067811865474f, 0.9238795325112865f};
static __device__ const float kSin16[16] = {0.0f, 0.3826834323650898f, 0.7071067811865475f, 0.9238795325112867f, 1.0f, 0.9238795325112867f, 0.7071067811865476f, 0.3826834323650899f, 0.0f, -0.38268343236508967f, -0.7071067811865475f, -0.9238795325112865f, -1.0f, -0.9238795325112866f, -0.7071067811865477f, -0.3826834323650904f};
// (radices 18 / 20 / 24: lines of 4320 or 7680 samples in three stages instead of four -- 8K images)
static __device__ const float kCos18[18] = {1.0f, 0.9396926207859084f, 0.766044443118978f, 0.5000000000000001f, 0.17364817766693041f, -0.1736481776669303f, -0.4999999999999998f, -0.7660444431189779f, -0.9396926207859083f, -1.0f, -0.9396926207859084f, -0.7660444431189783f, -0.5000000000000004f, -0.17364817766693033f, 0.17364817766692997f, 0.49999999999999933f, 0.7660444431189778f, 0.9396926207859084f};
static __device__ const float kSin18[18] = {0.0f, 0.3420201433256687f, 0.6427876096865393f, 0.8660254037844386f, 0.984807753012208f, 0.984807753012208f, 0.8660254037844387f, 0.6427876096865395f, 0.3420201433256689f, 0.0f, -0.34202014332566866f, -0.6427876096865389f, -0.8660254037844384f, -0.984807753012208f, -0.9848077530122081f, -0.866025403784439f, -0.6427876096865396f, -0.3420201433256686f};
static __device__ const float kCos20[20] = {1.0f, 0.9510565162951535f, 0.8090169943749475f, 0.5877852522924731f, 0.30901699437494745f, 0.0f, -0.30901699437494734f, -0.587785252292473f, -0.8090169943749473f, -0.9510565162951535f, -1.0f, -0.9510565162951538f, -0.8090169943749476f, -0.5877852522924732f, -0.30901699437494756f, 0.0f, 0.30901699437494723f, 0.5877852522924729f, 0.8090169943749473f, 0.9510565162951535f};
static __device__ const float kSin20[20] = {0.0f, 0.3090169943749474f, 0.5877852522924731f, 0.8090169943749475f, 0.9510565162951535f, 1.0f, 0.9510565162951536f, 0.8090169943749475f, 0.5877852522924732f, 0.3090169943749475f, 0.0f, -0.3090169943749469f, -0.587785252292473f, -0.8090169943749473f, -0.9510565162951535f, -1.0f, -0.9510565162951536f, -0.8090169943749476f, -0.5877852522924734f, -0.3090169943749476f};
static __device__ const float kCos24[24] = {1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.5000000000000001f, 0.25881904510252074f, 0.0f, -0.25881904510252063f, -0.4999999999999998f, -0.7071067811865475f, -0.8660254037844387f, -0.9659258262890682f, -1.0f, -0.9659258262890683f, -0.8660254037844388f, -0.7071067811865479f, -0.5000000000000004f, -0.25881904510252063f, 0.0f, 0.2588190451025203f, 0.5000000000000001f, 0.7071067811865474f, 0.8660254037844384f, 0.9659258262890681f};
static __device__ const float kSin24[24] = {0.0f, 0.25881904510252074f, 0.49999999999999994f, 0.7071067811865475f, 0.8660254037844386f, 0.9659258262890683f, 1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.49999999999999994f, 0.258819045102521f, 0.0f, -0.2588190451025208f, -0.4999999999999997f, -0.7071067811865471f, -0.8660254037844384f, -0.9659258262890683f, -1.0f, -0.9659258262890684f, -0.8660254037844386f, -0.7071067811865477f, -0.5000000000000004f, -0.25881904510252157f};
template <int A, int B> __device__ __forceinline__ void dft_ct(cf (&v)[A * B], const float *cs, const float *sn) {
    cf y[A * B];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        cf col[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) col[n1] = v[n1 * B + n2];
        dft_small<A>(col);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            const int m = (n2 * k1) % (A * B);
            y[k1 * B + n2] = (m == 0) ? col[k1] : cmulc(col[k1], (cf){cs[m], -sn[m]});
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        cf row[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) row[n2] = y[k1 * B + n2];
        dft_small<B>(row);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = row[k2];
    }
}
template <> __device__ __forceinline__ void dft_small<6>(cf (&v)[6]) { dft_ct<2, 3>(v, kCos6, kSin6); }
template <> __device__ __forceinline__ void dft_small<8>(cf (&v)[8]) { dft_ct<4, 2>(v, kCos8, kSin8); }
template <> __device__ __forceinline__ void dft_small<9>(cf (&v)[9]) { dft_ct<3, 3>(v, kCos9, kSin9); }
template <> __device__ __forceinline__ void dft_small<10>(cf (&v)[10]) { dft_ct<2, 5>(v, kCos10, kSin10); }
template <> __device__ __forceinline__ void dft_small<12>(cf (&v)[12]) { dft_ct<4, 3>(v, kCos12, kSin12); }
template <> __device__ __forceinline__ void dft_small<15>(cf (&v)[15]) { dft_ct<3, 5>(v, kCos15, kSin15); }
template <> __device__ __forceinline__ void dft_small<16>(cf (&v)[16]) { dft_ct<4, 4>(v, kCos16, kSin16); }
template <> __device__ __forceinline__ void dft_small<18>(cf (&v)[18]) { dft_ct<2, 9>(v, kCos18, kSin18); }
template <> __device__ __forceinline__ void dft_small<20>(cf (&v)[20]) { dft_ct<4, 5>(v, kCos20, kSin20); }
template <> __device__ __forceinline__ void dft_small<24>(cf (&v)[24]) { dft_ct<4, 6>(v, kCos24, kSin24); }

// t / m and t % m for 0 <= t < 2^23 with a float reciprocal and a one-step fix-up
__device__ __forceinline__ void divmod(int t, int m, float inv_m, int &q, int &r) {
    q = (int)((float)t * inv_m);
    r = t - q * m;
    if (r < 0) { r += m; --q; }
    else if (r >= m) { r -= m; ++q; }
}

// The R - 1 twiddles W^q of one butterfly, W = tw[m]: the powers 1, 2, 4, 8 (, 16) come from the table (exact), the others
// are products of two of them or of an earlier product (at most three roundings) -- four gathers instead of fifteen,
// and no integer multiply per gather.
// (in two halves, so that a caller can request the table's values long before it needs the products: twiddle_fetch +
// twiddle_fill == twiddle_powers, the same operations on the same operands)
template <int R>
__device__ __forceinline__ void twiddle_fetch(cf (&w)[R], const float2 *__restrict__ tw, int m) {
#pragma unroll
    for (int q = 1; q < R; ++q) {
        if ((q & (q - 1)) == 0) w[q] = reinterpret_cast<const cf *>(tw)[m * q];
    }
}
template <int R>
__device__ __forceinline__ void twiddle_fill(cf (&w)[R]) {
#pragma unroll
    for (int q = 3; q < R; ++q) {
        if ((q & (q - 1)) != 0) {
            const int hi = q >= 16 ? 16 : (q >= 8 ? 8 : (q >= 4 ? 4 : 2));
            w[q] = cmul(w[hi], w[q - hi]);
        }
    }
}
template <int R>
__device__ __forceinline__ void twiddle_powers(cf (&w)[R], const float2 *__restrict__ tw, int m) {
    twiddle_fetch<R>(w, tw, m);
    twiddle_fill<R>(w);
}

// One in-place stage on all NB interleaved lines.  L = current block length (a multiple of R).
// DIT == false:  butterfly, then twiddle W_L^{n' k}          (decimation in frequency)
// DIT == true :  twiddle W_L^{n' k}, then butterfly           (its transpose)
// tw[m] = exp(-2 pi i m / N); tw_step = N / L.
template <int R, bool DIT>
__device__ __forceinline__ void stage(float2 *s, int N, int lognb, int L, const float2 *__restrict__ tw) {
    const int M = L / R;
    const float inv_m = 1.0f / (float)M;
    const int tw_step = N / L;
    const int nb = 1 << lognb;
    const int work = (N / R) << lognb;
    for (int w = threadIdx.x; w < work; w += (int)blockDim.x) {
        const int j = w & (nb - 1);
        const int t = w >> lognb;
        int blk, np;
        divmod(t, M, inv_m, blk, np);
        cf *base = reinterpret_cast<cf *>(s) + (((long)blk * L + np) << lognb) + j;
        const int stride = M << lognb;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = base[q * stride];
        // M == 1 (the innermost stage): np == 0, every twiddle is 1 -- nothing to fetch or multiply
        if (DIT) {
            if (M > 1) {
                cf wq[R];
                twiddle_powers<R>(wq, tw, np * tw_step);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], wq[q]);
            }
            dft_small<R>(v);
        } else {
            dft_small<R>(v);
            if (M > 1) {
                cf wq[R];
                twiddle_powers<R>(wq, tw, np * tw_step);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], wq[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) base[q * stride] = v[q];
    }
}

// MAXR: the largest radix compiled in (a kernel's register allocation is that of its widest butterfly, taken or not: the
// radices above 16 are only in the variants that run the plans holding them)
template <bool DIT, int MAXR = 16>
__device__ __forceinline__ void stage_any(float2 *s, int N, int lognb, int L, int radix, const float2 *tw) {
    switch (radix) {
        case 24: if constexpr (MAXR >= 24) stage<24, DIT>(s, N, lognb, L, tw); break;
        case 20: if constexpr (MAXR >= 24) stage<20, DIT>(s, N, lognb, L, tw); break;
        case 18: if constexpr (MAXR >= 18) stage<18, DIT>(s, N, lognb, L, tw); break;
        case 16: stage<16, DIT>(s, N, lognb, L, tw); break;
        case 15: stage<15, DIT>(s, N, lognb, L, tw); break;
        case 12: stage<12, DIT>(s, N, lognb, L, tw); break;
        case 10: stage<10, DIT>(s, N, lognb, L, tw); break;
        case 9: stage<9, DIT>(s, N, lognb, L, tw); break;
        case 8: stage<8, DIT>(s, N, lognb, L, tw); break;
        case 6: stage<6, DIT>(s, N, lognb, L, tw); break;
        case 4: stage<4, DIT>(s, N, lognb, L, tw); break;
        case 2: stage<2, DIT>(s, N, lognb, L, tw); break;
        case 3: stage<3, DIT>(s, N, lognb, L, tw); break;
        case 5: stage<5, DIT>(s, N, lognb, L, tw); break;
        default: stage<7, DIT>(s, N, lognb, L, tw); break;
    }
}

struct DevPlan {
    int n;            // transform length of the mixed-radix core (== line length unless bluestein)
    int nstage;
    int radix[24];
    const float2 *tw;
    const float *drev;        // derivative multiplier / n, digit-reversed order (direct plans)
    // bluestein
    int line_n;               // the real line length (== n for direct plans)
    const float2 *chirp;      // exp(+i pi k^2 / line_n)
    const float2 *bfilt_rev;  // FFT_n(chirp filter) / n in digit-reversed order
    const float *dnat;        // derivative multiplier / line_n in natural order
};

// natural -> digit-reversed forward DFT of length plan.n (all NB lines)
__device__ __forceinline__ void forward_dif(float2 *s, const DevPlan &p, int lognb) {
    int L = p.n;
    for (int i = 0; i < p.nstage; ++i) {
        stage_any<false>(s, p.n, lognb, L, p.radix[i], p.tw);
        L /= p.radix[i];
        __syncthreads();
    }
}
// digit-reversed -> natural forward DFT of length plan.n
__device__ __forceinline__ void forward_dit(float2 *s, const DevPlan &p, int lognb) {
    int L = 1;
    for (int i = p.nstage - 1; i >= 0; --i) {
        L *= p.radix[i];
        stage_any<true>(s, p.n, lognb, L, p.radix[i], p.tw);
        __syncthreads();
    }
}

// ---- the same transform with three of its LDS round trips removed ----------------------------------------------------
// The first DIF stage takes its inputs from `io.load` (global memory) instead of LDS; the last DIF stage, the derivative
// multiplier and the first DIT stage touch the same R points and run back to back in registers; the last DIT stage hands
// its outputs to `io.store` instead of writing them to LDS.  Needs a direct plan with at least two stages.
//   io.load(p, j)            -> the two real samples (line 2j, line 2j+1) at position p, as one complex value
//   io.prefetch(p, j)        -> whatever io.store wants fetched before the butterfly (its loads overlap the arithmetic)
//   io.store(p, j, v, pre)   -> v = conj(d line_2j/dn + i d line_2j+1/dn) at position p
template <int R, class IO>
__device__ __forceinline__ void first_stage(float2 *s, int N, int lognb, const float2 *__restrict__ tw, IO &io) {
    const int M = N / R;
    const int nb = 1 << lognb;
    const int stride = M << lognb;
    for (int w = threadIdx.x; w < stride; w += (int)blockDim.x) {
        const int j = w & (nb - 1), np = w >> lognb;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = to_cf(io.load(np + q * M, j));
        dft_small<R>(v);
        cf wq[R];
        twiddle_powers<R>(wq, tw, np);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], wq[q]);
        cf *base = reinterpret_cast<cf *>(s) + w;
#pragma unroll
        for (int q = 0; q < R; ++q) base[q * stride] = v[q];
    }
}

template <int R>
__device__ __forceinline__ void centre_stage(float2 *s, int N, int lognb, const float *__restrict__ drev) {
    const int nb = 1 << lognb;
    const int work = (N / R) << lognb;
    for (int w = threadIdx.x; w < work; w += (int)blockDim.x) {
        const int j = w & (nb - 1), t = w >> lognb;
        cf *base = reinterpret_cast<cf *>(s) + ((t * R) << lognb) + j;
        cf v[R];
        float d[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { v[q] = base[q << lognb]; d[q] = drev[t * R + q]; }
        dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = (cf){-d[q] * v[q].y, -d[q] * v[q].x};                // conj(i d z)
        dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) base[q << lognb] = v[q];
    }
}

template <int R, class IO>
__device__ __forceinline__ void last_stage(const float2 *s, int N, int lognb, const float2 *__restrict__ tw, IO &io) {
    const int M = N / R;
    const int nb = 1 << lognb;
    const int stride = M << lognb;
    for (int w = threadIdx.x; w < stride; w += (int)blockDim.x) {
        const int j = w & (nb - 1), np = w >> lognb;
        typename IO::Pre pre[R];
#pragma unroll
        for (int q = 0; q < R; ++q) pre[q] = io.prefetch(np + q * M, j);
        cf v[R];
        const cf *base = reinterpret_cast<const cf *>(s) + w;
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = base[q * stride];
        cf wq[R];
        twiddle_powers<R>(wq, tw, np);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], wq[q]);
        dft_small<R>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) io.store(np + q * M, j, to_f2(v[q]), pre[q]);
    }
}

#define PB_FFT_RADIX_SWITCH(radix, CALL)                                                                       \
    switch (radix) {                                                                                           \
        case 24: if constexpr (MAXR >= 24) { CALL(24); } break;                                                \
        case 20: if constexpr (MAXR >= 24) { CALL(20); } break;                                                \
        case 18: if constexpr (MAXR >= 18) { CALL(18); } break;                                                \
        case 16: CALL(16); break;                                                                              \
        case 15: CALL(15); break;                                                                              \
        case 12: CALL(12); break;                                                                              \
        case 10: CALL(10); break;                                                                              \
        case 9: CALL(9); break;                                                                                \
        case 8: CALL(8); break;                                                                                \
        case 6: CALL(6); break;                                                                                \
        case 4: CALL(4); break;                                                                                \
        case 2: CALL(2); break;                                                                                \
        case 3: CALL(3); break;                                                                                \
        case 5: CALL(5); break;                                                                                \
        default: CALL(7); break;                                                                               \
    }

__device__ __forceinline__ bool fused_plan(const DevPlan &p) { return p.line_n == p.n && p.nstage >= 2; }

// Must be called by all threads of the workgroup; s needs no initialisation and holds nothing of interest afterwards.
template <int MAXR = 16, class IO>
__device__ __forceinline__ void spectral_derivative_fused(float2 *s, const DevPlan &p, int lognb, IO &io) {
    const int last = p.nstage - 1;
    PB_FT(0);
#define PB_FIRST(R) first_stage<R>(s, p.n, lognb, p.tw, io)
    PB_FFT_RADIX_SWITCH(p.radix[0], PB_FIRST)
#undef PB_FIRST
    PB_FT(1);
    __syncthreads();
    PB_FT(2);
    int L = p.n / p.radix[0];
    for (int i = 1; i < last; ++i) {
        stage_any<false, MAXR>(s, p.n, lognb, L, p.radix[i], p.tw);
        L /= p.radix[i];
        __syncthreads();
    }
    PB_FT(3);
#define PB_CENTRE(R) centre_stage<R>(s, p.n, lognb, p.drev)
    PB_FFT_RADIX_SWITCH(p.radix[last], PB_CENTRE)
#undef PB_CENTRE
    __syncthreads();
    PB_FT(4);
    for (int i = last - 1; i >= 1; --i) {
        L *= p.radix[i];
        stage_any<true, MAXR>(s, p.n, lognb, L, p.radix[i], p.tw);
        __syncthreads();
    }
    PB_FT(5);
#define PB_LAST(R) last_stage<R>(s, p.n, lognb, p.tw, io)
    PB_FFT_RADIX_SWITCH(p.radix[0], PB_LAST)
#undef PB_LAST
    PB_FT(6);
}

// s holds NB interleaved complex lines of length plan.line_n in natural order (for bluestein
// plans the buffer must have room for plan.n entries per line).  On return s[p] = conj of the
// complex line (da/dn + i db/dn): the derivative of the real part is s.x, of the imaginary
// part is -s.y.  Must be called by all threads of the workgroup; ends with a barrier.
__device__ __forceinline__ void spectral_derivative(float2 *s, const DevPlan &p, int lognb) {
    const int nb = 1 << lognb;
    if (p.line_n == p.n) {
        forward_dif(s, p, lognb);
        for (int e = threadIdx.x; e < (p.n << lognb); e += (int)blockDim.x) {
            const float d = p.drev[e >> lognb];
            const float2 z = s[e];
            s[e] = make_float2(-d * z.y, -d * z.x);        // conj(i d z)
        }
        __syncthreads();
        forward_dit(s, p, lognb);
        return;
    }
    // ---- Bluestein: X[k] = conj(w[k]) * sum_n (x[n] conj(w[n])) w[k-n],  w[n] = exp(i pi n^2/N)
    const int N = p.line_n, M = p.n;
    for (int pass = 0; pass < 2; ++pass) {
        // a[n] = x[n] * conj(w[n]), zero padded to M
        for (int e = threadIdx.x; e < (M << lognb); e += (int)blockDim.x) {
            const int n = e >> lognb;
            float2 v = make_float2(0.f, 0.f);
            if (n < N) {
                const float2 w = p.chirp[n];
                v = cmul(s[e], make_float2(w.x, -w.y));
            }
            s[e] = v;
        }
        __syncthreads();
        forward_dif(s, p, lognb);
        // multiply by the filter spectrum (already / M), conj for the inverse transform
        for (int e = threadIdx.x; e < (M << lognb); e += (int)blockDim.x) {
            const float2 v = cmul(s[e], p.bfilt_rev[e >> lognb]);
            s[e] = make_float2(v.x, -v.y);
        }
        __syncthreads();
        forward_dit(s, p, lognb);
        // c[k] = conj(s[k]);  X[k] = conj(w[k]) c[k]
        for (int e = threadIdx.x; e < (M << lognb); e += (int)blockDim.x) {
            const int k = e >> lognb;
            if (k < N) {
                const float2 w = p.chirp[k];
                float2 X = cmul(make_float2(s[e].x, -s[e].y), make_float2(w.x, -w.y));
                if (pass == 0) {
                    // Y = i d X; the second pass computes DFT(conj(Y)) whose conj is N * ifft(Y)
                    const float d = p.dnat[k];
                    X = make_float2(-d * X.y, -d * X.x);   // conj(i d X)
                }
                s[e] = X;
            }
        }
        __syncthreads();
    }
    (void)nb;
}

}  // namespace pbfft
