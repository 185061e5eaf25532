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
// Lengths whose prime factors are all <= 7 use radices 4/2/3/5/7 directly; any other length
// goes through Bluestein's chirp-z with a power-of-two inner length (plan.bluestein_m).
#pragma once
#include "common.h"

namespace pbfft {

constexpr int NT = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward quarter turn)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

static __device__ const float kCos3[3] = {1.f, -0.5f, -0.5f};
static __device__ const float kSin3[3] = {0.f, 0.86602540378443864676f, -0.86602540378443864676f};
static __device__ const float kCos5[5] = {1.f, 0.30901699437494742410f, -0.80901699437494742410f,
                                          -0.80901699437494742410f, 0.30901699437494742410f};
static __device__ const float kSin5[5] = {0.f, 0.95105651629515357212f, 0.58778525229247312917f,
                                          -0.58778525229247312917f, -0.95105651629515357212f};
static __device__ const float kCos7[7] = {1.f, 0.62348980185873353053f, -0.22252093395631440429f,
                                          -0.90096886790241912624f, -0.90096886790241912624f,
                                          -0.22252093395631440429f, 0.62348980185873353053f};
static __device__ const float kSin7[7] = {0.f, 0.78183148246802980871f, 0.97492791218182360702f,
                                          0.43388373911755812048f, -0.43388373911755812048f,
                                          -0.97492791218182360702f, -0.78183148246802980871f};

// forward DFT of R points held in registers: v[k] <- sum_q v[q] exp(-2 pi i q k / R)
template <int R> __device__ __forceinline__ void dft_small(float2 (&v)[R]);

template <> __device__ __forceinline__ void dft_small<2>(float2 (&v)[2]) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <> __device__ __forceinline__ void dft_small<4>(float2 (&v)[4]) {
    const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const float2 s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(s02, s13);
    v[1] = cadd(d02, d13);
    v[2] = csub(s02, s13);
    v[3] = csub(d02, d13);
}
template <int R> __device__ __forceinline__ void dft_odd(float2 (&v)[R], const float *cs, const float *sn) {
    float2 o[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        float2 acc = v[0];
#pragma unroll
        for (int q = 1; q < R; ++q) {
            const int m = (q * k) % R;
            const float c = cs[m], s = -sn[m];      // exp(-i phi)
            acc.x += v[q].x * c - v[q].y * s;
            acc.y += v[q].x * s + v[q].y * c;
        }
        o[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = o[k];
}
template <> __device__ __forceinline__ void dft_small<3>(float2 (&v)[3]) { dft_odd<3>(v, kCos3, kSin3); }
template <> __device__ __forceinline__ void dft_small<5>(float2 (&v)[5]) { dft_odd<5>(v, kCos5, kSin5); }
template <> __device__ __forceinline__ void dft_small<7>(float2 (&v)[7]) { dft_odd<7>(v, kCos7, kSin7); }

// t / m and t % m for 0 <= t < 2^23 with a float reciprocal and a one-step fix-up
__device__ __forceinline__ void divmod(int t, int m, float inv_m, int &q, int &r) {
    q = (int)((float)t * inv_m);
    r = t - q * m;
    if (r < 0) { r += m; --q; }
    else if (r >= m) { r -= m; ++q; }
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
    for (int w = threadIdx.x; w < work; w += NT) {
        const int j = w & (nb - 1);
        const int t = w >> lognb;
        int blk, np;
        divmod(t, M, inv_m, blk, np);
        float2 *base = s + (((long)blk * L + np) << lognb) + j;
        const int stride = M << lognb;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = base[q * stride];
        if (DIT) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul(v[q], tw[np * q * tw_step]);
            dft_small<R>(v);
        } else {
            dft_small<R>(v);
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul(v[q], tw[np * q * tw_step]);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) base[q * stride] = v[q];
    }
}

template <bool DIT>
__device__ __forceinline__ void stage_any(float2 *s, int N, int lognb, int L, int radix, const float2 *tw) {
    switch (radix) {
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

// s holds NB interleaved complex lines of length plan.line_n in natural order (for bluestein
// plans the buffer must have room for plan.n entries per line).  On return s[p] = conj of the
// complex line (da/dn + i db/dn): the derivative of the real part is s.x, of the imaginary
// part is -s.y.  Must be called by all NT threads; ends with a barrier.
__device__ __forceinline__ void spectral_derivative(float2 *s, const DevPlan &p, int lognb) {
    const int nb = 1 << lognb;
    if (p.line_n == p.n) {
        forward_dif(s, p, lognb);
        for (int e = threadIdx.x; e < (p.n << lognb); e += NT) {
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
        for (int e = threadIdx.x; e < (M << lognb); e += NT) {
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
        for (int e = threadIdx.x; e < (M << lognb); e += NT) {
            const float2 v = cmul(s[e], p.bfilt_rev[e >> lognb]);
            s[e] = make_float2(v.x, -v.y);
        }
        __syncthreads();
        forward_dit(s, p, lognb);
        // c[k] = conj(s[k]);  X[k] = conj(w[k]) c[k]
        for (int e = threadIdx.x; e < (M << lognb); e += NT) {
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
