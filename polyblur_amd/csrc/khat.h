// The spectrum of an image's kernel and the choice of body for its reblurring passes (shared by conv_fft.hip, whose
// khat_kernel serves records the host built, and estimate.hip, whose parameter kernel ends with it: one launch less on
// the critical path of every iteration).
#pragma once
#include "common.h"
#ifndef PB_PT
#define PB_PT(i)
#endif

namespace {

constexpr int KH_FT_N = 64, KH_THREADS = 256;
// cos / sin (2 pi m / 64) in double, for the spectra
static __device__ const double kCos64[64] = {
    1, 0.99518472667219693, 0.98078528040323043, 0.95694033573220882,
    0.92387953251128674, 0.88192126434835505, 0.83146961230254524, 0.77301045336273699,
    0.70710678118654757, 0.63439328416364549, 0.55557023301960229, 0.47139673682599781,
    0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077,
    0, -0.098017140329560645, -0.19509032201612819, -0.29028467725446216,
    -0.38268343236508973, -0.4713967368259977, -0.55557023301960196, -0.63439328416364538,
    -0.70710678118654746, -0.77301045336273699, -0.83146961230254535, -0.88192126434835494,
    -0.92387953251128674, -0.95694033573220882, -0.98078528040323043, -0.99518472667219682,
    -1, -0.99518472667219693, -0.98078528040323043, -0.95694033573220894,
    -0.92387953251128685, -0.88192126434835505, -0.83146961230254546, -0.7730104533627371,
    -0.70710678118654768, -0.63439328416364593, -0.55557023301960218, -0.47139673682599786,
    -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.098017140329560451,
    0, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205,
    0.38268343236509, 0.47139673682599759, 0.55557023301960184, 0.6343932841636456,
    0.70710678118654735, 0.77301045336273666, 0.83146961230254524, 0.88192126434835483,
    0.92387953251128652, 0.95694033573220882, 0.98078528040323032, 0.99518472667219693
};
static __device__ const double kSin64[64] = {
    0, 0.098017140329560604, 0.19509032201612825, 0.29028467725446233,
    0.38268343236508978, 0.47139673682599764, 0.55557023301960218, 0.63439328416364549,
    0.70710678118654746, 0.77301045336273699, 0.83146961230254524, 0.88192126434835494,
    0.92387953251128674, 0.95694033573220894, 0.98078528040323043, 0.99518472667219682,
    1, 0.99518472667219693, 0.98078528040323043, 0.95694033573220894,
    0.92387953251128674, 0.88192126434835505, 0.83146961230254546, 0.7730104533627371,
    0.70710678118654757, 0.63439328416364549, 0.55557023301960218, 0.47139673682599786,
    0.38268343236508989, 0.29028467725446239, 0.19509032201612861, 0.098017140329560826,
    0, -0.09801714032956059, -0.19509032201612836, -0.29028467725446211,
    -0.38268343236508967, -0.47139673682599764, -0.55557023301960196, -0.63439328416364527,
    -0.70710678118654746, -0.77301045336273666, -0.83146961230254524, -0.88192126434835494,
    -0.92387953251128652, -0.95694033573220882, -0.98078528040323032, -0.99518472667219693,
    -1, -0.99518472667219693, -0.98078528040323043, -0.95694033573220894,
    -0.92387953251128663, -0.88192126434835505, -0.83146961230254546, -0.77301045336273688,
    -0.70710678118654768, -0.63439328416364593, -0.55557023301960218, -0.47139673682599792,
    -0.38268343236509039, -0.2902846772544625, -0.19509032201612872, -0.098017140329560506
};
constexpr int KH_SLICES = 16;          // workgroups per image: each forms the spectrum at 4 of the 64 (8 of the 128) x positions
constexpr int KH_SLICES_LEAN = 64;     // ... of the parameter kernel's short chain (estimate.hip): 1 of the 64 (2 of the 128) -- the last phase, one output per thread, was 10.6 k of that chain's 35.6 k cycles at 16 slices (tools/params_trace.py)

// What khat_body needs of a record, for a caller that has just formed it and still holds it in LDS (estimate.hip's
// parameter kernel): no wait for the record's stores to land, no read back.
struct RecLds { const float *taps; int radius, nph, separable; };

static __device__ const double kCos128[128] = {
    1.0, 0.9987954562051724, 0.9951847266721969, 0.989176509964781,
    0.9807852804032304, 0.970031253194544, 0.9569403357322088, 0.9415440651830208,
    0.9238795325112867, 0.9039892931234433, 0.881921264348355, 0.8577286100002721,
    0.8314696123025452, 0.8032075314806449, 0.773010453362737, 0.7409511253549591,
    0.7071067811865476, 0.6715589548470183, 0.6343932841636455, 0.5956993044924335,
    0.5555702330196023, 0.5141027441932217, 0.4713967368259978, 0.4275550934302822,
    0.38268343236508984, 0.33688985339222005, 0.29028467725446233, 0.24298017990326398,
    0.19509032201612833, 0.14673047445536175, 0.09801714032956077, 0.049067674327418126,
    6.123233995736766e-17, -0.04906767432741801, -0.09801714032956065, -0.14673047445536164,
    -0.1950903220161282, -0.24298017990326387, -0.29028467725446216, -0.33688985339221994,
    -0.3826834323650897, -0.42755509343028186, -0.4713967368259977, -0.5141027441932217,
    -0.555570233019602, -0.5956993044924334, -0.6343932841636454, -0.6715589548470184,
    -0.7071067811865475, -0.7409511253549589, -0.773010453362737, -0.8032075314806448,
    -0.8314696123025453, -0.857728610000272, -0.8819212643483549, -0.9039892931234433,
    -0.9238795325112867, -0.9415440651830207, -0.9569403357322088, -0.970031253194544,
    -0.9807852804032304, -0.989176509964781, -0.9951847266721968, -0.9987954562051724,
    -1.0, -0.9987954562051724, -0.9951847266721969, -0.989176509964781,
    -0.9807852804032304, -0.970031253194544, -0.9569403357322089, -0.9415440651830208,
    -0.9238795325112868, -0.9039892931234434, -0.881921264348355, -0.8577286100002721,
    -0.8314696123025455, -0.8032075314806449, -0.7730104533627371, -0.7409511253549591,
    -0.7071067811865477, -0.6715589548470187, -0.6343932841636459, -0.5956993044924331,
    -0.5555702330196022, -0.5141027441932218, -0.47139673682599786, -0.4275550934302825,
    -0.38268343236509034, -0.33688985339221994, -0.29028467725446244, -0.24298017990326412,
    -0.19509032201612866, -0.1467304744553623, -0.09801714032956045, -0.04906767432741803,
    -1.8369701987210297e-16, 0.04906767432741766, 0.09801714032956009, 0.14673047445536194,
    0.1950903220161283, 0.24298017990326376, 0.29028467725446205, 0.3368898533922196,
    0.38268343236509, 0.42755509343028214, 0.4713967368259976, 0.5141027441932216,
    0.5555702330196018, 0.5956993044924329, 0.6343932841636456, 0.6715589548470183,
    0.7071067811865474, 0.7409511253549589, 0.7730104533627367, 0.803207531480645,
    0.8314696123025452, 0.857728610000272, 0.8819212643483548, 0.9039892931234431,
    0.9238795325112865, 0.9415440651830208, 0.9569403357322088, 0.970031253194544,
    0.9807852804032303, 0.9891765099647809, 0.9951847266721969, 0.9987954562051724
};
static __device__ const double kSin128[128] = {
    0.0, 0.049067674327418015, 0.0980171403295606, 0.14673047445536175,
    0.19509032201612825, 0.24298017990326387, 0.29028467725446233, 0.33688985339222005,
    0.3826834323650898, 0.4275550934302821, 0.47139673682599764, 0.5141027441932217,
    0.5555702330196022, 0.5956993044924334, 0.6343932841636455, 0.6715589548470183,
    0.7071067811865475, 0.7409511253549591, 0.773010453362737, 0.8032075314806448,
    0.8314696123025452, 0.8577286100002721, 0.8819212643483549, 0.9039892931234433,
    0.9238795325112867, 0.9415440651830208, 0.9569403357322089, 0.970031253194544,
    0.9807852804032304, 0.989176509964781, 0.9951847266721968, 0.9987954562051724,
    1.0, 0.9987954562051724, 0.9951847266721969, 0.989176509964781,
    0.9807852804032304, 0.970031253194544, 0.9569403357322089, 0.9415440651830208,
    0.9238795325112867, 0.9039892931234434, 0.881921264348355, 0.8577286100002721,
    0.8314696123025455, 0.8032075314806449, 0.7730104533627371, 0.740951125354959,
    0.7071067811865476, 0.6715589548470186, 0.6343932841636455, 0.5956993044924335,
    0.5555702330196022, 0.5141027441932218, 0.47139673682599786, 0.42755509343028203,
    0.3826834323650899, 0.33688985339222033, 0.2902846772544624, 0.24298017990326407,
    0.1950903220161286, 0.1467304744553618, 0.09801714032956083, 0.049067674327417966,
    1.2246467991473532e-16, -0.049067674327417724, -0.09801714032956059, -0.14673047445536158,
    -0.19509032201612836, -0.24298017990326382, -0.2902846772544621, -0.3368898533922201,
    -0.38268343236508967, -0.4275550934302818, -0.47139673682599764, -0.5141027441932216,
    -0.555570233019602, -0.5956993044924332, -0.6343932841636453, -0.6715589548470184,
    -0.7071067811865475, -0.7409511253549589, -0.7730104533627367, -0.803207531480645,
    -0.8314696123025452, -0.857728610000272, -0.8819212643483549, -0.9039892931234431,
    -0.9238795325112865, -0.9415440651830208, -0.9569403357322088, -0.970031253194544,
    -0.9807852804032303, -0.9891765099647809, -0.9951847266721969, -0.9987954562051724,
    -1.0, -0.9987954562051724, -0.9951847266721969, -0.9891765099647809,
    -0.9807852804032304, -0.970031253194544, -0.9569403357322089, -0.9415440651830209,
    -0.9238795325112866, -0.9039892931234433, -0.881921264348355, -0.8577286100002722,
    -0.8314696123025455, -0.8032075314806453, -0.7730104533627369, -0.7409511253549591,
    -0.7071067811865477, -0.6715589548470187, -0.6343932841636459, -0.5956993044924332,
    -0.5555702330196022, -0.5141027441932219, -0.4713967368259979, -0.42755509343028253,
    -0.3826834323650904, -0.33688985339222, -0.2902846772544625, -0.24298017990326418,
    -0.19509032201612872, -0.1467304744553624, -0.0980171403295605, -0.04906767432741809
};

// The polynomial's spectrum on the 128 x 128 grid, in the order conv_w128.hip reads it: [wave w][register][lane], where in
// the row phase wave w, lane (j, hh) holds row slot 32 w + j -- the slot 2 g + h is the column transform's register g in
// lane half h, i.e. frequency fy = 2 K(g) + h with K(g) = (g >> 3) + 8 (g & 7) the 64-point transform's register order -- and
// register `reg` of lane half hh holds fx = 2 K(reg) + hh.  Slice s of KH_SLICES = 16 forms registers 4 s .. 4 s + 3 (8 fx
// values).  sk: the taps in LDS (zero outside the record's box); both transforms' 1/128 folded in.  Called by all KH_THREADS.
// c8 / s8: cos / sin (2 pi m / 128) in LDS, m = 0 .. 127 (the caller fills them -- early, so that their trip through memory
// is not on the chain -- and has passed a barrier since).
template <int SL>
__device__ __forceinline__ void khat128_body(const float *sk, float *out, int slice, const PolySpec ps, const double *c8, const double *s8) {
    constexpr int NU = PB_KRAD + 1, NX = 2 * 64 / SL, RPS = 64 / SL;      // fx values / registers per slice
    static_assert(RPS >= 1, "at most 64 slices");
    __shared__ double2 G8[NU * NX];
    const int tid = threadIdx.x;
    PB_PT(27);
    auto K64 = [](int g) { return (g >> 3) + 8 * (g & 7); };
    if (tid < NU * NX) {
        const int u = tid / NX + PB_KRAD, xi = tid % NX, reg = RPS * slice + (xi >> 1), fx = 2 * K64(reg) + (xi & 1);
        double ar = 0.0, ai = 0.0;
#pragma unroll 5
        for (int v = 0; v < PB_KSIZE; ++v) {
            const int m = (fx * (v - PB_KRAD)) & 127;
            const double k = (double)sk[u * PB_KSIZE + v];
            ar += k * c8[m]; ai += k * s8[m];
        }
        G8[tid] = make_double2(ar, ai);
    }
    __syncthreads();
    PB_PT(28);
    for (int idx = tid; idx < NX * 128; idx += KH_THREADS) {
        const int xi = idx >> 7, s = idx & 127, fy = 2 * K64(s >> 1) + (s & 1);
        double ar = 0.5 * G8[xi].x;
#pragma unroll 4
        for (int u = 1; u < NU; ++u) {
            const int m = (fy * u) & 127;
            const double2 g = G8[u * NX + xi];
            ar += g.x * c8[m] - g.y * s8[m];
        }
        double v = 2.0 * ar;                                              // the kernel's transform at (fx, fy): real
        v = (((double)ps.a3 * v + (double)ps.a2) * v + (double)ps.a1) * v + (double)ps.b;
        const int reg = RPS * slice + (xi >> 1), hh = xi & 1, w = s >> 5, j = s & 31;
        out[(w * 64 + reg) * 64 + j + 32 * hh] = (float)(v * (1.0 / 16384.0));
    }
}

#ifndef PB_HALO_TOL
#define PB_HALO_TOL 1e-8f
#endif
constexpr float KH_HALO_TOL = PB_HALO_TOL;     // tap mass (absolute values) a window halo may leave outside: see khat_body

// Smallest r with  sum_{|i - n| > r} m[i] < tol  for the 2n+1 non-negative values m[0 .. 2n] (n <= 63): one wave, lane l
// holds the pair of offsets +-(63 - l), a prefix sum over the lanes is the tail beyond each offset.
template <typename F> __device__ __forceinline__ int tail_radius(F m, int n, float tol, int lane) {
    const int o = 63 - lane;
    float v = (o >= 1 && o <= n) ? m(n - o) + m(n + o) : 0.f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    const unsigned long long b = __ballot(v >= tol);
    return b ? 63 - __builtin_ctzll(b) : 0;
}

// Called by all KH_THREADS threads of a workgroup: slice `slice` (0 .. KH_SLICES - 1) of the spectrum of `info`'s taps,
// and (slice 0) the image's choice of body.  The record may have been written by this same workgroup just before (global
// memory, then a barrier): it is read through the vector path.
// lean: the caller is a short-chain workgroup of the parameter kernel (estimate.hip): `strip` is the record workgroup's to write.
template <int SL = KH_SLICES>
__device__ __forceinline__ void khat_body(const pb_blur_info *info, float *out, pb_fft_sel *sel, int min_phases, int slice,
                                          const PolySpec ps = no_poly(), const RecLds *rl = nullptr, bool lean = false) {
    constexpr int NR = PB_KRAD + 1, PXS = KH_FT_N / SL;
    __shared__ double c8[128], s8[128];
    constexpr int K1 = PB_KSIZE, K2 = 2 * PB_KSIZE - 1, K3 = 3 * PB_KSIZE - 2;
    __shared__ double2 G[NR * PXS];
    __shared__ double cs[KH_FT_N], sn[KH_FT_N];
    __shared__ float sk[PB_KSIZE * PB_KSIZE];
    __shared__ float s_m[2][K1], s_mp[2][3 * K1 - 2], s_m2[2][K2 + 2 * (K1 - 1)], s_m3[2][K3];   // |taps| marginals over the x / y offsets (s_mp, s_m2: zero-padded) and their powers
    __shared__ int s_h[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nph = rl ? rl->nph : info->nphase[0] + info->nphase[1] + info->nphase[2];
    const int R = rl ? rl->radius : info->radius;
    const int separable = rl ? rl->separable : info->separable;
    if (tid < KH_FT_N) { cs[tid] = kCos64[tid]; sn[tid] = kSin64[tid]; }
    if (ps.on == 3 && tid >= 64 && tid < 192) { c8[tid - 64] = kCos128[tid - 64]; s8[tid - 64] = kSin128[tid - 64]; }   // (for khat128_body: requested here, met by the barriers below)
    bool sym = true;
    // (the short chain: the taps are the caller's LDS array as it is -- full support, nothing outside the box to mask -- and
    // point-symmetric by construction, PolySpec.always)
    const float *skp = lean ? rl->taps : sk;
    // (lean: the short chain does not look -- the record workgroup beside it does, off the critical path, and reports what it
    // finds in pb_fft_sel.pad_[1]: estimate.hip)
    if (!lean)
    for (int i = tid; i < PB_KSIZE * PB_KSIZE; i += KH_THREADS) {
        const int u = i / PB_KSIZE - PB_KRAD, v = i % PB_KSIZE - PB_KRAD;
        const bool in = abs(u) <= R && abs(v) <= R;
        const float k = rl ? rl->taps[i] : info->kernel[i];
        const float km = rl ? rl->taps[PB_KSIZE * PB_KSIZE - 1 - i] : info->kernel[PB_KSIZE * PB_KSIZE - 1 - i];
        sk[i] = in ? k : 0.f;
        sym = sym && (!in || k == km);
    }
    constexpr int PD = K1 - 1;                                       // zeros on either side of m (s_mp) and of m * m (s_m2), see below
    if (tid < 2 * (K1 + 2 * PD)) s_mp[tid / (K1 + 2 * PD)][tid % (K1 + 2 * PD)] = 0.f;
    if (tid < 2 * (K2 + 2 * PD)) s_m2[tid / (K2 + 2 * PD)][tid % (K2 + 2 * PD)] = 0.f;
    PB_PT(20);
    // (PolySpec.always: the host vouches for point-symmetric taps -- the estimation's own Gaussians -- and has no other launch
    // to fall back on; taps that compare unequal there are NaNs, which the one-pass form turns into the same zeros)
    // The reduction is kept under `always` too: what it finds is reported (pb_fft_sel.rf = -1, pb_body_selection's second
    // column), so that a caller -- or a debug run -- can tell a record the host wrongly vouched for from a good one.
    const bool sym_found = __syncthreads_and(sym) != 0;
    const bool symm = (sym_found || ps.always != 0) && min_phases >= 0;
    const bool vouched_wrongly = ps.always != 0 && !sym_found;
    // The window halo of the tile-spectrum body, per axis.  The spectrum below holds EVERY tap of the record's box; the halo
    // only has to cover the taps that matter to overlap-save: what lies beyond it wraps around inside the window, an error
    // of at most that mass times the range of the operand.  The record's radius counts taps until they underflow to zero
    // (full support) -- 8 for sigma 0.6 whose taps beyond +-4 carry 1e-13 of the mass --, so the halo along an axis is the
    // radius outside which the taps' absolute values, summed over the other axis, come to < KH_HALO_TOL = 1e-8: the aliasing
    // term is at most that mass times the operand's range -- for samples in [0, 1] below a sixth of fp32's unit roundoff
    // (6e-8), so it cannot move a result by more than a rounding does (rounds 3 - 5 asked for 1e-10: two samples more of halo
    // on either side of every window for a term nothing can see; any caller-supplied taps are measured the same way).  An
    // oblique Gaussian's marginals differ: sigma 1.66 / rho 1.0 at 66 degrees needs 6 samples along x and 9 along y.
    //   The one-pass polynomial's filter a3 K^3 + a2 K^2 + a1 K + b is measured the same way without being formed: the
    // marginal of |K * K| is at most the marginal of |K| convolved with itself, so |a3| m^3 + |a2| m^2 + |a1| m (1-D
    // convolution powers of the kernel's marginal m) bounds the composite's marginal from above -- 13 and 18 samples for the
    // kernel above where the box of 3 x 12 would say 36: Gaussian tails compound like sqrt(3), not like 3.
    // (fixed trip counts over zero-padded arrays, fully unrolled: every load of a sum is in flight at once -- with the
    // bounds of the overlap as loop limits these three steps took 10 k cycles of the kernel's 41 k)
    if (tid < K1) {
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < K1; ++u) a += fabsf(skp[u * K1 + tid]);
        s_m[0][tid] = a; s_mp[0][PD + tid] = a;
    } else if (tid >= 64 && tid < 64 + K1) {
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < K1; ++v) a += fabsf(skp[(tid - 64) * K1 + v]);
        s_m[1][tid - 64] = a; s_mp[1][PD + tid - 64] = a;
    }
    __syncthreads();
    PB_PT(24);
    if (ps.on) {
        const int ax = tid >> 7, i = tid & 127;
        if (i < K2) {                                               // (m * m)[i] = sum_j m[j] m[i - j]
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < K1; ++j) a += s_m[ax][j] * s_mp[ax][PD + i - j];
            s_m2[ax][PD + i] = a;
        }
        __syncthreads();
        if (i < K3) {                                               // (m * m * m)[i] = sum_j m[j] (m * m)[i - j]
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < K1; ++j) a += s_m[ax][j] * s_m2[ax][PD + i - j];
            s_m3[ax][i] = a;
        }
        __syncthreads();
    }
    PB_PT(25);
    {
        // wave 0 / 1: the kernel along x / y; wave 2 / 3: the composite along x / y
        const int ax = wave & 1;
        int r;
        if (wave < 2) {
            r = tail_radius([&](int i) { return s_m[ax][i]; }, PB_KRAD, KH_HALO_TOL, lane);
        } else if (ps.on) {
            const float c3 = fabsf(ps.a3), c2 = fabsf(ps.a2), c1 = fabsf(ps.a1);
            r = tail_radius([&](int i) {
                    float v = c3 * s_m3[ax][i];
                    if (i >= PB_KRAD && i < PB_KRAD + K2) v += c2 * s_m2[ax][K1 - 1 + i - PB_KRAD];
                    if (i >= 2 * PB_KRAD && i < 2 * PB_KRAD + K1) v += c1 * s_m[ax][i - 2 * PB_KRAD];
                    return v;
                }, 3 * PB_KRAD, KH_HALO_TOL, lane);
        } else {
            r = 3 * PB_KRAD;
        }
        if (lane == 0) s_h[wave] = r;
    }
    __syncthreads();
    PB_PT(26);
    // halos: x in multiples of 4 (windows stay on 16-byte boundaries), y even (conv_wfft.hip stages two window rows at a time)
    const int hxk = max(4, (s_h[0] + 3) & ~3), hyk = max(2, (s_h[1] + 1) & ~1);
    const int hxp = max(4, (s_h[2] + 3) & ~3), hyp = max(2, (s_h[3] + 1) & ~1);
    const int Rh = max(4, (max(s_h[0], s_h[1]) + 3) & ~3);          // the workgroup form's class (same halo on both axes)
    const bool dense = symm && separable == 0;
    // (windows with the 4-sample halo need 8 phases more to beat the stencil body, whose tile is then cheapest: measured)
    const bool use3 = ps.always == 2 ? symm : (dense && nph >= min_phases + (R <= 4 && min_phases > 0 ? 8 : 0));
    // One-pass polynomial.  Mode 1: a kernel within the 4-sample halo class, whatever its phase count and whether rank-1 or
    // not -- the polynomial of a rank-1 kernel is not rank-1, its spectrum is as good as any -- (composite class 12: both
    // bodies).  Mode 2 (wave form only): wherever one pass over windows with the composite's halos costs less than what the
    // image would take otherwise -- three passes over windows with the kernel's halos (window counts compared; a one-pass
    // window needs no x operand, and the pass moves 2 words per sample instead of 8: `gain`), or the stencil bodies
    // (`min_area`: the tile area from which one window pass beats three stencil passes).
    bool poly = false, poly128 = false;
    if (ps.on == 1) poly = symm && Rh <= 4 && max(hxp, hyp) <= 12;
    if (ps.on >= 2 && symm) {
        const int txp = KH_FT_N - 2 * hxp, typ = KH_FT_N - 2 * hyp;
        // (min_area also bounds the job grid: a launch that may carry one-pass images is sized for tiles of that area, and
        // with records built on the device every surplus workgroup is dispatched to find that out)
        const float ak = (float)((KH_FT_N - 2 * hxk) * (KH_FT_N - 2 * hyk));
        // cost of each form in 64 x 64 window pairs per output sample (a stencil evaluation counts like one pass at min_area)
        const float c3 = ps.always == 1 ? INFINITY : (use3 ? 3.f / (ps.gain * ak) : 1.f / (float)ps.min_area);
        float c64 = INFINITY, c128 = INFINITY;
        if (txp >= PB_POLY_MIN_TX && typ >= PB_POLY_MIN_TY && txp * typ >= ps.min_area) c64 = 1.f / (float)(txp * typ);
        const int tx8 = 2 * KH_FT_N - 2 * hxp, ty8 = 2 * KH_FT_N - 2 * hyp;
        if (ps.on == 3 && ps.cost128 > 0.f && tx8 >= PB_POLY128_MIN_T && ty8 >= PB_POLY128_MIN_T) c128 = ps.cost128 / (float)(tx8 * ty8);
        poly128 = c128 < c64 && c128 < c3;
        poly = poly128 || c64 <= c3;
    }
    const bool use = poly || use3;
    if (tid == 0 && slice == 0) {
        sel->use_fft = use ? 1 : 0;
        sel->rf = poly ? (ps.on == 1 ? 12 : (vouched_wrongly ? -1 : 0)) : Rh;
        sel->hx = poly ? hxp : hxk; sel->hy = poly ? hyp : hyk;
        if (!lean) sel->strip = (separable != 0 && R > 8) ? 1 : 0;
        sel->poly = poly128 ? 2 : (poly ? 1 : 0);
        sel->pad_[0] = 0;
        if (!lean) sel->pad_[1] = 0;                 // (lean: the record workgroup's word -- 1 = taps not point-symmetric)
    }
    PB_PT(23);
    if (poly128) { khat128_body<SL>(skp, out, slice, ps, c8, s8); PB_PT(22); return; }
    if (!use) return;
    // the kernel is point-symmetric: rows 12 - u and 12 + u of the first sum are complex conjugates, so only rows 12 .. 24
    // are formed and the second sum is  G[12] + 2 sum_{u > 12} Re(G[u] e^{i phi_u}); this workgroup's x positions only
    const int px0 = slice * PXS;
    if (tid < NR * PXS) {
        const int u = tid / PXS + PB_KRAD, px = px0 + tid % PXS, fx = (px >> 3) + 8 * (px & 7);
        double ar = 0.0, ai = 0.0;
#pragma unroll 5
        for (int v = 0; v < PB_KSIZE; ++v) {
            const int m = (fx * (v - PB_KRAD)) & 63;
            const double k = (double)skp[u * PB_KSIZE + v];
            ar += k * cs[m]; ai += k * sn[m];
        }
        G[tid] = make_double2(ar, ai);
    }
    __syncthreads();
    PB_PT(21);
    for (int idx = tid; idx < PXS * KH_FT_N; idx += KH_THREADS) {          // stored transposed: [x position][y position]
        const int pxl = idx >> 6, py = idx & 63, fy = (py >> 3) + 8 * (py & 7);
        double ar = 0.5 * G[pxl].x;
#pragma unroll 4
        for (int u = 1; u < NR; ++u) {
            const int m = (fy * u) & 63;
            const double2 g = G[u * PXS + pxl];
            ar += g.x * cs[m] - g.y * sn[m];
        }
        double v = 2.0 * ar;                                              // the kernel's transform at (fx, fy): real
        if (poly) v = (((double)ps.a3 * v + (double)ps.a2) * v + (double)ps.a1) * v + (double)ps.b;
        out[(px0 + pxl) * KH_FT_N + py] = (float)(v * (1.0 / 4096.0));
    }
    PB_PT(22);
}

}  // namespace
