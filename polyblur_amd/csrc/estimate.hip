// Blur estimation on the device (reference blur_estimation.py:18-79) and the spectral image
// gradient it is built on (filters.py:159-186).
//
//   gray_minmax_kernel   imgc.mean(dim=1) + amin/amax           blur_estimation.py:36-37,107-108
//   grad_rows_kernel     d/dx by row-wise spectral derivative    filters.py:172-181 (+ normalize :92-109)
//   grad_cols_kernel     d/dy by column-wise spectral derivative filters.py:182-184,
//                        fused with the directional maxima       blur_estimation.py:122-134
//                        and the saturation mask                 blur_estimation.py:83-88,117-118
//   blur_params_kernel   cubic interpolation, argmin, affine model, 25x25 Gaussian
//                                                                blur_estimation.py:138-232
// Everything stays on the stream: no host synchronisation between stages.
#include <algorithm>
#include <cmath>
#include <functional>
#include <cstdlib>

#include "common.h"
// Lab build only (tools/build_variant.sh ltrace "-DPB_EXPERIMENTAL -DPB_LINES_TRACE" estimate.hip): wall-clock stamps (100 MHz)
// of the fused line transform's stages, by thread 0 of workgroups 0, 1/4, 1/2 and the last of the launch -- the first eight
// slots the row kernel's, the next eight the column kernel's (tools/lines_trace.py).
#if defined(PB_EXPERIMENTAL) && defined(PB_LINES_TRACE)
__device__ unsigned long long g_lines_trace[2 * 4 * 8];
__device__ __forceinline__ void pb_lines_stamp(int i) {
    if (threadIdx.x != 0) return;
    const int kernel = blockDim.x >= 512 ? 1 : 0;                  // (4K: 256-thread row workgroups, 1024-thread column workgroups)
    const unsigned g = gridDim.x, b = blockIdx.x;
    const int w = b == 0 ? 0 : (b == g / 4 ? 1 : (b == g / 2 ? 2 : (b == g - 9 ? 3 : -1)));
    if (w >= 0) g_lines_trace[(kernel * 4 + w) * 8 + i] = wall_clock64();
}
#define PB_FT(i) pb_lines_stamp(i)
extern "C" int pb_debug_lines_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lines_trace), sizeof(unsigned long long) * 64);
}
#endif
#include "fft.h"
// Lab build only (tools/build_variant.sh ptrace "-DPB_EXPERIMENTAL -DPB_PARAMS_TRACE" estimate.hip conv_fft.hip): shader-clock
// stamps of the parameter kernel's phases, first workgroup (tools/params_trace.py).  Not in the product build, nor in a plain
// --experimental one.
#if defined(PB_EXPERIMENTAL) && defined(PB_PARAMS_TRACE)
__device__ unsigned long long g_params_trace[32];
#ifdef PB_PT_WANT       // (with it: the kernel body runs twice and the stamps are those of repetition PB_PT_WANT -- 1 = warm instruction cache)
__device__ int g_pt_rep;
#define PB_PT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && g_pt_rep == PB_PT_WANT) g_params_trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PB_PT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_params_trace[i] = __builtin_readcyclecounter(); } while (0)
#endif
extern "C" int pb_debug_params_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_params_trace), sizeof(unsigned long long) * 32);
}
#endif
#include "khat.h"
#include "lines_fixed.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------
// FFT plans (host)
// ------------------------------------------------------------------------------------
static void factorize(int n, std::vector<int> &radix, int &rest, bool ext_ok = false) {
    // greedy: the largest in-register radix that divides what is left (fewer LDS passes and barriers)
    static const int pref[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 7, 3, 2};
    radix.clear();
    const int n0 = n;
    bool found = true;
    while (n > 1 && found) {
        found = false;
        for (int r : pref)
            if (n % r == 0) { radix.push_back(r); n /= r; found = true; break; }
    }
    rest = n;
    // Lines held in LDS: where radices 18 / 20 / 24 save a stage over the greedy plan (4320 = 16 x 15 x 9 x 2 -> 18 x 16 x 15,
    // 7680 = 16 x 16 x 15 x 2 -> 24 x 20 x 16: two LDS round trips and barriers fewer per line), the plan with the fewest
    // stages and, among those, the smallest sum of radices; every other length keeps its greedy plan.
    if (!ext_ok || rest != 1 || (long)n0 * 8 > 160 * 1024 || radix.size() < 3) return;
    static const int ext[] = {24, 20, 18, 16, 15, 12, 10, 9, 8, 6, 5, 4, 7, 3, 2};
    std::vector<int> best, cur;
    int best_sum = 0;
    const size_t limit = radix.size() - 1;                // only plans with fewer stages are of interest
    std::function<void(int, int, int)> dfs = [&](int left, int max_r, int sum) {
        if (left == 1) {
            if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && sum < best_sum)) { best = cur; best_sum = sum; }
            return;
        }
        if (cur.size() >= limit || (!best.empty() && cur.size() >= best.size())) return;
        for (int r : ext) {
            if (r > max_r || left % r) continue;
            cur.push_back(r);
            dfs(left / r, r, sum + r);
            cur.pop_back();
        }
    };
    dfs(n0, 24, 0);
    if (best.empty() || best.size() >= radix.size()) return;
    // order: the first radix is the one of the stages that talk to global memory (and, in the column kernel, carry the
    // epilogue's prefetched operands): the largest one up to 16, as in the greedy plans; the rest ascending, so that the
    // widest butterfly is the innermost stage, which has no twiddles to hold
    std::sort(best.begin(), best.end());
    int first = -1;
    for (int i = (int)best.size() - 1; i >= 0; --i) if (best[i] <= 16) { first = i; break; }
    if (first > 0) std::rotate(best.begin(), best.begin() + first, best.begin() + first + 1);
    radix = best;
}

// frequency index held at position p after the DIF stages
static int digit_reversed_freq(int p, int n, const std::vector<int> &radix) {
    int k = 0, weight = 1, stride = n;
    for (int r : radix) {
        stride /= r;
        const int digit = p / stride;
        p -= digit * stride;
        k += digit * weight;
        weight *= r;
    }
    return k;
}

static double deriv_freq(int k, int n) {
    // filters.py:175-176 in un-shifted order, Nyquist bin dropped (see fft.h)
    if (n % 2 == 0 && k == n / 2) return 0.0;
    return (k <= (n - 1) / 2) ? (double)k / n : (double)(k - n) / n;
}

template <typename T> static T *upload(pb_ctx *ctx, const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    (void)ctx;
    return d;
}

}  // namespace

// The spectral derivative keeps whole lines in LDS (160 KB): lengths up to 20480 when every prime factor is <= 7,
// up to 8192 otherwise (Bluestein with a power-of-two core of at least 2n-1 points) -> 1.  Longer lines, up to 65536
// samples, take the same stages on a line buffer in global memory (grad_*_long_kernel) -> 2.  Beyond that -> 0.
constexpr int kMaxLineLength = 65536;
extern "C" int pb_fft_length_supported(int n) {
    if (n < 2 || n > kMaxLineLength) return 0;
    std::vector<int> radix;
    int rest = 1;
    factorize(n, radix, rest);
    long core = n;
    if (rest != 1) {
        core = 1;
        while (core < 2L * n - 1) core *= 2;
    }
    return core * (long)sizeof(float2) <= 160 * 1024 ? 1 : 2;
}

const FftPlan *pb_get_plan(pb_ctx *ctx, int n, bool ext_radices, int first) {
    // (two plans per length: the kernel variants without the radices above 16 take the greedy one; and one per first radix asked for)
    const int key = (ext_radices ? -1 : 1) * (n + (first << 17));
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) return &it->second;
    FftPlan pl;
    pl.n = n;
    std::vector<int> radix;
    int rest = 1;
    factorize(n, radix, rest, ext_radices && ctx->fft_ext_radix != 0);
    const double two_pi = 6.283185307179586476925286766559;
    int core = n;
    if (rest != 1) {               // Bluestein with a power-of-two core
        core = 1;
        while (core < 2 * n - 1) core *= 2;
        pl.bluestein_m = core;
        factorize(core, radix, rest);
    }
    if (first > 0 && !pl.bluestein_m) {
        auto it = std::find(radix.begin(), radix.end(), first);
        if (it != radix.end()) std::rotate(radix.begin(), it, it + 1);
    }
    if ((int)radix.size() > 24) { pb_fail(ctx, PB_ERR_UNSUPPORTED, "fft length %d: too many stages", n); return nullptr; }
    pl.nstage = (int)radix.size();
    for (int i = 0; i < pl.nstage; ++i) pl.radix[i] = radix[i];
    std::vector<float2> tw(core);
    for (int m = 0; m < core; ++m) {
        const double a = -two_pi * m / core;
        tw[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    pl.tw = upload(ctx, tw);
    bool ok = pl.tw != nullptr;
    if (!pl.bluestein_m) {
        std::vector<float> drev(n);
        for (int p = 0; p < n; ++p) drev[p] = (float)(two_pi * deriv_freq(digit_reversed_freq(p, n, radix), n) / n);
        pl.drev = upload(ctx, drev);
        ok = ok && pl.drev;
    } else {
        const int M = core;
        std::vector<float2> chirp(n);
        std::vector<double> cre(n), cim(n);
        for (int k = 0; k < n; ++k) {
            const long long k2 = ((long long)k * k) % (2LL * n);
            const double a = two_pi * 0.5 * (double)k2 / n;
            cre[k] = std::cos(a); cim[k] = std::sin(a);
            chirp[k] = make_float2((float)cre[k], (float)cim[k]);
        }
        // filter b_M[m] = w[|m|] wrapped; its DFT in double (O(M log M) not needed: direct O(M*N) is fine
        // for a one-off plan, but keep it cheap with a simple radix-2 recursion-free FFT)
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        for (int m = 0; m < n; ++m) {
            br[m] = cre[m]; bi[m] = cim[m];
            if (m) { br[M - m] = cre[m]; bi[M - m] = cim[m]; }
        }
        // iterative radix-2 FFT (M is a power of two)
        for (int i = 1, j = 0; i < M; ++i) {
            int bit = M >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
        }
        for (int len = 2; len <= M; len <<= 1) {
            const double ang = -two_pi / len;
            for (int i = 0; i < M; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const double wr = std::cos(ang * k), wi = std::sin(ang * k);
                    const int a = i + k, b = i + k + len / 2;
                    const double xr = br[b] * wr - bi[b] * wi, xi = br[b] * wi + bi[b] * wr;
                    br[b] = br[a] - xr; bi[b] = bi[a] - xi;
                    br[a] += xr; bi[a] += xi;
                }
        }
        std::vector<float2> brev(M);
        for (int p = 0; p < M; ++p) {
            const int k = digit_reversed_freq(p, M, radix);
            brev[p] = make_float2((float)(br[k] / M), (float)(bi[k] / M));
        }
        std::vector<float> dnat(n);
        for (int k = 0; k < n; ++k) dnat[k] = (float)(two_pi * deriv_freq(k, n) / n);
        pl.chirp = upload(ctx, chirp);
        pl.bfilt_rev = upload(ctx, brev);
        pl.dnat = upload(ctx, dnat);
        ok = ok && pl.chirp && pl.bfilt_rev && pl.dnat;
    }
    if (!ok) { pb_fail(ctx, PB_ERR_NOMEM, "fft plan %d: device allocation failed", n); return nullptr; }
    auto res = ctx->plans.emplace(key, pl);
    return &res.first->second;
}

const float *pb_get_interp_weights(pb_ctx *ctx, int n_angles, int n_interp) {
    if (ctx->interp_w && ctx->interp_na == n_angles && ctx->interp_ni == n_interp) return ctx->interp_w;
    if (ctx->interp_w) { (void)hipFree(ctx->interp_w); ctx->interp_w = nullptr; }
    // Keys cubic weights exactly as blur_estimation.py:138-147 evaluates them in fp32:
    // both angle grids are truncated to integers (deblurring.py:62-63) and divided by N.
    const int na = n_angles + 1;
    std::vector<float> w((size_t)n_interp * na);
    for (int i = 0; i < n_interp; ++i) {
        const float xn = (float)(long)((double)i * (180.0 / n_interp)) / (float)n_interp;
        float sum = 0.f;
        for (int k = 0; k < na; ++k) {
            const float xo = (float)(long)(180.0 * k / n_angles) / (float)n_interp;
            const float d = std::fabs(xn - xo);
            float v = 0.f;
            if (d < 1.f) v = (1.5f * d - 2.5f) * d * d + 1.f;
            else if (d < 2.f) v = ((-0.5f * d + 2.5f) * d - 4.f) * d + 2.f;
            w[(size_t)i * na + k] = v;
            sum += v;
        }
        for (int k = 0; k < na; ++k) w[(size_t)i * na + k] /= (sum + 1e-5f);
    }
    ctx->interp_w = upload(ctx, w);
    if (!ctx->interp_w) { pb_fail(ctx, PB_ERR_NOMEM, "interp weights: allocation failed"); return nullptr; }
    ctx->interp_na = n_angles;
    ctx->interp_ni = n_interp;
    return ctx->interp_w;
}

namespace {

pbfft::DevPlan dev_plan(const FftPlan *pl) {
    pbfft::DevPlan d;
    d.line_n = pl->n;
    d.n = pl->bluestein_m ? pl->bluestein_m : pl->n;
    d.nstage = pl->nstage;
    for (int i = 0; i < 24; ++i) d.radix[i] = i < pl->nstage ? pl->radix[i] : 1;
    d.tw = pl->tw;
    d.drev = pl->drev;
    d.chirp = pl->chirp;
    d.bfilt_rev = pl->bfilt_rev;
    d.dnat = pl->dnat;
    return d;
}

// ------------------------------------------------------------------------------------
// gray + min/max
// ------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float4 ld4e(const T *p);
template <> __device__ __forceinline__ float4 ld4e<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4e<__half>(const __half *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&u.x)), b = __half22float2(*reinterpret_cast<const __half2 *>(&u.y));
    return make_float4(a.x, a.y, b.x, b.y);
}
template <> __device__ __forceinline__ float4 ld4e<unsigned char>(const unsigned char *p) {
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return make_float4(pb_from_ubyte(u & 255u), pb_from_ubyte((u >> 8) & 255u), pb_from_ubyte((u >> 16) & 255u), pb_from_ubyte(u >> 24));
}

// VEC: HW % 4 == 0, so every plane starts 16-byte aligned and the image is walked in float4 units.
// CC: compile-time channel count (1 or 3; 0 = run-time C) so that all channel loads of a sample
// group are independent and in flight together.
template <typename T, bool VEC, int CC>
__global__ __launch_bounds__(NT) void gray_minmax_kernel(const T *__restrict__ in, float *__restrict__ gray,
                                                         float2 *__restrict__ part, int Crt, long HW, int blocks_per_image) {
    const int C = CC ? CC : Crt;
    const int b = blockIdx.x / blocks_per_image;
    const int blk = blockIdx.x - b * blocks_per_image;
    const T *src = in + (long)b * C * HW;
    float *dst = gray + (long)b * HW;
    float lo = INFINITY, hi = -INFINITY;
    const float invc = 1.f / (float)C;
    if (VEC) {
        const long n4 = HW >> 2;
        for (long i = (long)blk * NT + threadIdx.x; i < n4; i += (long)blocks_per_image * NT) {
            float4 s4 = ld4e<T>(src + 4 * i);
            if (CC == 3) {
                const float4 t1 = ld4e<T>(src + HW + 4 * i), t2 = ld4e<T>(src + 2 * HW + 4 * i);
                s4.x = s4.x + t1.x + t2.x; s4.y = s4.y + t1.y + t2.y; s4.z = s4.z + t1.z + t2.z; s4.w = s4.w + t1.w + t2.w;
            } else {
                for (int c = 1; c < C; ++c) {
                    const float4 t = ld4e<T>(src + c * HW + 4 * i);
                    s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
                }
            }
            float4 g;
            if (C == 1) g = s4;
            else if (C == 3) g = make_float4(s4.x / 3.0f, s4.y / 3.0f, s4.z / 3.0f, s4.w / 3.0f);
            else g = make_float4(s4.x * invc, s4.y * invc, s4.z * invc, s4.w * invc);
            *reinterpret_cast<float4 *>(dst + 4 * i) = g;
            lo = fminf(fminf(lo, fminf(g.x, g.y)), fminf(g.z, g.w));
            hi = fmaxf(fmaxf(hi, fmaxf(g.x, g.y)), fmaxf(g.z, g.w));
        }
    } else {
        for (long i = (long)blk * NT + threadIdx.x; i < HW; i += (long)blocks_per_image * NT) {
            float s = pb_ld(src + i);
            for (int c = 1; c < C; ++c) s += pb_ld(src + c * HW + i);
            const float g = (C == 1) ? s : ((C == 3) ? s / 3.0f : s * invc);
            dst[i] = g;
            lo = fminf(lo, g);
            hi = fmaxf(hi, g);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[NT / 64], shi[NT / 64];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        // one partial per workgroup, folded by minmax_reduce_kernel: atomics on one address serialise at
        // ~20 ns each (measured, tools/ubench3.hip) and would cost 3x the whole image read
        part[(long)b * blocks_per_image + blk] = make_float2(lo, hi);
    }
}

__global__ __launch_bounds__(NT) void minmax_reduce_kernel(const float2 *__restrict__ part, unsigned *__restrict__ mm,
                                                           int blocks_per_image) {
    const int b = blockIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < blocks_per_image; i += NT) {
        const float2 p = part[(long)b * blocks_per_image + i];
        lo = fminf(lo, p.x);
        hi = fmaxf(hi, p.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[NT / 64], shi[NT / 64];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        mm[2 * b] = pb_f2ord(lo);
        mm[2 * b + 1] = pb_f2ord(hi);
    }
}

// ------------------------------------------------------------------------------------
// quantile normalisation (q > 0): exact order statistics by 3-level radix select
// ------------------------------------------------------------------------------------
// normalize() clips the image to its q / 1-q quantiles (blur_estimation.py:102-105, torch.quantile with
// linear interpolation).  Each quantile needs two adjacent order statistics; the four targets per image
// are found exactly on the order-preserving 32-bit keys in three passes over the gray image
// (12 + 12 + 8 bits), each pass a histogram restricted to the prefixes chosen so far.
struct QuantSel {
    unsigned rank[4];      // residual rank of each target inside its current prefix
    unsigned prefix[4];    // key bits fixed so far (left-aligned)
    float weight[2];       // interpolation weights of the low / high quantile
};

template <int LEVEL>   // 0: bits 31..20, 1: bits 19..8, 2: bits 7..0
__global__ __launch_bounds__(NT) void quant_hist_kernel(const float *__restrict__ gray, const QuantSel *__restrict__ sel,
                                                        unsigned *__restrict__ hist, long HW, int blocks_per_image) {
    constexpr int NB = LEVEL == 2 ? 256 : 4096;
    constexpr int NTGT = LEVEL == 0 ? 1 : 4;
    extern __shared__ unsigned sh[];                       // NTGT x NB counters
    const int b = blockIdx.x / blocks_per_image, blk = blockIdx.x - b * blocks_per_image;
    for (int i = threadIdx.x; i < NTGT * NB; i += NT) sh[i] = 0u;
    unsigned pre[4] = {0u, 0u, 0u, 0u};
    if (LEVEL > 0)
        for (int t = 0; t < 4; ++t) pre[t] = sel[b].prefix[t];
    __syncthreads();
    const float *src = gray + (long)b * HW;
    for (long i = (long)blk * NT + threadIdx.x; i < HW; i += (long)blocks_per_image * NT) {
        const unsigned key = pb_f2ord(src[i]);
        if (LEVEL == 0) atomicAdd(&sh[key >> 20], 1u);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (LEVEL == 1 && (key >> 20) == (pre[t] >> 20)) atomicAdd(&sh[t * NB + ((key >> 8) & 0xfffu)], 1u);
                if (LEVEL == 2 && (key >> 8) == (pre[t] >> 8)) atomicAdd(&sh[t * NB + (key & 0xffu)], 1u);
            }
        }
    }
    __syncthreads();
    unsigned *dst = hist + (long)b * NTGT * NB;
    for (int i = threadIdx.x; i < NTGT * NB; i += NT)
        if (sh[i]) atomicAdd(dst + i, sh[i]);
}

// one workgroup per image: locate, for every target, the bin that holds its rank
template <int LEVEL>
__global__ __launch_bounds__(NT) void quant_scan_kernel(const unsigned *__restrict__ hist, QuantSel *__restrict__ sel,
                                                        unsigned *__restrict__ mm, long HW, float q_lo, float q_hi) {
    constexpr int NB = LEVEL == 2 ? 256 : 4096;
    constexpr int NTGT = LEVEL == 0 ? 1 : 4;
    constexpr int PER = NB / NT;
    __shared__ unsigned part[NT];
    __shared__ QuantSel s;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        if (LEVEL == 0) {
            // torch.quantile: rank = q * (n - 1) in the input dtype, lerp between floor and floor + 1
            const float n1 = (float)(HW - 1);
            // the product must be rounded to fp32 BEFORE the floor is subtracted, like torch does;
            // the empty asm keeps the compiler from contracting (q * n1 - floor) into one FMA
            float r_lo = q_lo * n1, r_hi = q_hi * n1;
            asm volatile("" : "+v"(r_lo), "+v"(r_hi));
            const float f_lo = floorf(r_lo), f_hi = floorf(r_hi);
            s.rank[0] = (unsigned)f_lo; s.rank[1] = (unsigned)fminf(f_lo + 1.f, n1);
            s.rank[2] = (unsigned)f_hi; s.rank[3] = (unsigned)fminf(f_hi + 1.f, n1);
            s.weight[0] = r_lo - f_lo; s.weight[1] = r_hi - f_hi;
            for (int t = 0; t < 4; ++t) s.prefix[t] = 0u;
        } else s = sel[b];
    }
    __syncthreads();
    for (int t = 0; t < 4; ++t) {
        const unsigned *h = hist + ((long)b * NTGT + (NTGT == 1 ? 0 : t)) * NB;
        unsigned sum = 0;
        for (int i = 0; i < PER; ++i) sum += h[threadIdx.x * PER + i];
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned rank = s.rank[t], acc = 0;
            int seg = 0;
            for (; seg < NT - 1; ++seg) {                      // find the 1/NT segment, then the bin
                if (acc + part[seg] > rank) break;
                acc += part[seg];
            }
            int bin = seg * PER;
            for (; bin < seg * PER + PER - 1; ++bin) {
                if (acc + h[bin] > rank) break;
                acc += h[bin];
            }
            s.rank[t] = rank - acc;
            s.prefix[t] |= LEVEL == 0 ? ((unsigned)bin << 20) : (LEVEL == 1 ? ((unsigned)bin << 8) : (unsigned)bin);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (LEVEL < 2) sel[b] = s;
        else {
            const float a0 = pb_ord2f(s.prefix[0]), a1 = pb_ord2f(s.prefix[1]);
            const float b0 = pb_ord2f(s.prefix[2]), b1 = pb_ord2f(s.prefix[3]);
            // torch.lerp
            const float wl = s.weight[0], wh = s.weight[1];
            const float lo = wl < 0.5f ? a0 + wl * (a1 - a0) : a1 - (a1 - a0) * (1.f - wl);
            const float hi = wh < 0.5f ? b0 + wh * (b1 - b0) : b1 - (b1 - b0) * (1.f - wh);
            mm[2 * b] = pb_f2ord(lo);
            mm[2 * b + 1] = pb_f2ord(hi);
        }
    }
}

// ------------------------------------------------------------------------------------
// spectral derivative along rows: one workgroup = two rows packed as one complex line
// ------------------------------------------------------------------------------------
// normalize: lines are range-normalised on load with the per-image (lo, hi) in mm
// (blur_estimation.py:92-93); planes_per_image maps a plane to its image's min/max.
// (x - lo) / scale clipped to [0, 1] (blur_estimation.py:92-93) with the quotient in three instructions instead of
// the dozen of an IEEE division: q = a * r, then one correction step with the exact residual, r = RN(1 / scale)
// computed once per workgroup (Markstein: the corrected quotient is the correctly rounded one, but for a scale whose
// significand is all ones).  A constant image (scale == 0) gives NaN -> 0 either way.
__device__ __forceinline__ float norm01(float x, float lo, float scale, float inv) {
    const float a = x - lo;
    float q = a * inv;
    q = fmaf(fmaf(-q, scale, a), inv, q);
    return fminf(fmaxf(q, 0.f), 1.f);
}

// how the fused transform (fft.h, spectral_derivative_fused) reaches the two rows of a workgroup
struct RowsIO {
    struct Pre {};
    const float *row0, *row1;
    float *o0;
    int W;
    bool has1, normalize;
    float lo, scale, inv;
    __device__ __forceinline__ float2 load(int p, int) const {
        float a = row0[p], b = has1 ? row1[p] : 0.f;
        if (normalize) {
            a = norm01(a, lo, scale, inv);
            b = has1 ? norm01(b, lo, scale, inv) : 0.f;
        }
        return make_float2(a, b);
    }
    __device__ __forceinline__ Pre prefetch(int, int) const { return Pre(); }
    __device__ __forceinline__ void store(int p, int, float2 v, Pre) const {
        o0[p] = v.x;
        if (has1) o0[W + p] = -v.y;
    }
};

// FUSED: the fused transform (direct plans of two or more stages), with 128 threads for lines of up to 4096 samples;
// otherwise the general one (Bluestein, one stage)
template <int NTH, bool FUSED>
__global__ __launch_bounds__(NTH, (NTH == 256 && FUSED) ? 5 : 1) void grad_rows_kernel(const float *__restrict__ planes, float *__restrict__ gx,
                                                       int H, int W, int normalize, const unsigned *__restrict__ mm,
                                                       int planes_per_image, pbfft::DevPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    const int pairs = (H + 1) / 2;
    const int plane = blockIdx.x / pairs;
    const int r0 = 2 * (blockIdx.x - plane * pairs);
    const bool has1 = r0 + 1 < H;
    const float *row0 = planes + ((long)plane * H + r0) * W;
    const float *row1 = row0 + W;
    float lo = 0.f, scale = 1.f;
    if (normalize) {
        const int img = plane / planes_per_image;
        lo = pb_ord2f(mm[2 * img]);
        scale = pb_ord2f(mm[2 * img + 1]) - lo;
    }
    const float inv = 1.f / scale;
    if constexpr (FUSED) {
        RowsIO io{row0, row1, gx + ((long)plane * H + r0) * W, W, has1, normalize != 0, lo, scale, inv};
        pbfft::spectral_derivative_fused<(NTH == 512 ? 24 : 16)>(sfft, plan, 0, io);    // (512 threads: the variant of the long lines, radices up to 24)
        return;
    }
    const bool vec = (W & 3) == 0;                 // rows are 16-byte aligned: whole-row float4 traffic
    if (vec) {
        const int W4 = W >> 2;
        const float4 *r0 = reinterpret_cast<const float4 *>(row0), *r1 = reinterpret_cast<const float4 *>(row1);
        for (int base = 0; base < W4; base += 4 * NTH) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                      // all loads of the batch in flight together
                const int i = base + u * NTH + threadIdx.x;
                a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                b[u] = a[u];
                if (i < W4) { a[u] = r0[i]; if (has1) b[u] = r1[i]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * NTH + threadIdx.x;
                if (i < W4) {
                    float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (normalize) {
                            av[e] = norm01(av[e], lo, scale, inv);
                            bv[e] = has1 ? norm01(bv[e], lo, scale, inv) : 0.f;
                        }
                        sfft[4 * i + e] = make_float2(av[e], bv[e]);
                    }
                }
            }
        }
    } else {
        for (int n = threadIdx.x; n < W; n += NTH) {
            float a = row0[n], b = has1 ? row1[n] : 0.f;
            if (normalize) {
                a = norm01(a, lo, scale, inv);
                b = has1 ? norm01(b, lo, scale, inv) : 0.f;
            }
            sfft[n] = make_float2(a, b);
        }
    }
    __syncthreads();
    pbfft::spectral_derivative(sfft, plan, 0);
    float *o0 = gx + ((long)plane * H + r0) * W;
    if (vec) {
        const int W4 = W >> 2;
        for (int i = threadIdx.x; i < W4; i += NTH) {
            const float2 v0 = sfft[4 * i], v1 = sfft[4 * i + 1], v2 = sfft[4 * i + 2], v3 = sfft[4 * i + 3];
            reinterpret_cast<float4 *>(o0)[i] = make_float4(v0.x, v1.x, v2.x, v3.x);
            if (has1) reinterpret_cast<float4 *>(o0 + W)[i] = make_float4(-v0.y, -v1.y, -v2.y, -v3.y);
        }
    } else {
        for (int n = threadIdx.x; n < W; n += NTH) {
            const float2 v = sfft[n];
            o0[n] = v.x;
            if (has1) o0[W + n] = -v.y;
        }
    }
}

// ------------------------------------------------------------------------------------
// gray + its range + the row transform in ONE pass (q == 0, lines that take the fused transform): the first stage's loads
// read the image's channels, form the gray samples exactly as gray_minmax_kernel does (blur_estimation.py:36), store them
// for the column pass and fold them into the workgroup's (min, max) partial -- the gray plane is written once and read
// once instead of written once and read twice, and one launch of a whole-image pass goes away.
// ------------------------------------------------------------------------------------
template <typename T, int CC> struct GrayRowsIO {
    struct Pre {};
    const T *row0;             // channel 0, first row of the pair
    long cstride;              // samples between channels
    float *g0, *o0;            // gray and gx, first row of the pair
    int W, C;
    bool has1;
    float invc, lo, hi;
    __device__ __forceinline__ float2 load(int p, int) {
        const int nc = CC ? CC : C;
        float a = pb_ld(row0 + p), b = has1 ? pb_ld(row0 + W + p) : 0.f;
        if (CC == 3) {
            const float a1 = pb_ld(row0 + cstride + p), a2 = pb_ld(row0 + 2 * cstride + p);
            const float b1 = has1 ? pb_ld(row0 + cstride + W + p) : 0.f, b2 = has1 ? pb_ld(row0 + 2 * cstride + W + p) : 0.f;
            a = a + a1 + a2; b = b + b1 + b2;
        } else {
            for (int c = 1; c < nc; ++c) {
                a += pb_ld(row0 + c * cstride + p);
                if (has1) b += pb_ld(row0 + c * cstride + W + p);
            }
        }
        if (nc == 3) { a = a / 3.0f; b = b / 3.0f; }
        else if (nc != 1) { a *= invc; b *= invc; }
        g0[p] = a;
        lo = fminf(lo, a); hi = fmaxf(hi, a);
        if (has1) { g0[W + p] = b; lo = fminf(lo, b); hi = fmaxf(hi, b); }
        return make_float2(a, b);
    }
    __device__ __forceinline__ Pre prefetch(int, int) const { return Pre(); }
    __device__ __forceinline__ void store(int p, int, float2 v, Pre) const {
        o0[p] = v.x;
        if (has1) o0[W + p] = -v.y;
    }
};

template <typename T, int CC, int NTH>
__global__ __launch_bounds__(NTH, NTH == 256 ? 5 : 1) void gray_rows_kernel(const T *__restrict__ in, float *__restrict__ gray, float *__restrict__ gx,
                                                                            float2 *__restrict__ part, int C, int H, int W, pbfft::DevPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    const int pairs = (H + 1) / 2;
    const int b = blockIdx.x / pairs, pr = blockIdx.x - b * pairs;
    const int r0 = 2 * pr;
    const long HW = (long)H * W;
    GrayRowsIO<T, CC> io;
    io.row0 = in + (long)b * C * HW + (long)r0 * W;
    io.cstride = HW;
    io.g0 = gray + (long)b * HW + (long)r0 * W;
    io.o0 = gx + (long)b * HW + (long)r0 * W;
    io.W = W; io.C = C; io.has1 = r0 + 1 < H; io.invc = 1.f / (float)C;
    io.lo = INFINITY; io.hi = -INFINITY;
    pbfft::spectral_derivative_fused<(NTH == 512 ? 24 : 16)>(sfft, plan, 0, io);
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

// ------------------------------------------------------------------------------------
// spectral derivative along columns: one workgroup = 2*NB adjacent columns (NB complex lines)
// MODE 0: write gy.  MODE 1: fuse the directional maxima (needs gx of the same plane).
// ------------------------------------------------------------------------------------
// workgroup maximum of every direction -> one partial per column tile, stored direction-major (row k of an image holds
// the tiles' maxima of direction k, so that blur_params_kernel folds a row with contiguous 16-byte loads; no contended
// atomics).  red: NTH/64 * PB_MAX_ANGLES floats of LDS nobody else is using; mags_tile = &row 0 [this tile].
template <int NTH>
__device__ __forceinline__ void reduce_maxima(const float (&best)[PB_MAX_ANGLES], float *red, unsigned *__restrict__ mags_tile,
                                              int tiles_pad, int n_angles) {
#pragma unroll
    for (int k = 0; k < PB_MAX_ANGLES; ++k) {
        float m = best[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * PB_MAX_ANGLES + k] = m;
    }
    __syncthreads();
    if ((int)threadIdx.x <= n_angles) {
        float m = red[threadIdx.x];
        for (int w = 1; w < NTH / 64; ++w) m = fmaxf(m, red[w * PB_MAX_ANGLES + threadIdx.x]);
        mags_tile[(long)threadIdx.x * tiles_pad] = __float_as_uint(m);   // m >= 0
    }
}

// cos / sin of the n_angles + 1 directions k pi / n_angles (blur_estimation.py:129-131), evaluated once on the host
struct AngleTable {
    float cs[PB_MAX_ANGLES], sn[PB_MAX_ANGLES];
};

// how the fused transform reaches the 2*nb columns of a workgroup.  MODE 0 writes gy; MODE 1 folds every output
// sample, with the gx of the same position, into the directional maxima (NA > 0: n_angles + 1 at compile time).
template <int MODE, int NA> struct ColsIO {
    struct Pre { float2 dx, g; };
    const float *src, *gxp;
    float *dst;
    int W, c0, na;
    bool vec2, discard_sat, normalize;
    float lo, scale, inv, sat_threshold;
    AngleTable ang;
    float best[PB_MAX_ANGLES];
    __device__ __forceinline__ float2 load(int p, int j) const {
        const int c = c0 + 2 * j;
        float2 v = make_float2(0.f, 0.f);
        if (vec2) v = *reinterpret_cast<const float2 *>(src + (long)p * W + c);
        else {
            if (c < W) v.x = src[(long)p * W + c];
            if (c + 1 < W) v.y = src[(long)p * W + c + 1];
        }
        if (normalize) {
            v.x = norm01(v.x, lo, scale, inv);
            v.y = norm01(v.y, lo, scale, inv);
        }
        return v;
    }
    __device__ __forceinline__ Pre prefetch(int p, int j) const {
        Pre r;
        r.dx = make_float2(0.f, 0.f);
        r.g = make_float2(0.f, 0.f);
        if (MODE == 1) {
            const int c = c0 + 2 * j;
            const long idx = (long)p * W + c;
            if (vec2) {
                r.dx = *reinterpret_cast<const float2 *>(gxp + idx);
                if (discard_sat) r.g = *reinterpret_cast<const float2 *>(src + idx);
            } else {
                if (c < W) { r.dx.x = gxp[idx]; if (discard_sat) r.g.x = src[idx]; }
                if (c + 1 < W) { r.dx.y = gxp[idx + 1]; if (discard_sat) r.g.y = src[idx + 1]; }
            }
        }
        return r;
    }
    __device__ __forceinline__ void fold(float dx, float dy) {
        // m_k = max |cos(t_k) gx - sin(t_k) gy|  (blur_estimation.py:129-133)
#pragma unroll
        for (int k = 0; k < (NA ? NA : PB_MAX_ANGLES); ++k)
            if (NA || k < na) best[k] = fmaxf(best[k], pbfft::dir_abs(ang.cs[k], ang.sn[k], dx, dy));
    }
    __device__ __forceinline__ void store(int p, int j, float2 v, const Pre &pre) {
        const int c = c0 + 2 * j;
        if (MODE == 0) {
            if (c < W) dst[(long)p * W + c] = v.x;
            if (c + 1 < W) dst[(long)p * W + c + 1] = -v.y;
        } else {
            // gradients are zeroed under the saturation mask (blur_estimation.py:117-118): they cannot raise a maximum
            if (c < W && !(discard_sat && pre.g.x > sat_threshold)) fold(pre.dx.x, v.x);
            if (c + 1 < W && !(discard_sat && pre.g.y > sat_threshold)) fold(pre.dx.y, -v.y);
        }
    }
};

template <int MODE, int NA, int NTH, int MAXR = 16>
__global__ __launch_bounds__(NTH) void grad_cols_kernel(const float *__restrict__ planes, const float *__restrict__ gx,
                                                       float *__restrict__ gy, int H, int W, int lognb, int normalize,
                                                       const unsigned *__restrict__ mm, int planes_per_image,
                                                       unsigned *__restrict__ mags, int n_angles, int discard_sat,
                                                       float sat_threshold, int total_tiles, pbfft::DevPlan plan,
                                                       AngleTable ang) {
    extern __shared__ __attribute__((aligned(16))) float2 sfft[];
    const int nb = 1 << lognb;
    const int tc = 2 * nb;
    const int tiles = (W + tc - 1) / tc;
    // adjacent column tiles share 128-byte lines: give each XCD (= blockIdx % 8, speed only) a contiguous
    // run of tiles so that those lines are fetched into one L2 once (grid padded to a multiple of 8)
    const int chunk = gridDim.x >> 3;
    const int tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int plane = tile_id / tiles;
    const int c0 = (tile_id - plane * tiles) * tc;
    const int tiles_pad = (tiles + 3) & ~3;
    unsigned *mags_tile = mags + (long)plane * PB_MAX_ANGLES * tiles_pad + (tile_id - plane * tiles);
    const float *src = planes + (long)plane * H * W;
    float lo = 0.f, scale = 1.f;
    if (normalize) {
        const int img = plane / planes_per_image;
        lo = pb_ord2f(mm[2 * img]);
        scale = pb_ord2f(mm[2 * img + 1]) - lo;
    }
    const float inv = 1.f / scale;
    // element e = p*nb + j  <->  row p, columns c0+2j, c0+2j+1
    const bool vec2 = (W & 1) == 0 && c0 + tc <= W;          // every pair is an aligned, in-range float2
    if (pbfft::fused_plan(plan)) {
        ColsIO<MODE, NA> io;
        io.src = src;
        io.gxp = MODE == 1 ? gx + (long)plane * H * W : nullptr;
        io.dst = MODE == 0 ? gy + (long)plane * H * W : nullptr;
        io.W = W; io.c0 = c0; io.na = n_angles + 1; io.vec2 = vec2; io.discard_sat = discard_sat != 0; io.normalize = normalize != 0;
        io.lo = lo; io.scale = scale; io.inv = inv; io.sat_threshold = sat_threshold; io.ang = ang;
#pragma unroll
        for (int k = 0; k < PB_MAX_ANGLES; ++k) io.best[k] = 0.f;
        pbfft::spectral_derivative_fused<MAXR>(sfft, plan, lognb, io);
        if (MODE == 1) {
            __syncthreads();
            reduce_maxima<NTH>(io.best, reinterpret_cast<float *>(sfft), mags_tile, tiles_pad, n_angles);
        }
        return;
    }
    for (int base = 0; base < (H << lognb); base += 8 * NTH) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                          // 8 independent loads in flight per thread
            const int e = base + u * NTH + threadIdx.x;
            v[u] = make_float2(0.f, 0.f);
            if (e < (H << lognb)) {
                const int p = e >> lognb, j = e & (nb - 1);
                const int c = c0 + 2 * j;
                if (vec2) v[u] = *reinterpret_cast<const float2 *>(src + (long)p * W + c);
                else {
                    if (c < W) v[u].x = src[(long)p * W + c];
                    if (c + 1 < W) v[u].y = src[(long)p * W + c + 1];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * NTH + threadIdx.x;
            if (e < (H << lognb)) {
                float a = v[u].x, b = v[u].y;
                if (normalize) {
                    a = norm01(a, lo, scale, inv);
                    b = norm01(b, lo, scale, inv);
                }
                sfft[e] = make_float2(a, b);
            }
        }
    }
    __syncthreads();
    pbfft::spectral_derivative(sfft, plan, lognb);
    if (MODE == 0) {
        float *dst = gy + (long)plane * H * W;
        for (int e = threadIdx.x; e < (H << lognb); e += NTH) {
            const int p = e >> lognb, j = e & (nb - 1);
            const int c = c0 + 2 * j;
            const float2 v = sfft[e];
            if (c < W) dst[(long)p * W + c] = v.x;
            if (c + 1 < W) dst[(long)p * W + c + 1] = -v.y;
        }
    } else {
        // m_k = max |cos(t_k) gx - sin(t_k) gy|, t_k = k pi / n_angles  (blur_estimation.py:129-133)
        float best[PB_MAX_ANGLES];
#pragma unroll
        for (int k = 0; k < PB_MAX_ANGLES; ++k) best[k] = 0.f;
        const float *gxp = gx + (long)plane * H * W;
        for (int base = 0; base < (H << lognb); base += 8 * NTH) {
            float2 dxv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + threadIdx.x;
                dxv[u] = make_float2(0.f, 0.f);
                gv[u] = make_float2(0.f, 0.f);
                if (e < (H << lognb)) {
                    const int p = e >> lognb, j = e & (nb - 1);
                    const long idx = (long)p * W + c0 + 2 * j;
                    if (vec2) {
                        dxv[u] = *reinterpret_cast<const float2 *>(gxp + idx);
                        if (discard_sat) gv[u] = *reinterpret_cast<const float2 *>(src + idx);
                    } else {
                        if (c0 + 2 * j < W) { dxv[u].x = gxp[idx]; if (discard_sat) gv[u].x = src[idx]; }
                        if (c0 + 2 * j + 1 < W) { dxv[u].y = gxp[idx + 1]; if (discard_sat) gv[u].y = src[idx + 1]; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * NTH + threadIdx.x;
                if (e >= (H << lognb)) continue;
                const int j = e & (nb - 1);
                const int c = c0 + 2 * j;
                const float2 vv = sfft[e];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (c + h >= W) continue;
                    if (discard_sat && (h ? gv[u].y : gv[u].x) > sat_threshold) continue;   // gradients zeroed under the mask
                    const float dx = h ? dxv[u].y : dxv[u].x;
                    const float dy = h ? -vv.y : vv.x;
#pragma unroll
                    for (int k = 0; k < PB_MAX_ANGLES; ++k)
                        if (k <= n_angles) best[k] = fmaxf(best[k], pbfft::dir_abs(ang.cs[k], ang.sn[k], dx, dy));
                }
            }
        }
        __syncthreads();
        reduce_maxima<NTH>(best, reinterpret_cast<float *>(sfft), mags_tile, tiles_pad, n_angles);
    }
}

// ------------------------------------------------------------------------------------
// Lines that do not fit LDS (sides above 20480, or above 8192 with a prime factor > 7; the reference's torch.fft takes
// any length, filters.py:172-184): the same transform, stage by stage, on a line buffer in GLOBAL memory.  A workgroup
// owns one slot of the context's scratch (plan.n complex values per line: a few hundred KB, which stays in its XCD's L2
// between the stages), walks over its share of the lines, and a stage's barrier orders its global accesses like its LDS
// ones (all waves of a workgroup share the CU's L1, which is write-through).  Ten or so L2 round trips per line instead
// of LDS ones: a fallback, several times slower per sample than the in-LDS kernels, for sizes that used to raise.
// ------------------------------------------------------------------------------------
constexpr int LONG_NT = 1024;

__global__ __launch_bounds__(LONG_NT) void grad_rows_long_kernel(const float *__restrict__ planes, float *__restrict__ gx, int H, int W,
                                                                int normalize, const unsigned *__restrict__ mm, int planes_per_image,
                                                                pbfft::DevPlan plan, float2 *scratch, long items) {
    float2 *s = scratch + (size_t)blockIdx.x * (size_t)plan.n;
    const int pairs = (H + 1) / 2;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int plane = (int)(item / pairs);
        const int r0 = 2 * (int)(item - (long)plane * pairs);
        const bool has1 = r0 + 1 < H;
        const float *row0 = planes + ((long)plane * H + r0) * W;
        const float *row1 = row0 + W;
        float lo = 0.f, scale = 1.f;
        if (normalize) {
            const int img = plane / planes_per_image;
            lo = pb_ord2f(mm[2 * img]);
            scale = pb_ord2f(mm[2 * img + 1]) - lo;
        }
        const float inv = 1.f / scale;
        for (int n = threadIdx.x; n < W; n += LONG_NT) {
            float a = row0[n], b = has1 ? row1[n] : 0.f;
            if (normalize) {
                a = norm01(a, lo, scale, inv);
                b = has1 ? norm01(b, lo, scale, inv) : 0.f;
            }
            s[n] = make_float2(a, b);
        }
        __syncthreads();
        pbfft::spectral_derivative(s, plan, 0);
        float *o0 = gx + ((long)plane * H + r0) * W;
        for (int n = threadIdx.x; n < W; n += LONG_NT) {
            const float2 v = s[n];
            o0[n] = v.x;
            if (has1) o0[W + n] = -v.y;
        }
        __syncthreads();                                       // the slot is free for the next pair of rows
    }
}

// MODE as grad_cols_kernel: 0 writes gy, 1 folds (gx, gy) into the directional maxima of the tile
template <int MODE>
__global__ __launch_bounds__(LONG_NT) void grad_cols_long_kernel(const float *__restrict__ planes, const float *__restrict__ gx,
                                                                float *__restrict__ gy, int H, int W, int lognb, int normalize,
                                                                const unsigned *__restrict__ mm, int planes_per_image,
                                                                unsigned *__restrict__ mags, int n_angles, int discard_sat,
                                                                float sat_threshold, int total_tiles, pbfft::DevPlan plan,
                                                                AngleTable ang, float2 *scratch) {
    __shared__ float red[LONG_NT / 64 * PB_MAX_ANGLES];
    float2 *s = scratch + (size_t)blockIdx.x * ((size_t)plan.n << lognb);
    const int nb = 1 << lognb;
    const int tc = 2 * nb;
    const int tiles = (W + tc - 1) / tc;
    const int tiles_pad = (tiles + 3) & ~3;
    const int work = H << lognb;                               // element e = p * nb + j  <->  row p, columns c0 + 2j, c0 + 2j + 1
    for (int tile_id = blockIdx.x; tile_id < total_tiles; tile_id += gridDim.x) {
        const int plane = tile_id / tiles;
        const int c0 = (tile_id - plane * tiles) * tc;
        const float *src = planes + (long)plane * H * W;
        float lo = 0.f, scale = 1.f;
        if (normalize) {
            const int img = plane / planes_per_image;
            lo = pb_ord2f(mm[2 * img]);
            scale = pb_ord2f(mm[2 * img + 1]) - lo;
        }
        const float inv = 1.f / scale;
        for (int e = threadIdx.x; e < work; e += LONG_NT) {
            const int p = e >> lognb, c = c0 + 2 * (e & (nb - 1));
            float a = c < W ? src[(long)p * W + c] : 0.f, b = c + 1 < W ? src[(long)p * W + c + 1] : 0.f;
            if (normalize) {
                a = norm01(a, lo, scale, inv);
                b = norm01(b, lo, scale, inv);
            }
            s[e] = make_float2(a, b);
        }
        __syncthreads();
        pbfft::spectral_derivative(s, plan, lognb);
        if (MODE == 0) {
            float *dst = gy + (long)plane * H * W;
            for (int e = threadIdx.x; e < work; e += LONG_NT) {
                const int p = e >> lognb, c = c0 + 2 * (e & (nb - 1));
                const float2 v = s[e];
                if (c < W) dst[(long)p * W + c] = v.x;
                if (c + 1 < W) dst[(long)p * W + c + 1] = -v.y;
            }
            __syncthreads();
        } else {
            // m_k = max |cos(t_k) gx - sin(t_k) gy|, t_k = k pi / n_angles  (blur_estimation.py:129-133); gradients are
            // zeroed under the saturation mask (blur_estimation.py:117-118): they cannot raise a maximum
            float best[PB_MAX_ANGLES];
#pragma unroll
            for (int k = 0; k < PB_MAX_ANGLES; ++k) best[k] = 0.f;
            const float *gxp = gx + (long)plane * H * W;
            for (int e = threadIdx.x; e < work; e += LONG_NT) {
                const int p = e >> lognb, c = c0 + 2 * (e & (nb - 1));
                const float2 vv = s[e];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (c + h >= W) continue;
                    const long idx = (long)p * W + c + h;
                    if (discard_sat && src[idx] > sat_threshold) continue;
                    const float dx = gxp[idx];
                    const float dy = h ? -vv.y : vv.x;
#pragma unroll
                    for (int k = 0; k < PB_MAX_ANGLES; ++k)
                        if (k <= n_angles) best[k] = fmaxf(best[k], pbfft::dir_abs(ang.cs[k], ang.sn[k], dx, dy));
                }
            }
            __syncthreads();
            reduce_maxima<LONG_NT>(best, red, mags + (long)plane * PB_MAX_ANGLES * tiles_pad + (tile_id - plane * tiles), tiles_pad,
                                   n_angles);
            __syncthreads();                                   // red and the slot are free for the next tile
        }
    }
}

// ------------------------------------------------------------------------------------
// parameters and kernels
// ------------------------------------------------------------------------------------
__device__ float block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// The ker_size x ker_size Gaussian of (theta, sigma, rho) in the 25 x 25 record (blur_estimation.py:189-232), normalised:
// into sk (LDS) and, where given, into the record's taps in global memory.  Called by all NT threads of a block (barriers
// inside; sk is NOT yet visible to the other threads on return).
__device__ __forceinline__ void gaussian_taps(float *sk, float *gk, float theta, float sg, float rh, int ksize, int shift, float *red) {
    const int tid = threadIdx.x;
    const float th = -theta;
    const float c = cosf(th), s = sinf(th);
    const float i1 = 1.f / (sg * sg), i2 = 1.f / (rh * rh);
    const float a00 = c * c * i1 + s * s * i2;
    const float a01 = s * c * (i1 - i2);
    const float a11 = c * c * i2 + s * s * i1;
    float e[3];
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int idx = tid + q * NT;
        e[q] = 0.f;
        if (idx < PB_KSIZE * PB_KSIZE) {
            const int iy = idx / PB_KSIZE - PB_KRAD, ix = idx % PB_KSIZE - PB_KRAD;
            // A ker_size x ker_size kernel (blur_estimation.py:222) sits in the 25 x 25 record whose tap (iy, ix)
            // multiplies the sample (iy, ix) away from the output.  Odd sizes: centred, |offset| <= ker_size / 2.
            // EVEN sizes: the reference's grid arange(k) - (k - 1) // 2 = -k/2+1 .. k/2 is off-centre, and where the
            // taps land differs by method -- F.conv2d's 'same' padding (filters.py:46) puts k/2 - 1 samples in front
            // and k/2 behind and correlates: tap G(u) at offset u = -k/2+1 .. k/2; 'fft' rolls the kernel array by
            // k // 2 and convolves (filters.py:268-273): the same offsets, but entry i sits at offset k/2 - i, i.e.
            // tap G(1 - u) = G(u - 1) at offset u -- the Gaussian centred on offset +1 (shift).
            const int lo = (ksize & 1) ? -(ksize / 2) : -(ksize / 2) + 1, hi = ksize / 2;
            const float Y = (float)(iy - shift), X = (float)(ix - shift);
            const float quad = (X * a00 + Y * a01) * X + (X * a01 + Y * a11) * Y;
            e[q] = (iy >= lo && iy <= hi && ix >= lo && ix <= hi) ? expf(-0.5f * quad) : 0.f;
            part += e[q];
        }
    }
    const float total = block_sum(part, red);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int idx = tid + q * NT;
        if (idx < PB_KSIZE * PB_KSIZE) { sk[idx] = e[q] / total; if (gk) gk[idx] = sk[idx]; }
    }
}

// Fills kernel (unless from_taps), marginals, autocorrelations, separability and radius of one
// record.  Called by all NT threads of a block.
// shift: an EVEN ker_size under the wrap boundary ('fft') -- see the tap formula below.
__device__ void finish_record(pb_blur_info *info, int support, bool from_taps, float *red, int ksize,
                              const float *par = nullptr, int shift = 0, RecLds *rl = nullptr) {
    // Everything is derived in LDS from the taps; the record in global memory is only written (a dependent chain of
    // global round trips made this single-workgroup kernel the longest latency of small calls).
    __shared__ float sk[PB_KSIZE * PB_KSIZE], skx[PB_KSIZE], sky[PB_KSIZE];
    __shared__ int nz[PB_KSIZE], s_radius, s_first, s_sep;
    const int tid = threadIdx.x;
    __syncthreads();                                    // (a previous call's readers of the shared arrays are done)
    if (!from_taps) {
        // par (LDS): theta, sigma, rho as the caller has just computed them -- no round trip through the global record
        gaussian_taps(sk, info->kernel, par ? par[0] : info->theta, par ? par[1] : info->sigma, par ? par[2] : info->rho, ksize, shift, red);
    } else {
        for (int idx = tid; idx < PB_KSIZE * PB_KSIZE; idx += NT) sk[idx] = info->kernel[idx];
    }
    __syncthreads();
    PB_PT(4);
    float sx = 0.f, sy = 0.f;
    if (tid < PB_KSIZE) {
        int any = 0;
#pragma unroll
        for (int i = 0; i < PB_KSIZE; ++i) {                 // unrolled: the LDS reads are in flight together
            const float cv = sk[i * PB_KSIZE + tid], rv = sk[tid * PB_KSIZE + i];
            sx += cv;                                   // column sum -> kx[tid]
            sy += rv;                                   // row sum    -> ky[tid]
            any |= (cv != 0.f) | (rv != 0.f);
        }
        nz[tid] = any;
        skx[tid] = sx;
        sky[tid] = sy;
    }
    __syncthreads();
    // Symmetrise the marginals (a Gaussian's are symmetric up to the rounding of the sums above):
    // the rank-1 stencil body keeps only taps 0..12 of each in scalar registers.
    float sxm = 0.f, sym = 0.f;
    if (tid < PB_KSIZE) {
        sxm = 0.5f * (skx[tid] + skx[PB_KSIZE - 1 - tid]);
        sym = 0.5f * (sky[tid] + sky[PB_KSIZE - 1 - tid]);
    }
    __syncthreads();
    // The autocorrelations of the projections AS THEY ARE (edgetaper.py:11-21 transforms torch.sum(kernel, -1) itself): an even
    // ker_size sits off-centre in the record (gaussian_taps), its projections are not symmetric about tap 12 -- an
    // autocorrelation does not care where the taps sit, a symmetrised marginal does.  (Until round 6 the symmetrised
    // marginals were correlated, and edgetaping with an even ker_size was refused.)
    if (tid < PB_KSIZE) {
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int n = 0; n < PB_KSIZE; ++n) {
            if (n + tid < PB_KSIZE) {
                ax += skx[n] * skx[n + tid];
                ay += sky[n] * sky[n + tid];
            }
        }
        info->acorr_x[tid] = ax;
        info->acorr_y[tid] = ay;
    }
    __syncthreads();
    if (tid < PB_KSIZE) { skx[tid] = sxm; sky[tid] = sym; info->kx[tid] = sxm; info->ky[tid] = sym; }
    __syncthreads();
    PB_PT(5);
    for (int idx = tid; idx < (PB_KSIZE + 1) * 32; idx += NT) {
        const int y = idx >> 5, j = (idx & 31) - 3;
        const bool row = y < PB_KSIZE;                  // row 25: zeros, the taps of filler phases
        info->gtaps[idx] = (row && j >= 0 && j < PB_KSIZE) ? sk[y * PB_KSIZE + j] : 0.f;
        info->gtaps_odd[idx] = (row && j + 1 >= 0 && j + 1 < PB_KSIZE) ? sk[y * PB_KSIZE + j + 1] : 0.f;
    }
    // rank-1 residual  sum |k - ky (x) kx|  and the total mass (for arbitrary taps)
    float res = 0.f;
    for (int idx = tid; idx < PB_KSIZE * PB_KSIZE; idx += NT)
        res += fabsf(sk[idx] - sky[idx / PB_KSIZE] * skx[idx % PB_KSIZE]);
    PB_PT(6);
    const float resid = block_sum(res, red);
    PB_PT(7);
    // FULL: every tap that is not exactly 0.0f is evaluated (rows / columns whose taps all underflowed to
    // zero -- sigma below ~0.9 -- are skipped: bit-identical to evaluating them).  ADAPTIVE: the smallest
    // radius outside which both marginals carry < 1e-8 of the mass.
    const float thr = (support & 15) == PB_SUPPORT_ADAPTIVE ? 1e-8f : 0.f;
    bool live_t = false;
    if (tid < PB_KSIZE) live_t = thr > 0.f ? (fabsf(skx[tid]) >= thr || fabsf(sky[tid]) >= thr) : (nz[tid] != 0);
    const unsigned long long live_mask = __ballot(live_t);          // thread 0 reads wave 0's: lanes 0..24
    if (tid == 0) {
        // (an even-sized kernel is never taken for rank-1: that body keeps symmetrised halves of the marginals)
        s_sep = (resid < 1e-6f && !(support & PB_SUPPORT_FORCE_GENERAL) && (from_taps || (ksize & 1))) ? 1 : 0;
        info->separable = s_sep;
        int rad = 0;
        if (live_mask) {
            const int lo = __ffsll((long long)live_mask) - 1, hi = 63 - __clzll((long long)live_mask);
            rad = max(PB_KRAD - lo, hi - PB_KRAD);
        }
        s_radius = rad <= 4 ? 4 : (rad <= 6 ? 6 : (rad <= 8 ? 8 : (rad <= 10 ? 10 : PB_KRAD)));
        info->radius = s_radius;
        s_first = 0;
    }
    __syncthreads();
    // The general stencil body walks a list of live (kernel row, window chunk) phases.  Chunk q of class R covers
    // window elements 4q .. 4q+3, which meet the taps gtaps[row][n0 .. n0+6], n0 = 4q + 12 - R.  A phase is live when
    // one of its seven taps counts under the policy: FULL -- not exactly 0.0f; ADAPTIVE -- at least 1e-10 (inside the
    // box the marginals selected, the dropped taps are the corners outside the Gaussian's ellipse: < 7e-8 in total).
    {
        __shared__ int wcount[NT / 64];
        const int R = s_radius, off = PB_KRAD - R, nq = R / 2 + 1;
        const int row = tid / 7, q = tid - row * 7;
        const float tap_thr = (support & 15) == PB_SUPPORT_ADAPTIVE ? 1e-10f : 0.f;
        bool live = false;
        if (row <= 2 * R && q < nq) {
            for (int i = 0; i < 7; ++i) {
                const int j = 4 * q + off + i - 3;          // gtaps[y][n] = k[y][n - 3]
                const float t = (j >= 0 && j < PB_KSIZE) ? sk[(row + off) * PB_KSIZE + j] : 0.f;
                live |= tap_thr > 0.f ? (fabsf(t) >= tap_thr) : (t != 0.f);
            }
        }
        // grouped by kind (inner chunks, then first chunks, then last chunks of a window row), row-major within a kind;
        // every group is padded to an even length with a filler phase (LDS row 0, the all-zero tap row 25): the
        // stencil loop is unrolled by two with the taps double-buffered in scalar registers
        const int kind = q == 0 ? 1 : (q == nq - 1 ? 2 : 0);
        const int lane = tid & 63, wave = tid >> 6;
        int before = 0, total = 0;
        for (int k = 0; k < 3; ++k) {
            const unsigned long long bal = __ballot(live && kind == k);
            __syncthreads();
            if (lane == 0) wcount[wave] = __popcll(bal);
            __syncthreads();
            int mine = total + __popcll(bal & ((1ull << lane) - 1ull)), cnt = 0;
            for (int w = 0; w < NT / 64; ++w) { if (w < wave) mine += wcount[w]; cnt += wcount[w]; }
            if (kind == k) before = mine;
            if (tid == 0) {
                if (cnt & 1) info->phase[total + cnt] = (PB_KSIZE * 32) << 16;
                info->nphase[k] = cnt + (cnt & 1);
            }
            total += cnt + (cnt & 1);
        }
        // descriptor: byte offset of the chunk in the (64 + 2R)-wide LDS tile | index of its first tap in gtaps << 16
        const int desc = ((row * (64 + 2 * R) + 4 * q) * 4) | (((row + off) * 32 + 4 * q + off) << 16);
        if (live) info->phase[before] = desc;
        if (live && before == 0) s_first = desc;         // phase[0] is always a live phase (fillers come after their group)
        __syncthreads();
        if (tid < 3) info->phase[total + tid] = total ? s_first : 0;     // harmless targets for the prefetches
        PB_PT(8);
        if (rl) { rl->taps = sk; rl->radius = s_radius; rl->nph = total; rl->separable = s_sep; }      // (uniform; the taps stay in LDS)
    }
}

__global__ __launch_bounds__(NT) void blur_params_kernel(pb_blur_info *infos, const unsigned *__restrict__ mm,
                                                         const unsigned *__restrict__ mags_u,
                                                         const float *__restrict__ wts, int n_angles, int n_interp,
                                                         float c, float b, int support, float force_theta_deg,
                                                         int tiles_per_image, int ksize, int shift,
                                                         const float2 *__restrict__ part, int blocks_per_image, unsigned *mm_out,
                                                         float *khat_1, pb_fft_sel *fsel_1, int min_phases, const PolySpec ps_1, int lean,
                                                         float *khat_2, pb_fft_sel *fsel_2, const PolySpec ps_2, int set2_from) {
    // set2_from > 0: the workgroups blockIdx.y >= set2_from form a SECOND set of spectra + selections (khat_2, fsel_2) under the
    // spec ps_2 -- what a khat_kernel launch of its own between the estimation and the pass used to build (common.h:
    // poly_want2).  They walk the same chain from the same maxima BESIDE the first set's workgroups (the chip is empty under
    // this kernel: 65 workgroups per image), so the chain to the first set's spectra is not a cycle longer.
    const bool second = set2_from > 0 && (int)blockIdx.y >= set2_from;
    const int slice_y = second ? (int)blockIdx.y - set2_from : (int)blockIdx.y;
    float *khat = second ? khat_2 : khat_1;
    pb_fft_sel *fsel = second ? fsel_2 : fsel_1;
    const PolySpec ps = second ? ps_2 : ps_1;
    // lean (PolySpec.always under full support: every image takes a one-pass form, nobody reads the record's stencil parts
    // before the kernel ends): the grid is KH_SLICES_LEAN + 1 workgroups per image.  Workgroups 0 .. KH_SLICES_LEAN - 1 walk
    // the SHORT chain -- maxima, interpolation, (theta, sigma, rho), taps, halos, choice of window, their slice of the spectrum
    // -- and write nothing of the record; workgroup KH_SLICES_LEAN forms the whole record (marginals, autocorrelations, phase
    // lists ...) beside them, off the critical path: the same taps from the same instructions.
    // (tools/params_trace.py, 4K: the chain to the stored spectrum 48.6 k shader cycles with the record in it and 16 slices,
    // 35.6 k without the record, 27.6 k with 64 slices; the record workgroup ends at 25.2 k)
    const bool lean_wg = lean && slice_y < KH_SLICES_LEAN, rec_wg = !lean_wg;
    __shared__ float red[NT / 64];
    __shared__ float s_mags[PB_MAX_ANGLES], s_interp[PB_MAX_INTERP];
    __shared__ float s_lo[NT / 64], s_hi[NT / 64];
    pb_blur_info *info = infos + blockIdx.x;
    const int na = n_angles + 1;
#ifdef PB_PT_WANT
    for (int pt_rep = 0; pt_rep < 2; ++pt_rep) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_pt_rep = pt_rep;
    __syncthreads();
#endif
    PB_PT(0);
    // q == 0 (part != nullptr): the transforms ran on the UN-normalised gray image -- the derivative is linear and kills
    // the offset, and without quantiles the clip of normalize() never acts (blur_estimation.py:92-109) -- so the
    // gray kernel's per-workgroup (min, max) are folded here, off the transforms' critical path, and the maxima are
    // divided by the range below.
    float rng_lo = 0.f, rng_hi = 0.f, rng_inv = 1.f;
    float plo = INFINITY, phi = -INFINITY;
    // (the interpolation weights of this lane's angle: requested with everything else, used two barriers later)
    float wreg[PB_MAX_ANGLES];
#pragma unroll
    for (int k = 0; k < PB_MAX_ANGLES; ++k)
        wreg[k] = ((int)threadIdx.x < n_interp && k < na) ? wts[threadIdx.x * na + k] : 0.f;
    if (part) {
        // (requested here, folded behind the maxima's barrier below: one exposed memory latency for both)
        // (batches of eight loads in flight per thread: the gray kernel leaves up to 2048 partials per image, and one load
        // per loop trip made this the longest phase of the kernel -- 10.8 k of its 39 k cycles, tools/params_trace.py)
        const float2 *mine = part + (long)blockIdx.x * blocks_per_image;
        for (int base = 0; base < blocks_per_image; base += 8 * NT) {
            float2 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * NT + (int)threadIdx.x;
                p[u] = i < blocks_per_image ? mine[i] : make_float2(INFINITY, -INFINITY);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { plo = fminf(plo, p[u].x); phi = fmaxf(phi, p[u].y); }
        }
    }
    {
        // fold the per-tile partial maxima of this image: 16 lanes per direction, every lane's loads in flight together
        __shared__ float s_part[NT];
        const int k = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        const int tiles_pad = (tiles_per_image + 3) & ~3, n4 = tiles_pad >> 2;
        float m = 0.f;
        if (k < na) {
            const uint4 *row = reinterpret_cast<const uint4 *>(mags_u + ((long)blockIdx.x * PB_MAX_ANGLES + k) * tiles_pad);
            for (int base = 0; base < n4; base += 16 * 8) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * 16 + l16;
                    v[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (i < n4) v[u] = row[i];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = 4 * (base + u * 16 + l16);          // entries past the last tile were never written
                    if (e < tiles_per_image) m = fmaxf(m, __uint_as_float(v[u].x));
                    if (e + 1 < tiles_per_image) m = fmaxf(m, __uint_as_float(v[u].y));
                    if (e + 2 < tiles_per_image) m = fmaxf(m, __uint_as_float(v[u].z));
                    if (e + 3 < tiles_per_image) m = fmaxf(m, __uint_as_float(v[u].w));
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        s_part[threadIdx.x] = m;
        if (part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                plo = fminf(plo, __shfl_xor(plo, o));
                phi = fmaxf(phi, __shfl_xor(phi, o));
            }
            if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = plo; s_hi[threadIdx.x >> 6] = phi; }
        }
        __syncthreads();
        if (part) {
            float lo = s_lo[0], hi = s_hi[0];
            for (int w = 1; w < NT / 64; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); }
            rng_lo = lo; rng_hi = hi;
            // (a constant image: the reference's 0 / 0 makes every sample NaN; the maxima of NaN samples are 0 here as there)
            rng_inv = hi > lo ? 1.f / (hi - lo) : 0.f;
            if (threadIdx.x == 0 && rec_wg) { mm_out[2 * blockIdx.x] = pb_f2ord(lo); mm_out[2 * blockIdx.x + 1] = pb_f2ord(hi); }
        }
        if (threadIdx.x < PB_MAX_ANGLES) {
            const float v = (int)threadIdx.x < na ? s_part[threadIdx.x * 16] * rng_inv : 0.f;
            s_mags[threadIdx.x] = v;
            if (rec_wg) info->mags[threadIdx.x] = v;
        }
    }
    __syncthreads();
    PB_PT(1);
    // cubic interpolation to n_interp angles (blur_estimation.py:156-157): one lane per angle
    if (threadIdx.x < PB_MAX_INTERP) {
        float v = 0.f;
        if ((int)threadIdx.x < n_interp) {
#pragma unroll
            for (int k = 0; k < PB_MAX_ANGLES; ++k)
                if (k < na) v += wreg[k] * s_mags[k];
        }
        s_interp[threadIdx.x] = v;
        if (rec_wg) info->interp[threadIdx.x] = v;
    }
    __syncthreads();
    PB_PT(2);
    __shared__ float s_par[3];
    if (threadIdx.x < 64) {
        // argmin, first minimum (:160), over the lanes of one wave: (value, index) pairs, NaN never selected
        const int lane = threadIdx.x;
        float vmin = INFINITY;
        if (lane < n_interp) { const float x = s_interp[lane]; if (x < INFINITY) vmin = x; }
        int i_min = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(vmin, o);
            const int oi = __shfl_xor(i_min, o);
            if (ov < vmin || (ov == vmin && oi < i_min)) { vmin = ov; i_min = oi; }
        }
        if (lane == 0) {
            const float step = 180.0f / (float)n_interp;
            int theta_deg = (int)((float)i_min * step);            // interpolated_thetas.long()
            if (force_theta_deg >= 0.f) {
                theta_deg = (int)force_theta_deg;
                i_min = (int)((float)theta_deg / step);
                vmin = s_interp[i_min];
            }
            const int ortho_deg = (theta_deg + 90) % 180;
            const int i_ortho = (int)((float)ortho_deg / step);
            const float m_n = vmin, m_o = s_interp[i_ortho];
            const float cc = c * c, bb = b * b;
            s_par[1] = sqrtf(fminf(fmaxf(cc / (m_n * m_n + 1e-8f) - bb, 0.09f), 16.0f));
            s_par[2] = sqrtf(fminf(fmaxf(cc / (m_o * m_o + 1e-8f) - bb, 0.09f), 16.0f));
            s_par[0] = (float)theta_deg * 3.14159274101257324f / 180.0f;
            if (rec_wg) {
                info->theta = s_par[0];
                info->sigma = s_par[1];
                info->rho = s_par[2];
                info->i_min = i_min;
                info->gray_min = part ? rng_lo : pb_ord2f(mm[2 * blockIdx.x]);
                info->gray_max = part ? rng_hi : pb_ord2f(mm[2 * blockIdx.x + 1]);
            }
        }
    }
    __syncthreads();
    PB_PT(3);
    if (lean_wg) {
        // the taps and nothing else of the record (full support: what lies outside the record's radius is exactly zero, so the
        // box mask of khat_body changes nothing; phase count and separability only price forms PolySpec.always has no use for)
        __shared__ float sk_lean[PB_KSIZE * PB_KSIZE];
        gaussian_taps(sk_lean, nullptr, s_par[0], s_par[1], s_par[2], ksize, shift, red);
        __syncthreads();
        PB_PT(4);
        const RecLds rlean{sk_lean, PB_KRAD, 0, 0};
        khat_body<KH_SLICES_LEAN>(nullptr, khat + (long)blockIdx.x * PB_KHAT_STRIDE, fsel + blockIdx.x, min_phases, slice_y, ps, &rlean, true);
        if (second && threadIdx.x == 0 && slice_y == 0) { fsel[blockIdx.x].strip = 0; fsel[blockIdx.x].pad_[1] = 0; }      // (the first set's come from the record workgroup)
        return;
    }
    RecLds rl;
    finish_record(info, support, false, red, ksize, s_par, shift, &rl);
    if (lean) {                                      // (the record workgroup of a lean grid: the facts of the selection only it knows)
        // ... among them whether the taps ARE point-symmetric, as PolySpec.always vouches (NaN parameters make taps that compare
        // unequal to themselves): looked at here, off the critical path of the short chain; pb_body_selection reports it
        bool sym = true;
        for (int i = threadIdx.x; i < PB_KSIZE * PB_KSIZE; i += NT) sym = sym && rl.taps[i] == rl.taps[PB_KSIZE * PB_KSIZE - 1 - i];
        const int all_sym = __syncthreads_and(sym);
        if (threadIdx.x == 0) {
            fsel[blockIdx.x].strip = (rl.separable != 0 && rl.radius > 8) ? 1 : 0;
            fsel[blockIdx.x].pad_[1] = all_sym ? 0 : 1;
        }
        return;
    }
    // The spectrum of the kernel just built and the image's choice of body for the reblurring passes (khat.h): the grid is
    // KH_SLICES workgroups per image, every one of which has formed the whole record above (a few microseconds of
    // redundant latency-bound work, identical values) and now forms its slice -- one launch less on every iteration's
    // critical path than a kernel of its own behind this one.
    if (khat)      // (the record as finish_record left it in LDS: no wait for its stores, no read back)
        khat_body(info, khat + (long)blockIdx.x * PB_KHAT_STRIDE, fsel + blockIdx.x, min_phases, slice_y, ps, &rl);
#ifdef PB_PT_WANT
    __syncthreads();
    }
#endif
}

// method='direct_separable': the two correlation kernels of the x-t separable approximation of the image's Gaussian
// (intent of separable_gaussian2d.cpp:91-183; include/polyblur_hip.h, pb_options.separable_approx, states the definition).
// sep[b] = the 1-D pass along the narrower axis, sep[B + b] = the oblique pass (two taps per offset: linear
// interpolation).  Both go through finish_record like caller-supplied taps, so the first is evaluated by the rank-1
// body and the second by the general body over its ~25-50 live phases instead of 175.
__global__ __launch_bounds__(NT) void sep_records_kernel(const pb_blur_info *infos, pb_blur_info *sep, int B, int support,
                                                         int ksize) {
    __shared__ float red[NT / 64];
    __shared__ float g1[PB_KSIZE], g2[PB_KSIZE], sums[2];
    const pb_blur_info *src = infos + blockIdx.x;
    pb_blur_info *r1 = sep + blockIdx.x, *r2 = sep + B + blockIdx.x;
    const int tid = threadIdx.x, rad = ksize / 2;
    const float th = -src->theta;
    const float c = cosf(th), sn = sinf(th);
    const float i1 = 1.f / (src->sigma * src->sigma), i2 = 1.f / (src->rho * src->rho);
    const float A = c * c * i1 + sn * sn * i2, Bq = sn * c * (i1 - i2), C = c * c * i2 + sn * sn * i1;
    const bool x_first = A >= C;                       // the narrower axis first: |shear| <= 1
    const float p = x_first ? A : C, o = x_first ? C : A;
    const float det = p * o - Bq * Bq;
    if (tid < PB_KSIZE) {
        const float t = (float)(tid - PB_KRAD);
        const bool in = abs(tid - PB_KRAD) <= rad;
        g1[tid] = in ? expf(-0.5f * p * t * t) : 0.f;
        g2[tid] = in ? expf(-0.5f * (det / p) * t * t) : 0.f;
    }
    for (int idx = tid; idx < PB_KSIZE * PB_KSIZE; idx += NT) { r1->kernel[idx] = 0.f; r2->kernel[idx] = 0.f; }
    __syncthreads();
    if (tid < 2) {
        const float *g = tid ? g2 : g1;
        float s = 0.f;
        for (int i = 0; i < PB_KSIZE; ++i) s += g[i];
        sums[tid] = s;
    }
    __syncthreads();
    if (tid < PB_KSIZE && abs(tid - PB_KRAD) <= rad) {
        const int i = tid - PB_KRAD;
        const float w1 = g1[tid] / sums[0], w2 = g2[tid] / sums[1];
        const float pos = (-Bq / p) * (float)i;
        const float fl = floorf(pos), f = pos - fl;
        const int m = (int)fl;
        if (x_first) r1->kernel[PB_KRAD * PB_KSIZE + tid] = w1; else r1->kernel[tid * PB_KSIZE + PB_KRAD] = w1;
        const float wa = w2 * (1.f - f), wb = w2 * f;
        // the same numbers for the one-launch x-t body (conv_xt.hip), which applies them without forming the 2-D kernels
        r2->xt_g1[tid] = w1;
        r2->xt_m[tid] = m;
        r2->xt_wa[tid] = (abs(m) <= rad) ? wa : 0.f;
        r2->xt_wb[tid] = (abs(m + 1) <= rad) ? wb : 0.f;
        if (abs(m) <= rad && wa != 0.f) {
            if (x_first) r2->kernel[tid * PB_KSIZE + PB_KRAD + m] = wa; else r2->kernel[(PB_KRAD + m) * PB_KSIZE + tid] = wa;
        }
        if (abs(m + 1) <= rad && wb != 0.f) {
            if (x_first) r2->kernel[tid * PB_KSIZE + PB_KRAD + m + 1] = wb; else r2->kernel[(PB_KRAD + m + 1) * PB_KSIZE + tid] = wb;
        }
    }
    if (tid < PB_KSIZE && abs(tid - PB_KRAD) > rad) { r2->xt_g1[tid] = 0.f; r2->xt_m[tid] = 0; r2->xt_wa[tid] = 0.f; r2->xt_wb[tid] = 0.f; }
    if (tid == 0) {
        r2->xt_first = x_first ? 1 : 0;
        r2->xt_exact_rank1 = src->separable;
        r1->theta = r2->theta = src->theta; r1->sigma = r2->sigma = src->sigma; r1->rho = r2->rho = src->rho;
        r1->i_min = r2->i_min = src->i_min; r1->gray_min = r2->gray_min = src->gray_min; r1->gray_max = r2->gray_max = src->gray_max;
    }
    __syncthreads();
    finish_record(r1, support | PB_SUPPORT_FORCE_GENERAL, true, red, ksize);   // 7 live phases: cheaper than the rank-1 body, whose y pass would multiply by 24 zero taps
    __syncthreads();
    finish_record(r2, support, true, red, ksize);
}

__global__ __launch_bounds__(NT) void make_kernels_kernel(pb_blur_info *infos, int support, int from_taps, int ksize) {
    __shared__ float red[NT / 64];
    finish_record(infos + blockIdx.x, support, from_taps != 0, red, ksize);
}

bool plan_ext(const FftPlan *pl) {                    // a plan holding radices above 16: only some kernel variants run it
    for (int i = 0; i < pl->nstage; ++i) if (pl->radix[i] > 16) return true;
    return false;
}

size_t fft_lds_bytes(const FftPlan *pl, int nb) {
    const size_t n = pl->bluestein_m ? pl->bluestein_m : pl->n;
    return n * nb * sizeof(float2);
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename K> int allow_lds(pb_ctx *ctx, K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PB_OK;
}

// choose how many complex lines a column workgroup transforms together, and with how many threads.  Base rule: the
// widest tile whose lines fit 80 KB of LDS (two 256-thread workgroups per CU).  For the estimation's default kernel
// (maxima of 7 directions, fused plan) one 512-thread workgroup with a tile twice as wide is 7 % faster when it fits:
// the same waves per CU, but half as many 128-byte lines requested per useful byte of the column segments
// (measured, 4K: 73.7 -> 68.5 us; 8 x 1080p: 132 -> 122 us; 512 threads on the narrow tile: 95 us; one 700x500 image,
// whose 44 narrow tiles already leave most CUs idle: 31.7 -> 35.9 us, hence the workgroup-count condition).
int pick_lognb(pb_ctx *ctx, const FftPlan *pl, int W, int P, bool wide_ok, int *threads, bool fixed_maxima = false) {
    const int forced = ctx->fft_lognb;
    if (fft_lds_bytes(pl, 1) > kMaxLds) {          // lines through global memory (grad_cols_long_kernel): 8-column tiles
        int lognb = 2;
        while (lognb > 0 && (2 << (lognb - 1)) >= W * 2) --lognb;
        if (threads) *threads = LONG_NT;
        return lognb;
    }
    int lognb = forced >= 0 ? forced : 3;
    while (lognb > 0 && fft_lds_bytes(pl, 1 << lognb) > 80 * 1024) --lognb;
    while (lognb > 0 && (2 << (lognb - 1)) >= W * 2) --lognb;
    int nt = NT;
    const bool wide_off = ctx->cols_wide == 0;
    if (wide_ok && !wide_off && forced < 0 && !pl->bluestein_m && pl->nstage >= 2 &&
        fft_lds_bytes(pl, 2 << lognb) <= 140 * 1024 && (long)P * (W / (4 << lognb)) >= 200 &&   // still fills the chip
        // (1080-point lines through lines_fixed.hip: two 512-thread workgroups per CU on 16-column tiles -- 69 KB each, one's
        // requests under the other's stages -- beat one 1024-thread workgroup on a 32-column tile: 32 x 1080p 174 against 198 us)
        !(fixed_maxima && pl->n == 1080 && (W % 16) == 0)) {
        ++lognb;
        nt = 512;
    }
    if (threads) *threads = nt;
    return lognb;
}

// Workgroups of a through-memory transform: one per work item up to what keeps the slots within 256 MB (at least 64)
long long_grid(long items, size_t slot_bytes) {
    long g = (long)((256UL << 20) / slot_bytes);
    g = std::max(64L, std::min(1024L, g));
    return std::max(1L, std::min(items, g));
}

// Rows: a workgroup's transform is a chain of dependent stages, so the kernel is as fast as the number of chains in
// flight.  Lines of up to 4096 samples (32 KB of LDS per two rows) fit five workgroups per CU; with many more
// workgroups than that they run 128 threads.  Longer lines are limited by LDS to two or three workgroups per CU and
// keep 256 threads (measured at 7680: 241 us per 8K image against 339 us with 128).
#ifdef PB_EXPERIMENTAL
// The directional maxima of (gx, gy) planes (blur_estimation.py:122-134, under the saturation mask :117-118) as a pass of
// its own: what grad_cols_kernel<1> folds in its epilogue, for the estimation of a single small batch whose row and column
// transforms run side by side on two streams (pb_estimate_impl).  One partial per workgroup in the layout of the column
// tiles' partials (blur_params_kernel folds them); the same products and the same maximum: bit-identical results.
template <int NA>
__global__ __launch_bounds__(NT) void dir_maxima_kernel(const float *__restrict__ gx, const float *__restrict__ gy,
                                                         const float *__restrict__ gray, long HW, int bpp, unsigned *__restrict__ mags,
                                                         int tiles_pad, int n_angles, int discard_sat, float thr, AngleTable ang) {
    __shared__ float red[(NT / 64) * PB_MAX_ANGLES];
    const int plane = blockIdx.x / bpp, blk = blockIdx.x - plane * bpp;
    const float *px = gx + (long)plane * HW, *py = gy + (long)plane * HW, *pg = gray + (long)plane * HW;
    float best[PB_MAX_ANGLES];
#pragma unroll
    for (int k = 0; k < PB_MAX_ANGLES; ++k) best[k] = 0.f;
    const int na = n_angles + 1;
    auto fold = [&](float dx, float dy, float g) {
        if (discard_sat && g > thr) return;
#pragma unroll
        for (int k = 0; k < (NA ? NA : PB_MAX_ANGLES); ++k)
            if (NA || k < na) best[k] = fmaxf(best[k], pbfft::dir_abs(ang.cs[k], ang.sn[k], dx, dy));
    };
    if ((HW & 3) == 0) {
        const long n4 = HW >> 2;
        for (long i = (long)blk * NT + threadIdx.x; i < n4; i += (long)bpp * NT) {
            const float4 a = reinterpret_cast<const float4 *>(px)[i], b = reinterpret_cast<const float4 *>(py)[i];
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (discard_sat) g = reinterpret_cast<const float4 *>(pg)[i];
            fold(a.x, b.x, g.x); fold(a.y, b.y, g.y); fold(a.z, b.z, g.z); fold(a.w, b.w, g.w);
        }
    } else {
        for (long i = (long)blk * NT + threadIdx.x; i < HW; i += (long)bpp * NT) fold(px[i], py[i], discard_sat ? pg[i] : 0.f);
    }
    reduce_maxima<NT>(best, red, mags + (long)plane * PB_MAX_ANGLES * tiles_pad + blk, tiles_pad, n_angles);
}

#endif

// How many threads the row kernels run a line pair with (launch_rows and launch_gray_rows; the comments there).
static int rows_threads(pb_ctx *ctx, const FftPlan *pl, size_t lds, long blocks) {
    const int force_nt = ctx->rows_nt;
    const bool one_round = blocks <= 256L * 5;
    if (!plan_ext(pl) && lds <= 32 * 1024 && (force_nt == 128 || (force_nt != 256 && !one_round))) return 128;
    if (plan_ext(pl) || force_nt == 512 || (force_nt == 0 && lds > 40 * 1024)) return 512;
    return 256;
}

// Which of a plan's radices its first and last stage should take -- the two that talk to global memory: the smallest one
// (up to 16) that is at least `always_from`, or at least `min_r` with n / r <= max_items butterflies per line; 0 = the
// plan's own first radix.  A function of the plan and the line length ALONE: the order of the stages decides the roundings,
// and an image gets the same bits alone and in a batch (tests/test_gpu_fullsize.py, test_gpu_round5_forms.py).
static int first_radix_for(const FftPlan *pl, int n, int max_items, int always_from, int min_r) {
    if (pl->bluestein_m || pl->nstage < 2) return 0;
    int first = pl->radix[0];
    for (int i = 1; i < pl->nstage; ++i) {
        const int r = pl->radix[i];
        if (r <= 16 && r < first && (r >= always_from || (r >= min_r && n / r <= max_items))) first = r;
    }
    return first == pl->radix[0] ? 0 : first;
}

// The row transforms' plan: of its radices (2 .. 16) the smallest whose n / r butterflies are one trip for 256 threads goes
// first.  Fewer loads and a smaller butterfly between a thread's loads and its LDS writes, every thread busy: 1920 = 16 x
// 15 x 8 is 120 butterflies of radix 16 or 240 of radix 8.  Measured per launch (one image, 256 threads): 1080p 24.2 ->
// 21.0 us (8 first), 1280-sample rows 23.0 -> 19.5 (5), 700-sample rows 16.3 -> 15.1 (7), 1024 21.0 -> 16.0 (4), 720 22.6
// -> 15.9 (3), 512 18.6 -> 12.4 (2: 256 threads loading instead of 32); batches on 128 threads: 16 x 720-sample rows 126 ->
// 93 us, 8 x 1024 57 -> 50, 32 x 1080p -- where radix 8 is a second trip -- 304 -> 307 (15 first: 289).  launch_cols has
// the column transform's rule.
static const FftPlan *rows_plan(pb_ctx *ctx, int n) {
    const FftPlan *pl = pb_get_plan(ctx, n, true);       // (the 512-thread variant holds the radices above 16)
    if (!pl || ctx->fft_first_rows == 0 || fft_lds_bytes(pl, 1) > kMaxLds) return pl;
    const int first = ctx->fft_first_rows > 0 ? ctx->fft_first_rows : first_radix_for(pl, n, 256, 17, 2);
    return first > 0 && first != pl->radix[0] ? pb_get_plan(ctx, n, true, first) : pl;
}

int launch_rows(pb_ctx *ctx, const float *planes, float *gx, int P, int H, int W, bool normalize,
                const unsigned *mm, int planes_per_image, int gx_dtype = PB_F32) {      // (gx_dtype PB_F16: __half planes behind `gx`, fixed-plan kernels only)
    const long blocks = (long)P * ((H + 1) / 2);
    const FftPlan *pl = rows_plan(ctx, W);
    if (!pl) return PB_ERR_NOMEM;
    const size_t lds = fft_lds_bytes(pl, 1);
    const pbfft::DevPlan dp = dev_plan(pl);
    if (lds > kMaxLds) {                            // the line buffer in global memory, one slot per workgroup
        if (gx_dtype != PB_F32) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "row transform: fp16 planes out of the compiled-plan kernels only");
        if (!pb_fft_length_supported(W)) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "image width %d: lines of up to %d samples are supported", W, kMaxLineLength);
        const long grid = long_grid(blocks, lds);
        float2 *slots = static_cast<float2 *>(pb_scratch(ctx, "fft.long", (size_t)grid * lds));
        if (!slots) return PB_ERR_NOMEM;
        ProfScope prof(ctx, PB_PROF_GRAD_ROWS);
        hipLaunchKernelGGL(grad_rows_long_kernel, dim3((unsigned)grid), dim3(LONG_NT), 0, ctx->stream, planes, gx, H, W,
                           normalize ? 1 : 0, mm, planes_per_image, dp, slots, blocks);
        PB_LAUNCH_CHECK();
        return PB_OK;
    }
    const bool fused = !pl->bluestein_m && pl->nstage >= 2;          // as pbfft::fused_plan
    ProfScope prof(ctx, PB_PROF_GRAD_ROWS);
#define PB_ROWS(NTH, FUSED)                                                                                      \
    do {                                                                                                         \
        int rc = allow_lds(ctx, grad_rows_kernel<NTH, FUSED>, lds);                                              \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((grad_rows_kernel<NTH, FUSED>), dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream,  \
                           planes, gx, H, W, normalize ? 1 : 0, mm, planes_per_image, dp);                       \
    } while (0)
    // A grid that fits the chip in ONE round of five workgroups per CU (a single 4K image: 1080) is a race of dependent
    // chains, and 256 threads -- one butterfly per thread and stage, held to 96 registers so that five such workgroups
    // still fit a CU -- shorten every chain: 41.0 -> 33.4 us at 4K, 17.4 -> 15.8 us at 700 x 500.  Larger grids are
    // throughput-bound and keep 128 threads (8 x 1080p: 63.7 us against 70.0 with 256).
    const int nth = rows_threads(ctx, pl, lds, blocks);
    // (line lengths whose plan is compiled in -- lines_fixed.hip: 3840, 1920, 7680 --: the same butterflies in a one-plan kernel
    // on a padded LDS line; PB_ROWS_FIXED=0: this file's kernel, the tests' reference)
    if (fused && !normalize && ctx->rows_fixed) {
        const int rcf = pb_launch_rows_fixed(ctx, planes, 0, nullptr, gx, nullptr, P, H, W, nth, pl, gx_dtype);
        if (rcf != PB_ERR_UNSUPPORTED) return rcf;
    }
    if (gx_dtype != PB_F32) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "row transform: fp16 planes out of the compiled-plan kernels only");
    if (!fused) PB_ROWS(256, false);
    else if (nth == 128) PB_ROWS(128, true);
    else if (nth == 512) PB_ROWS(512, true);
    else PB_ROWS(256, true);
#undef PB_ROWS
    PB_LAUNCH_CHECK();
    return PB_OK;
}

// gray + range partials + row transform in one launch (gray_rows_kernel); PB_ERR_UNSUPPORTED where the lines do not take the
// fused transform (the caller then runs the gray pass and launch_rows).  *partials = partials per image.
int launch_gray_rows(pb_ctx *ctx, const void *in, int dtype, int B, int C, int H, int W, float *gray, float *gx, float2 **part,
                     int *partials) {
    const int on = ctx->est_gray_rows;
    if (!on) return PB_ERR_UNSUPPORTED;
    const FftPlan *pl = rows_plan(ctx, W);
    if (!pl) return PB_ERR_NOMEM;
    const size_t lds = fft_lds_bytes(pl, 1);
    if (lds > kMaxLds || pl->bluestein_m || pl->nstage < 2) return PB_ERR_UNSUPPORTED;
    // Measured (same box, PB_EST_GRAY_ROWS=0/1): 4K fp32 0.820 -> 0.811 ms per call, 32 x 1080p fp32 7.11 -> 7.00 ms; but 8K
    // lines (two workgroups per CU: six load streams with nothing to hide them behind) 4.22 -> 4.33 ms, and fp16 / 8-bit
    // planes (2- and 1-byte loads where the gray pass reads 8 and 4 bytes per lane) 35.0 -> 35.4 ms for 64 x 1080p fp16: the
    // fused launch is built for fp32 planes and taken for lines of up to 4096 samples (PB_EST_GRAY_ROWS=2: any length).
    if (dtype != PB_F32 || (on < 2 && lds > 32 * 1024)) return PB_ERR_UNSUPPORTED;
    const int pairs = (H + 1) / 2;
    const long blocks = (long)B * pairs;
    if (blocks > 0x7fffffffL) return PB_ERR_UNSUPPORTED;
    float2 *pt = static_cast<float2 *>(pb_scratch(ctx, "est.part", sizeof(float2) * (size_t)blocks));
    if (!pt) return PB_ERR_NOMEM;
    const pbfft::DevPlan dp = dev_plan(pl);
    ProfScope prof(ctx, PB_PROF_GRAD_ROWS);
#define PB_GROWS(T, CC, NTH)                                                                                       \
    do {                                                                                                           \
        int rc = allow_lds(ctx, gray_rows_kernel<T, CC, NTH>, lds);                                                \
        if (rc) return rc;                                                                                         \
        hipLaunchKernelGGL((gray_rows_kernel<T, CC, NTH>), dim3((unsigned)blocks), dim3(NTH), lds, ctx->stream,    \
                           static_cast<const T *>(in), gray, gx, pt, C, H, W, dp);                                 \
    } while (0)
#define PB_GROWS_C(T, NTH) do { if (C == 3) PB_GROWS(T, 3, NTH); else PB_GROWS(T, 0, NTH); } while (0)
#define PB_GROWS_T(NTH) PB_GROWS_C(float, NTH)
    // (as launch_rows; lines above 40 KB of LDS -- 8K rows -- leave room for two or three workgroups per CU: 512 threads each
    // keep the CU's SIMDs supplied)
    const int nth = rows_threads(ctx, pl, lds, blocks);
    if (ctx->rows_fixed && (C == 3 || C == 1)) {                       // (as launch_rows)
        const int rcf = pb_launch_rows_fixed(ctx, static_cast<const float *>(in), C, gray, gx, pt, B, H, W, nth, pl);
        if (rcf != PB_ERR_UNSUPPORTED) {
            if (rcf == PB_OK) { *part = pt; *partials = pairs; }
            return rcf;
        }
    }
    if (nth == 128) PB_GROWS_T(128);
    else if (nth == 512) PB_GROWS_T(512);
    else PB_GROWS_T(256);
#undef PB_GROWS_T
#undef PB_GROWS_C
#undef PB_GROWS
    PB_LAUNCH_CHECK();
    *part = pt; *partials = pairs;
    return PB_OK;
}

int launch_cols(pb_ctx *ctx, const float *planes, const float *gx, float *gy, int P, int H, int W, int mode,
                bool normalize, const unsigned *mm, int planes_per_image, unsigned *mags, int n_angles,
                int discard_sat, int gy_dtype = PB_F32) {                           // (gy_dtype PB_F16: as launch_rows)
    // (the 1024-thread variants of the gy-writing and the 7-direction kernels exist with the radices above 16)
    const bool ext_variant = mode == 0 || (mode == 1 && n_angles == 6);
    const FftPlan *pl = pb_get_plan(ctx, H, ext_variant);
    if (!pl) return PB_ERR_NOMEM;
    const bool ext = plan_ext(pl);
    int nt = NT;
    const int lognb = pick_lognb(ctx, pl, W, P, (mode == 1 && n_angles == 6) || mode == 0, &nt, mode == 1 && n_angles == 6 && !normalize && ctx->cols_fixed);
    const size_t lds = fft_lds_bytes(pl, 1 << lognb);
    const bool through_memory = fft_lds_bytes(pl, 1) > kMaxLds;
    // Which of the plan's radices the first and the last stage take -- the two that talk to global memory, the last one with
    // the epilogue's operands (gx, the gray sample) prefetched for every output of its butterfly and the directional maxima
    // folded behind it.  The SMALLEST: fewer values per thread between the loads and the butterfly, fewer operands held for
    // the epilogue (the 1024-thread variants have 128 registers per thread), and trips over the workgroup's threads that
    // are full -- 2160 = 16 x 15 x 9 in tiles of 16 columns is 1080 butterflies of radix 16 for 1024 threads, a second
    // trip for 56 of them, or 1920 of radix 9.  Measured per launch, greedy order -> smallest first: 4K 48.5 -> 43.3 us
    // (9; 15 first: 45.2), 1080p 40.7 -> 32.0 (6; 12 first: 36.5), 1440p 50.5 -> 45.3 (6), 1920 x 1200 41.1 -> 35.2 (5),
    // 1920 x 1280 40.0 -> 35.9 (5), 720p 28.3 -> 26.9 (3), 500 x 700 31.1 -> 28.4 (7), 2048^2 33.7 -> 33.1 (8);
    // 32 x 1080p 394 -> 323 us.  Radices below 5 only for lines of up to 256 of their butterflies (every trip is a round
    // trip to memory): 512 x 512 24.6 -> 20.8 us with 2 first, 3000-sample columns -- twelve trips of radix 2 -- no different.
    // A function of the line length alone, as rows_plan's.
    if (!through_memory && ctx->fft_first != 0) {
        const int first = ctx->fft_first > 0 ? ctx->fft_first : first_radix_for(pl, H, 256, 5, 2);
        if (first > 0 && first != pl->radix[0]) {
            pl = pb_get_plan(ctx, H, ext_variant, first);
            if (!pl) return PB_ERR_NOMEM;
        }
    }
    if (through_memory && !pb_fft_length_supported(H))
        return pb_fail(ctx, PB_ERR_UNSUPPORTED, "image height %d: lines of up to %d samples are supported", H, kMaxLineLength);
    const int tc = 2 << lognb;
    const long blocks = (long)P * ((W + tc - 1) / tc);
    if (blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "column transform: bad grid");
    const pbfft::DevPlan dp = dev_plan(pl);
    const float thr = 0.99f;
    AngleTable ang;
    for (int k = 0; k < PB_MAX_ANGLES; ++k) {
        const float t = n_angles > 0 ? 3.14159265358979323846f * (float)k / (float)n_angles : 0.f;
        ang.cs[k] = std::cos(t);
        ang.sn[k] = std::sin(t);
    }
    if (through_memory) {                           // as launch_rows: a slot of 2 << lognb columns per workgroup
        const long grid = long_grid(blocks, lds);
        float2 *slots = static_cast<float2 *>(pb_scratch(ctx, "fft.long", (size_t)grid * lds));
        if (!slots) return PB_ERR_NOMEM;
        ProfScope prof(ctx, PB_PROF_GRAD_COLS);
        if (mode == 0)
            hipLaunchKernelGGL(grad_cols_long_kernel<0>, dim3((unsigned)grid), dim3(LONG_NT), 0, ctx->stream, planes, gx, gy, H, W, lognb,
                               normalize ? 1 : 0, mm, planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang, slots);
        else
            hipLaunchKernelGGL(grad_cols_long_kernel<1>, dim3((unsigned)grid), dim3(LONG_NT), 0, ctx->stream, planes, gx, gy, H, W, lognb,
                               normalize ? 1 : 0, mm, planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang, slots);
        PB_LAUNCH_CHECK();
        return PB_OK;
    }
    ProfScope prof(ctx, PB_PROF_GRAD_COLS);
    // the line lengths whose plan is compiled in (lines_fixed.hip: 2160, 1080, 4320, on every tile width pick_lognb gives them):
    // the same arithmetic in a kernel that holds one plan -- bit-identical maxima (PB_COLS_FIXED=0: the run-time-plan kernel,
    // the tests' reference)
    if (mode == 1 && !normalize && ctx->cols_fixed) {
        const int rcf = pb_launch_cols_fixed(ctx, planes, gx, P, H, W, lognb, mags, n_angles, discard_sat, pl);
        if (rcf != PB_ERR_UNSUPPORTED) return rcf;
    }
    if (mode == 0 && !normalize && ctx->cols_fixed && !through_memory) {
        const int rcf = pb_launch_cols_fixed(ctx, planes, nullptr, P, H, W, lognb, nullptr, 0, 0, pl, gy, gy_dtype);
        if (rcf != PB_ERR_UNSUPPORTED) return rcf;
    }
    if (gy_dtype != PB_F32) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "column transform: fp16 planes out of the compiled-plan kernels only");
#define PB_COLS(MODE, NA)                                                                                        \
    do {                                                                                                         \
        int rc = allow_lds(ctx, grad_cols_kernel<MODE, NA, NT>, lds);                                            \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((grad_cols_kernel<MODE, NA, NT>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(NT),   \
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,                 \
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);            \
    } while (0)
#define PB_COLS_EXT(MODE, NA)                                                                                    \
    do {                                                                                                         \
        int rc = allow_lds(ctx, grad_cols_kernel<MODE, NA, 1024, 24>, lds);                                      \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((grad_cols_kernel<MODE, NA, 1024, 24>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(1024), \
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,                 \
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);            \
    } while (0)
    // (a plan whose widest radix is 18 -- 4320 = 16 x 15 x 18, the 8K height -- runs the variant without 20 and 24: fewer
    // registers spilled around the butterflies it does take)
    int maxr = 0;
    for (int i = 0; i < pl->nstage; ++i) maxr = std::max(maxr, pl->radix[i]);
    if (ext && mode == 1 && maxr <= 18) {
        int rc = allow_lds(ctx, grad_cols_kernel<1, 7, 1024, 18>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((grad_cols_kernel<1, 7, 1024, 18>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(1024),
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);
    }
    else if (ext && mode == 0) PB_COLS_EXT(0, 0);
    else if (ext) PB_COLS_EXT(1, 7);
    else if (mode == 0 && nt == 512) {
        // (192 x 1080p planes: 2.27 -> 1.61 ms against 8-column tiles with 256 threads)
        int rc = allow_lds(ctx, grad_cols_kernel<0, 0, 1024>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((grad_cols_kernel<0, 0, 1024>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(1024),
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);
    }
    else if (mode == 0) PB_COLS(0, 0);
    else if (n_angles == 6 && nt == 512) {
        // 16-column tiles, one workgroup per CU: 1024 threads (16 waves) hide the stages' LDS latency better than 512
        // (measured: 4K 62 -> 55 us, 8 x 1080p 113 -> 95 us, 8K 277 -> 268 us)
        int rc = allow_lds(ctx, grad_cols_kernel<1, 7, 1024>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((grad_cols_kernel<1, 7, 1024>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(1024),
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);
    }
    else if (n_angles == 6) {
        // 8-column (or narrower) tiles, two workgroups per CU: 512 threads each (measured against 256: 700x500 31 -> 23 us,
        // 1080p 56 -> 40 us, 512x512 28 -> 25 us)
        int rc = allow_lds(ctx, grad_cols_kernel<1, 7, 512>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((grad_cols_kernel<1, 7, 512>), dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(512),
                           lds, ctx->stream, planes, gx, gy, H, W, lognb, normalize ? 1 : 0, mm,
                           planes_per_image, mags, n_angles, discard_sat, thr, (int)blocks, dp, ang);
    }
    else PB_COLS(1, 0);
#undef PB_COLS_EXT
#undef PB_COLS
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

int pb_fourier_gradients_impl(pb_ctx *ctx, const float *planes, int P, int H, int W, float *gx, float *gy) {
    return pb_fourier_gradients_typed(ctx, planes, P, H, W, gx, gy, PB_F32);
}

// out_dtype PB_F16: gx / gy are __half planes (pb_gradient_planes_half says whether this context builds them for the shape)
int pb_fourier_gradients_typed(pb_ctx *ctx, const float *planes, int P, int H, int W, void *gx, void *gy, int out_dtype) {
    if (P <= 0 || H < 2 || W < 2) return pb_fail(ctx, PB_ERR_BADARG, "fourier_gradients: bad shape");
    if (gx) { int rc = launch_rows(ctx, planes, static_cast<float *>(gx), P, H, W, false, nullptr, 1, out_dtype); if (rc) return rc; }
    if (gy) { int rc = launch_cols(ctx, planes, nullptr, static_cast<float *>(gy), P, H, W, 0, false, nullptr, 1, nullptr, 0, 0, out_dtype); if (rc) return rc; }
    return PB_OK;
}

bool pb_gradient_planes_half(pb_ctx *ctx, int H, int W) {
    return ctx->cols_fixed && ctx->rows_fixed && ctx->fft_lognb < 0 && ctx->fft_first < 0 && ctx->fft_first_rows < 0 &&
           ctx->fft_ext_radix != 0 && pb_lines_fixed_shape(H, W);
}

int pb_estimate_impl(pb_ctx *ctx, const void *in, int dtype, int B, int C, int H, int W, const pb_options *opt,
                     pb_blur_info *dev_info) {
    if (opt->q < 0.f || opt->q >= 0.5f) return pb_fail(ctx, PB_ERR_BADARG, "q must be in [0, 0.5)");
    const int ksize = pb_kernel_size(opt);
    if (!ksize) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "ker_size %d: sizes from 2 to %d are built", opt->ker_size, PB_KSIZE_MAX);
    if (opt->n_angles < 1 || opt->n_angles + 1 > PB_MAX_ANGLES || opt->n_interpolated_angles < 1 ||
        opt->n_interpolated_angles > PB_MAX_INTERP)
        return pb_fail(ctx, PB_ERR_BADARG, "n_angles / n_interpolated_angles out of range");
    const long HW = (long)H * W;
    const FftPlan *plh = pb_get_plan(ctx, H);
    if (!plh) return PB_ERR_NOMEM;
    const int est_lognb = pick_lognb(ctx, plh, W, B, opt->n_angles == 6, nullptr, opt->n_angles == 6 && !(opt->q > 0.f) && ctx->cols_fixed);   // as launch_cols
    const int col_tiles = (W + (2 << est_lognb) - 1) / (2 << est_lognb);
    // (see below: transforms side by side + a maxima pass; an experiment that measured SLOWER -- 0.90 against 0.85 ms per 4K call,
    // 0.35 against 0.32 ms at 700 x 500: the column workgroups take a CU's whole LDS, so the row workgroups do not run beside
    // them, and the maxima pass and the fork / join come on top -- and is only in the --experimental build, behind PB_EST_OVERLAP=1)
#ifdef PB_EXPERIMENTAL      // (python -m polyblur_amd.build --experimental)
    const int overlap_env = ctx->est_overlap;
    const bool lines_in_lds = pb_fft_length_supported(H) == 1 && pb_fft_length_supported(W) == 1;
    const bool overlap = ctx->aux && !ctx->prof_on && lines_in_lds && overlap_env > 0;
#else
    const bool overlap = false;
#endif
    const int est_tiles = overlap ? 512 : col_tiles;          // partial maxima per image: column tiles, or the maxima pass's workgroups
    float *gray = static_cast<float *>(pb_scratch(ctx, "est.gray", sizeof(float) * B * HW));
    float *gx = static_cast<float *>(pb_scratch(ctx, "est.gx", sizeof(float) * B * HW));
    unsigned *mm = static_cast<unsigned *>(pb_scratch(ctx, "est.mm", sizeof(unsigned) * 2 * B));
    unsigned *mags = static_cast<unsigned *>(pb_scratch(ctx, "est.mags", sizeof(unsigned) * (size_t)B * ((std::max(col_tiles, 512) + 3) & ~3) * PB_MAX_ANGLES));
    if (!gray || !gx || !mm || !mags) return PB_ERR_NOMEM;
    const float *wts = pb_get_interp_weights(ctx, opt->n_angles, opt->n_interpolated_angles);
    if (!wts) return PB_ERR_NOMEM;
    const float2 *part_q0 = nullptr;
    int bpi_q0 = 0;
    // q == 0: gray, its range and the row transform in one launch where the lines allow it
    bool rows_done = false;
    if (opt->q <= 0.f && !overlap) {
        float2 *pt = nullptr;
        const int rcg = launch_gray_rows(ctx, in, dtype, B, C, H, W, gray, gx, &pt, &bpi_q0);
        if (rcg == PB_OK) { rows_done = true; part_q0 = pt; }
        else if (rcg != PB_ERR_UNSUPPORTED) return rcg;
    }
    if (!rows_done) {
    ProfScope prof(ctx, PB_PROF_GRAY);
    const bool vec = (HW & 3) == 0;
    int bpi = (int)((HW / (vec ? 4 : 1) + NT * 4 - 1) / (NT * 4));
    const int bpi_max = (2048 + B - 1) / B;
    if (bpi > bpi_max) bpi = bpi_max;
    if (bpi < 1) bpi = 1;
    float2 *part = static_cast<float2 *>(pb_scratch(ctx, "est.part", sizeof(float2) * ((size_t)B * bpi)));
    if (!part) return PB_ERR_NOMEM;
#define PB_GRAY(T, V, CC)                                                                                           \
    hipLaunchKernelGGL((gray_minmax_kernel<T, V, CC>), dim3(B * bpi), dim3(NT), 0, ctx->stream,                     \
                       static_cast<const T *>(in), gray, part, C, HW, bpi)
#define PB_GRAY_C(T, V) do { if (C == 3) PB_GRAY(T, V, 3); else if (C == 1) PB_GRAY(T, V, 1); else PB_GRAY(T, V, 0); } while (0)
    if (dtype == PB_F32) { if (vec) PB_GRAY_C(float, true); else PB_GRAY_C(float, false); }
    else if (dtype == PB_F16) { if (vec) PB_GRAY_C(__half, true); else PB_GRAY_C(__half, false); }
    else { if (vec) PB_GRAY_C(unsigned char, true); else PB_GRAY_C(unsigned char, false); }
#undef PB_GRAY_C
#undef PB_GRAY
    part_q0 = part; bpi_q0 = bpi;
    // (q == 0: the parameter kernel folds the partials itself -- see there -- and nothing below waits for the range)
    if (opt->q > 0.f) hipLaunchKernelGGL(minmax_reduce_kernel, dim3(B), dim3(NT), 0, ctx->stream, part, mm, bpi);
    if (opt->q > 0.f) {
        // replace (min, max) by the (q, 1-q) quantiles: three histogram + scan rounds
        const size_t hbytes = sizeof(unsigned) * (size_t)B * 4 * 4096;
        unsigned *hist = static_cast<unsigned *>(pb_scratch(ctx, "est.qhist", hbytes));
        QuantSel *sel = static_cast<QuantSel *>(pb_scratch(ctx, "est.qsel", sizeof(QuantSel) * (size_t)B));
        if (!hist || !sel) return PB_ERR_NOMEM;
        int hb = (int)((HW + NT * 16 - 1) / (NT * 16));
        const int hb_max = (1024 + B - 1) / B;
        if (hb > hb_max) hb = hb_max;
        if (hb < 1) hb = 1;
        const float q_lo = opt->q, q_hi = (float)(1.0 - (double)opt->q);
        PB_HIP(hipMemsetAsync(hist, 0, hbytes, ctx->stream));
        hipLaunchKernelGGL(quant_hist_kernel<0>, dim3(B * hb), dim3(NT), 4096 * sizeof(unsigned), ctx->stream, gray, sel, hist, HW, hb);
        hipLaunchKernelGGL(quant_scan_kernel<0>, dim3(B), dim3(NT), 0, ctx->stream, hist, sel, mm, HW, q_lo, q_hi);
        PB_HIP(hipMemsetAsync(hist, 0, hbytes, ctx->stream));
        int rcq = allow_lds(ctx, quant_hist_kernel<1>, 4 * 4096 * sizeof(unsigned));
        if (rcq) return rcq;
        hipLaunchKernelGGL(quant_hist_kernel<1>, dim3(B * hb), dim3(NT), 4 * 4096 * sizeof(unsigned), ctx->stream, gray, sel, hist, HW, hb);
        hipLaunchKernelGGL(quant_scan_kernel<1>, dim3(B), dim3(NT), 0, ctx->stream, hist, sel, mm, HW, q_lo, q_hi);
        PB_HIP(hipMemsetAsync(hist, 0, hbytes, ctx->stream));
        hipLaunchKernelGGL(quant_hist_kernel<2>, dim3(B * hb), dim3(NT), 4 * 256 * sizeof(unsigned), ctx->stream, gray, sel, hist, HW, hb);
        hipLaunchKernelGGL(quant_scan_kernel<2>, dim3(B), dim3(NT), 0, ctx->stream, hist, sel, mm, HW, q_lo, q_hi);
    }
    PB_LAUNCH_CHECK();
    }
    const bool norm = opt->q > 0.f;                  // q == 0: transforms of the un-normalised image, maxima rescaled afterwards
    int rc = PB_OK;
#ifdef PB_EXPERIMENTAL
    if (overlap) {
        // (experiment, PB_EST_OVERLAP=1) both transforms are chains of dependent stages on an under-filled chip, so they are
        // issued side by side -- rows (-> gx) on the side stream, a gy-writing column pass here -- and one HBM-speed pass
        // folds the maxima.  Same results bit for bit; measured slower (above).
        float *gy = static_cast<float *>(pb_scratch(ctx, "est.gy", sizeof(float) * B * HW));
        if (!gy) return PB_ERR_NOMEM;
        PB_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        PB_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->aux;
        rc = launch_rows(ctx, gray, gx, B, H, W, norm, mm, 1);
        ctx->stream = main_stream;
        PB_HIP(hipEventRecord(ctx->ev_join, ctx->aux));
        if (!rc) rc = launch_cols(ctx, gray, nullptr, gy, B, H, W, 0, norm, mm, 1, nullptr, opt->n_angles, 0);
        PB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        if (rc) return rc;
        AngleTable ang;
        for (int k = 0; k < PB_MAX_ANGLES; ++k) {
            const float t = opt->n_angles > 0 ? 3.14159265358979323846f * (float)k / (float)opt->n_angles : 0.f;
            ang.cs[k] = std::cos(t); ang.sn[k] = std::sin(t);
        }
        ProfScope prof(ctx, PB_PROF_GRAD_COLS);
        const int tp = (est_tiles + 3) & ~3;
        if (opt->n_angles == 6)
            hipLaunchKernelGGL(dir_maxima_kernel<7>, dim3((unsigned)(B * est_tiles)), dim3(NT), 0, ctx->stream, gx, gy, gray, HW, est_tiles, mags, tp,
                               opt->n_angles, opt->discard_saturation, 0.99f, ang);
        else
            hipLaunchKernelGGL(dir_maxima_kernel<0>, dim3((unsigned)(B * est_tiles)), dim3(NT), 0, ctx->stream, gx, gy, gray, HW, est_tiles, mags, tp,
                               opt->n_angles, opt->discard_saturation, 0.99f, ang);
        PB_LAUNCH_CHECK();
    } else
#endif
    {
        if (!rows_done) rc = launch_rows(ctx, gray, gx, B, H, W, norm, mm, 1);
        if (rc) return rc;
        rc = launch_cols(ctx, gray, gx, nullptr, B, H, W, 1, norm, mm, 1, mags, opt->n_angles, opt->discard_saturation);
        if (rc) return rc;
    }
    float *khat = nullptr;
    pb_fft_sel *fsel = nullptr;
    if (ctx->fft_min_phases >= 0) {
        rc = pb_khat_buffers(ctx, B, &khat, &fsel);
        if (rc) return rc;
    }
    // (PolySpec.always under full support: the short chain to the spectra, the record beside it -- see the kernel)
    const int lean = (khat && ctx->poly_want.always == 1 && opt->support == PB_SUPPORT_FULL && ctx->est_lean) ? 1 : 0;
    // (the second set: where the call in progress has said it will need one -- api.hip sets poly_want2 around the estimation)
    float *khat2 = nullptr;
    pb_fft_sel *fsel2 = nullptr;
    const PolySpec ps2 = ctx->poly_want2;
    ctx->khat2_owner = ctx->khat2_owner == dev_info ? nullptr : ctx->khat2_owner;       // (these records are being rewritten)
    if (khat && (ps2.on != 0 || ps2.always != 0)) {
        rc = pb_khat2_buffers(ctx, B, &khat2, &fsel2);
        if (rc) return rc;
    }
    ProfScope prof(ctx, PB_PROF_PARAMS);
    const int set1 = khat ? (lean ? KH_SLICES_LEAN + 1 : KH_SLICES) : 1;
    const int set2 = khat2 ? (lean ? KH_SLICES_LEAN : KH_SLICES) : 0;
    hipLaunchKernelGGL(blur_params_kernel, dim3(B, set1 + set2), dim3(NT), 0, ctx->stream, dev_info, mm, mags, wts, opt->n_angles,
                       opt->n_interpolated_angles, opt->c, opt->b, opt->support, opt->force_theta_deg, est_tiles, ksize,
                       (!(ksize & 1) && opt->boundary == PB_WRAP) ? 1 : 0, norm ? nullptr : part_q0, bpi_q0, mm, khat, fsel,
                       ctx->fft_min_phases, ctx->poly_want, lean, khat2, fsel2, ps2, set2 ? set1 : 0);
    PB_LAUNCH_CHECK();
    if (khat2) { ctx->khat2_owner = dev_info; ctx->khat2_B = B; ctx->khat2_spec = ps2; }
    if (khat) { ctx->khat_owner = dev_info; ctx->khat_B = B; ctx->khat_by_estimate = true; ctx->poly_built = ctx->poly_want; ctx->khat_slot = ctx->sel_slot % PB_SEL_SLOTS; }
    return PB_OK;
}

int pb_kernel_size(const pb_options *opt) {
    const int k = opt->ker_size == 0 ? PB_KSIZE : opt->ker_size;
    return (k >= 2 && k <= PB_KSIZE_MAX) ? k : 0;
}

int pb_make_sep_records(pb_ctx *ctx, int B, const pb_blur_info *dev_info, pb_blur_info *sep, int support, int ksize) {
    ProfScope prof(ctx, PB_PROF_PARAMS);
    hipLaunchKernelGGL(sep_records_kernel, dim3(B), dim3(NT), 0, ctx->stream, dev_info, sep, B, support, ksize);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_make_kernels_dev(pb_ctx *ctx, int B, pb_blur_info *dev_info, int support, int from_taps, int ksize) {
    ProfScope prof(ctx, PB_PROF_PARAMS);
    hipLaunchKernelGGL(make_kernels_kernel, dim3(B), dim3(NT), 0, ctx->stream, dev_info, support, from_taps, ksize);
    PB_LAUNCH_CHECK();
    return PB_OK;
}
