// Dense (non rank-1) reblurring kernels evaluated per tile in the frequency domain.
//
// Same pass, same operands and same result as the general body of conv.hip -- one Horner step  t <- K*t + coef*x  of the
// polynomial deconvolution (reference deblurring.py:122-138 / :141-169), or one edgetaper blend (edgetaper.py:30-32), on
// the replicate-padded domain with the reference's boundary models (filters.py:14-49) -- but the exact 2-D stencil of a
// workgroup's tile is evaluated as a CIRCULAR correlation of a 64 x 64 window held in LDS: forward 2-D DFT, product with
// the kernel's 64 x 64 spectrum, inverse DFT, of which only the samples at least R from the window's edge (those no tap
// reaches around the wrap) are kept -- overlap-save.  Nothing leaves LDS between the two transforms; HBM sees what the
// stencil pass sees (a window read, the x operand, one store).  The 625 multiply-adds per sample of the dense stencil
// (fp32-vector-bound: 0.30 ms per 4K launch) become ~110 flops per sample, and the pass is bound by LDS and memory.
//
// * Two real tiles ride one complex transform: z = A + iB for two horizontally adjacent windows.  The kernel is real,
//   so K (*) z = K (*) A + i K (*) B -- no real-to-complex packing or unpacking step exists.
// * 64 = 8 x 8: every 1-D pass is two radix-8 stages (pbfft::dft_small<8>, complex values in aligned register pairs,
//   packed adds).  Decimation in frequency forward -- position 8a + b of a transformed axis holds frequency a + 8b --
//   and the mirrored stages backward, so no reordering pass exists either; the spectrum of the kernel is laid out in
//   the same permuted order by khat_kernel.
// * Stage order: columns (stage 1 straight from global memory, stage 2), rows (stage 1, stage 2 x spectrum x inverse
//   stage 2 in registers), rows^-1 stage 1, columns^-1 (stage 2, stage 1 straight into the epilogue and global memory).
//   Six LDS round trips per window pair.  The first and last stages touch global memory with 32 consecutive window
//   columns per half wave (128-byte segments).
// * LDS rows are 65 complex values long: every row-direction access (stride 8 or contiguous 8 per lane) and every
//   column-direction access is bank-conflict-free for ds_read_b64 / ds_write_b64.
// * Each thread keeps ONE set of seven inter-stage twiddles W64^(n2 k) for the whole kernel (its n2 = 2 wave + half).
//
// The window keeps R = 4, 8 or 12 samples of halo (the record's radius class rounded up to a multiple of 4: 16-byte
// aligned windows), i.e. 56, 48 or 40 outputs per side; every image picks its own on the device, like its body.
// No MFMA, no library FFT; the transforms exist only inside a workgroup's LDS.

#include "common.h"
#include "conv_common.h"
#include "fft.h"

namespace {

using pbfft::cf;

constexpr int FT_N = 64;          // window side
constexpr int FT_P = 65;          // LDS row pitch in complex values
constexpr int FT_NT = 256;
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

// inverse 8-point DFT (unnormalised): the forward one with its outputs read in mirrored order
__device__ __forceinline__ void idft8(cf (&v)[8]) {
    pbfft::dft_small<8>(v);
    cf t;
    t = v[1]; v[1] = v[7]; v[7] = t;
    t = v[2]; v[2] = v[6]; v[6] = t;
    t = v[3]; v[3] = v[5]; v[5] = t;
}

// ---------------------------------------------------------------------------------------------
// spectrum of every image's kernel (one workgroup per image)
// ---------------------------------------------------------------------------------------------
// khat[py][px] = (1/4096) sum_{u,v} k[u][v] exp(+2 pi i (fy (u-12) + fx (v-12)) / 64),  f = (p >> 3) + 8 (p & 7):
// the conjugate spectrum (the pass is a correlation, like the stencil bodies), in the transforms' permuted order, with
// both transforms' normalisation folded in.  Only the taps inside the record's support box count, exactly as in the
// stencil body.  Accumulated in double (the spectrum is then the correctly rounded fp32 one).
__global__ __launch_bounds__(FT_NT) void khat_kernel(const pb_blur_info *infos, float2 *khat, pb_fft_sel *sel, int min_phases) {
    __shared__ double2 G[PB_KSIZE * FT_N];
    __shared__ double cs[FT_N], sn[FT_N];
    __shared__ float sk[PB_KSIZE * PB_KSIZE];
    const pb_blur_info *info = infos + blockIdx.x;
    const int tid = threadIdx.x;
    const int nph = info->nphase[0] + info->nphase[1] + info->nphase[2];
    const int R = info->radius;
    const bool use = info->separable == 0 && nph >= min_phases && min_phases >= 0;
    if (tid == 0) { sel[blockIdx.x].use_fft = use ? 1 : 0; sel[blockIdx.x].rf = R <= 4 ? 4 : (R <= 8 ? 8 : 12); }
    if (!use) return;
    if (tid < FT_N) { double s, c; sincospi((double)tid / 32.0, &s, &c); cs[tid] = c; sn[tid] = s; }
    for (int i = tid; i < PB_KSIZE * PB_KSIZE; i += FT_NT) {
        const int u = i / PB_KSIZE - PB_KRAD, v = i % PB_KSIZE - PB_KRAD;
        sk[i] = (abs(u) <= R && abs(v) <= R) ? info->kernel[i] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < PB_KSIZE * FT_N; idx += FT_NT) {
        const int u = idx >> 6, px = idx & 63, fx = (px >> 3) + 8 * (px & 7);
        double ar = 0.0, ai = 0.0;
        for (int v = 0; v < PB_KSIZE; ++v) {
            const int m = (fx * (v - PB_KRAD)) & 63;
            const double k = (double)sk[u * PB_KSIZE + v];
            ar += k * cs[m]; ai += k * sn[m];
        }
        G[idx] = make_double2(ar, ai);
    }
    __syncthreads();
    float2 *out = khat + (long)blockIdx.x * (FT_N * FT_N);
    for (int idx = tid; idx < FT_N * FT_N; idx += FT_NT) {
        const int py = idx >> 6, px = idx & 63, fy = (py >> 3) + 8 * (py & 7);
        double ar = 0.0, ai = 0.0;
        for (int u = 0; u < PB_KSIZE; ++u) {
            const int m = (fy * (u - PB_KRAD)) & 63;
            const double2 g = G[u * FT_N + px];
            ar += g.x * cs[m] - g.y * sn[m];
            ai += g.x * sn[m] + g.y * cs[m];
        }
        out[idx] = make_float2((float)(ar * (1.0 / 4096.0)), (float)(ai * (1.0 / 4096.0)));
    }
}

// ---------------------------------------------------------------------------------------------
// the pass
// ---------------------------------------------------------------------------------------------
// one sample of the source at padded coordinates (py, px) under the pass's boundary model
template <typename T>
__device__ __forceinline__ float load_elem(const ConvPass &a, const T *plane, int py, int px) {
    const int iy = map_axis(py, a.H, a.in_kind, a.boundary, a.pad), ix = map_axis(px, a.W, a.in_kind, a.boundary, a.pad);
    if ((iy | ix) < 0) return 0.f;
    return pb_ld(plane + (long)iy * a.in_pitch + ix);
}
// the x operand of the output at padded (py, px): rows and columns of a virtual source clamp
template <typename TX>
__device__ __forceinline__ float load_x1(const ConvPass &a, const TX *xpl, int py, int px) {
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int xr = virt ? min(max(py - a.pad, 0), a.H - 1) : py;
    const int xc = virt ? min(max(px - a.pad, 0), a.W - 1) : px;
    return pb_ld(xpl + (long)xr * a.x_pitch + xc);
}
// epilogue + store of one output (the caller has checked that (py, px) lies in the output region)
template <typename TOut>
__device__ __forceinline__ void finish1(const ConvPass &a, const pb_blur_info *info, TOut *opl, int py, int px, float acc, float xv) {
    float v;
    if (a.epilogue == EPI_TAPER) {
        const float al = taper_weight(info->acorr_y, py, a.H + 2 * a.pad) * taper_weight(info->acorr_x, px, a.W + 2 * a.pad);
        v = al * xv + (1.f - al) * acc;
    } else {
        v = a.scale * acc + a.coef * xv;
    }
    if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    pb_st(opl + (long)(py - oo) * a.out_pitch + (px - oo), v);
}

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(FT_NT, 4) void conv_fft_kernel(const ConvPass a, int jobs_per_plane, int total_jobs) {
    extern __shared__ __attribute__((aligned(16))) float2 Z[];
    // XCD-aware order (see conv_tile_kernel): every XCD gets one contiguous run of window pairs
    const int chunk = gridDim.x >> 3;
    const int job = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (job >= total_jobs) return;
    const int plane = __builtin_amdgcn_readfirstlane(job / jobs_per_plane);
    const int local = job - plane * jobs_per_plane;
    const int img = __builtin_amdgcn_readfirstlane(plane / a.C);
    const pb_fft_sel sel = a.fsel[img];
    if (!sel.use_fft) return;                                   // a stencil body of conv_tile_kernel does this image
    const pb_blur_info *info = a.info + img;
    const int R = sel.rf, T = FT_N - 2 * R;
    const OutRegion rg = out_region(a);
    const int tiles_x = (rg.x_hi - rg.x_lo + T - 1) / T, pairs_x = (tiles_x + 1) >> 1, tiles_y = (rg.y_hi - rg.y_lo + T - 1) / T;
    if (local >= pairs_x * tiles_y) return;                     // the grid is sized for the smallest tile
    const int ty = __builtin_amdgcn_readfirstlane(local / pairs_x), pxi = local - ty * pairs_x;
    const int wy0 = rg.y_lo + ty * T - R;                       // window origin, padded coordinates
    const int wxA = rg.x_lo + 2 * pxi * T - R, wxB = wxA + T;
    const bool hasB = wxB + R < rg.x_hi;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l = lane & 31, n2 = 2 * wave + (lane >> 5);       // this thread's stage-1 column of the 8 x 8 index split
    cf tw[8];
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float2 w = kW64[(n2 * k) & 63]; tw[k] = (cf){w.x, w.y}; }

    // ---- columns, stage 1, straight from global memory: window rows 8 n1 + n2 of columns l, l + 32 ----
    {
        const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
        const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
        const bool inside = wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && hasB;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int x = l + 32 * t;
            cf v[8];
            if (inside) {
                const TIn *p = ipl + (long)(wy0 - lo + n2) * a.in_pitch + (wxA - lo + x);
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) v[n1] = (cf){pb_ld(p + (long)(8 * n1) * a.in_pitch), pb_ld(p + (long)(8 * n1) * a.in_pitch + T)};
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const int py = wy0 + 8 * n1 + n2;
                    v[n1] = (cf){load_elem<TIn>(a, ipl, py, wxA + x), hasB ? load_elem<TIn>(a, ipl, py, wxB + x) : 0.f};
                }
            }
            pbfft::dft_small<8>(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const cf r = k ? pbfft::cmul(v[k], tw[k]) : v[0];
                Z[(8 * k + n2) * FT_P + x] = make_float2(r.x, r.y);
            }
        }
    }
    __syncthreads();
    // ---- columns, stage 2: rows 8 k1 .. 8 k1 + 7 of column `lane` ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float2 *p = Z + (8 * (2 * wave + t)) * FT_P + lane;
        cf v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pbfft::to_cf(p[j * FT_P]);
        pbfft::dft_small<8>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j * FT_P] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- rows, stage 1: columns 8 n1 + n2 of rows l, l + 32 ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float2 *p = Z + (l + 32 * t) * FT_P + n2;
        cf v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pbfft::to_cf(p[8 * j]);
        pbfft::dft_small<8>(v);
#pragma unroll
        for (int k = 0; k < 8; ++k) p[8 * k] = pbfft::to_f2(k ? pbfft::cmul(v[k], tw[k]) : v[0]);
    }
    __syncthreads();
    // ---- rows, stage 2 -> x spectrum -> inverse stage 2: columns 8 k1 .. 8 k1 + 7 of one row ----
    {
        const float2 *kh = a.khat + (long)img * (FT_N * FT_N);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = (lane & 7) + 8 * wave + 32 * t, k1 = lane >> 3;
            const float4 *hp = reinterpret_cast<const float4 *>(kh + row * FT_N + 8 * k1);
            float4 h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = hp[j];
            float2 *p = Z + row * FT_P + 8 * k1;
            cf v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pbfft::to_cf(p[j]);
            pbfft::dft_small<8>(v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = pbfft::cmul(v[2 * j], (cf){h[j].x, h[j].y});
                v[2 * j + 1] = pbfft::cmul(v[2 * j + 1], (cf){h[j].z, h[j].w});
            }
            idft8(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = pbfft::to_f2(v[j]);
        }
    }
    __syncthreads();
    // ---- rows, inverse stage 1 ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float2 *p = Z + (l + 32 * t) * FT_P + n2;
        cf v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = pbfft::to_cf(p[8 * k]); if (k) v[k] = cmul_conj(v[k], tw[k]); }
        idft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[8 * j] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- columns, inverse stage 2 ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float2 *p = Z + (8 * (2 * wave + t)) * FT_P + lane;
        cf v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pbfft::to_cf(p[j * FT_P]);
        idft8(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j * FT_P] = pbfft::to_f2(v[j]);
    }
    __syncthreads();
    // ---- columns, inverse stage 1, into the epilogue: window rows 8 n1 + n2 of columns l, l + 32 ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int x = l + 32 * t;
        const bool colA = x >= R && x < FT_N - R && wxA + x < rg.x_hi;
        const bool colB = colA && hasB && wxB + x < rg.x_hi;
        // (operands first: they arrive while the last butterflies run)
        float xa[8], xb[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int i = 8 * n1 + n2, py = wy0 + i;
            const bool rowok = i >= R && i < FT_N - R && py < rg.y_hi;
            xa[n1] = (rowok && colA) ? load_x1<TX>(a, xpl, py, wxA + x) : 0.f;
            xb[n1] = (rowok && colB) ? load_x1<TX>(a, xpl, py, wxB + x) : 0.f;
        }
        const float2 *p = Z + n2 * FT_P + x;
        cf v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = pbfft::to_cf(p[8 * k * FT_P]); if (k) v[k] = cmul_conj(v[k], tw[k]); }
        idft8(v);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int i = 8 * n1 + n2, py = wy0 + i;
            const bool rowok = i >= R && i < FT_N - R && py < rg.y_hi;
            if (rowok && colA) finish1<TOut>(a, info, opl, py, wxA + x, v[n1].x, xa[n1]);
            if (rowok && colB) finish1<TOut>(a, info, opl, py, wxB + x, v[n1].y, xb[n1]);
        }
    }
}

template <typename TIn, typename TX, typename TOut>
int launch_fft_typed(pb_ctx *ctx, const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    constexpr int Tmin = FT_N - 2 * PB_KRAD;                   // 40: the grid covers the smallest tile; workgroups past an image's own count exit
    const long tiles_x = (ow + Tmin - 1) / Tmin, tiles_y = (oh + Tmin - 1) / Tmin;
    const long jpp = ((tiles_x + 1) / 2) * tiles_y;
    const long jobs = jpp * p.P;
    if (jobs <= 0 || jobs > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "conv pass: bad grid");
    const long grid = (jobs + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_fft_kernel<TIn, TX, TOut>), dim3((unsigned)grid), dim3(FT_NT), kFftLds, ctx->stream, p, (int)jpp, (int)jobs);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

bool pb_conv_fft_supports(const ConvPass &p) {
    return p.in_dtype != PB_U8 && p.x_dtype != PB_U8 && p.out_dtype != PB_U8;
}

// Spectra + per-image body selection for the B images of `info` (B = P / C).
int pb_build_khat(pb_ctx *ctx, const pb_blur_info *info, int B, float2 **khat, pb_fft_sel **sel, bool launch) {
    float2 *k = static_cast<float2 *>(pb_scratch(ctx, "conv.khat", sizeof(float2) * FT_N * FT_N * (size_t)B));
    pb_fft_sel *s = static_cast<pb_fft_sel *>(pb_scratch(ctx, "conv.fftsel", sizeof(pb_fft_sel) * (size_t)B));
    if (!k || !s) return PB_ERR_NOMEM;
    if (launch) {
        ProfScope prof(ctx, PB_PROF_PARAMS);
        hipLaunchKernelGGL(khat_kernel, dim3((unsigned)B), dim3(FT_NT), 0, ctx->stream, info, k, s, ctx->fft_min_phases);
        PB_LAUNCH_CHECK();
    }
    *khat = k; *sel = s;
    return PB_OK;
}

int pb_launch_conv_fft(pb_ctx *ctx, const ConvPass &p) {
    ProfScope prof(ctx, PB_PROF_CONV_FFT);
    const int key = p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype;
    switch (key) {
        case 0: return launch_fft_typed<float, float, float>(ctx, p);
        case 1: return launch_fft_typed<float, float, __half>(ctx, p);
        case 3: return launch_fft_typed<float, __half, float>(ctx, p);
        case 4: return launch_fft_typed<float, __half, __half>(ctx, p);
        case 9: return launch_fft_typed<__half, float, float>(ctx, p);
        case 10: return launch_fft_typed<__half, float, __half>(ctx, p);
        case 12: return launch_fft_typed<__half, __half, float>(ctx, p);
        case 13: return launch_fft_typed<__half, __half, __half>(ctx, p);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "tile-spectrum pass: unsupported dtype combination %d", key);
    }
}
