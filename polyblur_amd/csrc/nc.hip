// Domain-transform NORMALIZED CONVOLUTION (Gastal & Oliveira NC), the variant the reference's author
// recommends over the recursive filter for parallel hardware (RF.cpp:7-11).  Follows NC.cpp:50-204 with
// its single-image semantics applied to every image of a batch (the reference's `find`, NC.cpp:10-47,
// returns the wrong index for batch > 1) and any channel count (the reference hard-codes 3, NC.cpp:128-130).
//
//   dHdx = 1 + sigma_s/sigma_r * sum_c |I[x] - I[x-1]|          ct = cumsum(dHdx)        NC.cpp:157-178
//   per iteration, per row:  l = first j with ct[j] > ct[x] - r,  u = first j with ct[j] > ct[x] + r
//                            F[x] = (SAT[u] - SAT[l]) / (u - l + 1e-4),  SAT = exclusive prefix sums of F
//   then the same along columns (the reference transposes, NC.cpp:181,200-203; so do we).
//
// Rounding follows the reference's CPU execution: torch.cumsum accumulates float32 inputs in double and
// rounds every prefix to float32, so both the domain positions (which decide the box limits by exact float
// comparisons) and the summed-area tables are built from fp64 scans.  One workgroup owns one image row:
// ct and up to three channels' SATs live in LDS, the box limits are found by binary search in LDS and are
// shared by the channels, and every global access is a coalesced row segment.
#include "common.h"

namespace {

constexpr int NT = 256;

unsigned nc_grid(long n, int cap = 8192) {
    long g = (n + NT - 1) / NT;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// dom = 1 + ratio * sum_c |diff|; product and sum rounded separately (float tensor ops in the reference)
template <typename T>
__global__ __launch_bounds__(NT) void nc_domain_kernel(const T *__restrict__ I, float *__restrict__ domx,
                                                       float *__restrict__ domy, int C, int H, int W, float ratio) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const T *src = I + (long)b * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        const int r = (int)(i / W), c = (int)(i - (long)r * W);
        float dx = 0.f, dy = 0.f;
        for (int ch = 0; ch < C; ++ch) {
            const float v = pb_ld(src + ch * HW + i);
            if (c > 0) dx += fabsf(v - pb_ld(src + ch * HW + i - 1));
            if (r > 0) dy += fabsf(v - pb_ld(src + ch * HW + i - W));
        }
        const float px = ratio * dx, py = ratio * dy;
        domx[(long)b * HW + i] = 1.f + px;
        domy[(long)b * HW + i] = 1.f + py;
    }
}

// In-place inclusive prefix sums of G rows of n floats held in LDS (row g at buf + g*stride), accumulated in
// double and rounded to float per element.  Thread t owns the contiguous chunk [t*chunk, (t+1)*chunk); chunk
// is odd so the strided LDS accesses of a wavefront fall in distinct banks.
template <int GMAX>
__device__ __forceinline__ void block_prefix_f64(float *buf, int stride, int G, int n, double *wsum /* [GMAX][NT/64] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = ((n + NT - 1) / NT) | 1;
    const int j0 = min(tid * chunk, n), j1 = min(j0 + chunk, n);
    double local[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        local[g] = 0.0;
        if (g < G)
            for (int j = j0; j < j1; ++j) local[g] += (double)buf[g * stride + j];
    }
    double incl[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        double v = local[g];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(v, o);
            if (lane >= o) v += up;
        }
        incl[g] = v;
        if (lane == 63) wsum[g * (NT / 64) + wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g >= G) continue;
        double base = incl[g] - local[g];                      // exclusive within the wave
        for (int w = 0; w < wave; ++w) base += wsum[g * (NT / 64) + w];
        for (int j = j0; j < j1; ++j) {
            base += (double)buf[g * stride + j];
            buf[g * stride + j] = (float)base;
        }
    }
    __syncthreads();
}

// ct[b,y,:] = cumsum(dom[b,y,:])                                              NC.cpp:177
__global__ __launch_bounds__(NT) void nc_prefix_rows_kernel(const float *__restrict__ dom, float *__restrict__ ct, int W) {
    extern __shared__ float smem[];
    __shared__ double wsum[NT / 64];
    const float *src = dom + (long)blockIdx.x * W;
    float *dst = ct + (long)blockIdx.x * W;
    for (int j = threadIdx.x; j < W; j += NT) smem[j] = src[j];
    __syncthreads();
    block_prefix_f64<1>(smem, 0, 1, W, wsum);
    for (int j = threadIdx.x; j < W; j += NT) dst[j] = smem[j];
}

// ct[b,:,x] = cumsum over rows of dom[b,:,x]; thread = (image, column)         NC.cpp:178
__global__ __launch_bounds__(NT) void nc_prefix_cols_kernel(const float *__restrict__ dom, float *__restrict__ ct, int H, int W,
                                                            long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;
    if (id >= cols_total) return;
    const long b = id / W;
    const int x = (int)(id - b * W);
    const float *s = dom + b * (long)H * W + x;
    float *d = ct + b * (long)H * W + x;
    double acc = 0.0;
    for (int y = 0; y < H; ++y) {
        acc += (double)s[(long)y * W];
        d[(long)y * W] = (float)acc;
    }
}

// (P, H, W) -> (P, W, H), 32x32 tiles through LDS
__global__ __launch_bounds__(NT) void nc_transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W) {
    __shared__ float tile[32][33];
    const long plane = blockIdx.z;
    const float *src = in + plane * (long)H * W;
    float *dst = out + plane * (long)H * W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + ty + 8 * k, x = x0 + tx;
        if (y < H && x < W) tile[ty + 8 * k][tx] = src[(long)y * W + x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + ty + 8 * k, y = y0 + tx;               // output row = input column
        if (x < W && y < H) dst[(long)x * H + y] = tile[tx][ty + 8 * k];
    }
}

// first j in [0, n] with pos[j] > thr (pos ascending; j = n stands for the reference's sentinel, NC.cpp:83-84)
__device__ __forceinline__ int first_greater(const float *pos, int n, float thr) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pos[mid] > thr) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// One horizontal box pass over row blockIdx.x of image b = row / H, channels [c0, c0+G).  NC.cpp:50-139
constexpr int NC_GMAX = 3;
__global__ __launch_bounds__(NT) void nc_box_rows_kernel(const float *__restrict__ F, const float *__restrict__ ct,
                                                         float *__restrict__ out, int C, int H, int W, float radius,
                                                         int c0, int G) {
    extern __shared__ float smem[];
    __shared__ double wsum[NC_GMAX * (NT / 64)];
    const int stride = W + 1;
    float *pos = smem;                       // [W]
    float *sat = smem + stride;              // [G][W+1]
    const long row = blockIdx.x;             // over B*H
    const long b = row / H;
    const int y = (int)(row - b * H);
    const float *crow = ct + row * W;
    for (int j = threadIdx.x; j < W; j += NT) pos[j] = crow[j];
    for (int g = 0; g < G; ++g) {
        const float *frow = F + ((b * C + c0 + g) * H + y) * (long)W;
        for (int j = threadIdx.x; j < W; j += NT) sat[g * stride + 1 + j] = frow[j];
        if (threadIdx.x == 0) sat[g * stride] = 0.f;
    }
    __syncthreads();
    block_prefix_f64<NC_GMAX>(sat + 1, stride, G, W, wsum);                   // SAT[j+1] = sum_{k<=j} F[k]   :113-114
    for (int j = threadIdx.x; j < W; j += NT) {
        const float c = pos[j];
        const int li = first_greater(pos, W, c - radius);                       // :65-66, 95-108
        const int ui = first_greater(pos, W, c + radius);
        const float den = (float)(ui - li) + 0.0001f;                           // :134
        for (int g = 0; g < G; ++g)
            out[((b * C + c0 + g) * H + y) * (long)W + j] = (sat[g * stride + ui] - sat[g * stride + li]) / den;
    }
}

template <typename T> __global__ void nc_to_float_kernel(const T *__restrict__ in, float *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = pb_ld(in + i);
}

int box_pass(pb_ctx *ctx, const float *F, const float *ct, float *out, int B, int C, int H, int W, float radius) {
    // channels per workgroup: as many as fit 64 KB of LDS next to the row of domain positions
    int G = (int)(65536 / (sizeof(float) * (W + 1))) - 1;
    if (G > NC_GMAX) G = NC_GMAX;
    if (G > C) G = C;
    if (G < 1) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "normalized convolution: rows longer than 8190 samples are not supported");
    for (int c0 = 0; c0 < C; c0 += G) {
        const int g = (C - c0 < G) ? C - c0 : G;
        const size_t lds = sizeof(float) * (size_t)(W + 1) * (1 + g);
        hipLaunchKernelGGL(nc_box_rows_kernel, dim3((unsigned)((long)B * H)), dim3(NT), lds, ctx->stream, F, ct, out, C, H, W,
                           radius, c0, g);
    }
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

int pb_nc_filter_impl(pb_ctx *ctx, const void *in, int dtype, float *out, int B, int C, int H, int W, float sigma_s,
                      float sigma_r, int num_iterations) {
    const long HW = (long)H * W, n = (long)B * C * HW;
    if (H > 8190 || W > 8190) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "normalized convolution: H and W must be <= 8190");
    ProfScope prof(ctx, PB_PROF_PREFILTER);
    float *dom = static_cast<float *>(pb_scratch(ctx, "nc.dom", sizeof(float) * 2 * B * HW));     // domx | domy, later ctV^T in domx's half
    float *ctH = static_cast<float *>(pb_scratch(ctx, "nc.ctH", sizeof(float) * B * HW));
    float *ctV = static_cast<float *>(pb_scratch(ctx, "nc.ctV", sizeof(float) * B * HW));
    float *tmp = static_cast<float *>(pb_scratch(ctx, "nc.tmp", sizeof(float) * n));
    if (!dom || !ctH || !ctV || !tmp) return PB_ERR_NOMEM;
    float *domx = dom, *domy = dom + (long)B * HW;
    const float ratio = sigma_s / sigma_r;                                      // float / float, NC.cpp:173
    dim3 dgrid(nc_grid(HW, 2048), B);
    if (dtype == PB_F32) {
        hipLaunchKernelGGL(nc_domain_kernel<float>, dgrid, dim3(NT), 0, ctx->stream, static_cast<const float *>(in), domx, domy, C, H, W, ratio);
        hipLaunchKernelGGL(nc_to_float_kernel<float>, dim3(nc_grid(n)), dim3(NT), 0, ctx->stream, static_cast<const float *>(in), out, n);
    } else {
        hipLaunchKernelGGL(nc_domain_kernel<__half>, dgrid, dim3(NT), 0, ctx->stream, static_cast<const __half *>(in), domx, domy, C, H, W, ratio);
        hipLaunchKernelGGL(nc_to_float_kernel<__half>, dim3(nc_grid(n)), dim3(NT), 0, ctx->stream, static_cast<const __half *>(in), out, n);
    }
    hipLaunchKernelGGL(nc_prefix_rows_kernel, dim3((unsigned)((long)B * H)), dim3(NT), sizeof(float) * W, ctx->stream, domx, ctH, W);
    const long cols_total = (long)B * W;
    hipLaunchKernelGGL(nc_prefix_cols_kernel, dim3((unsigned)((cols_total + NT - 1) / NT)), dim3(NT), 0, ctx->stream, domy, ctV, H, W, cols_total);
    float *ctVt = domx;                                                         // domx is dead once ctH exists
    dim3 tgrid((W + 31) / 32, (H + 31) / 32, B), tgrid_t((H + 31) / 32, (W + 31) / 32, B);
    hipLaunchKernelGGL(nc_transpose_kernel, tgrid, dim3(NT), 0, ctx->stream, ctV, ctVt, H, W);
    PB_LAUNCH_CHECK();
    const int N = num_iterations;
    dim3 pgrid((W + 31) / 32, (H + 31) / 32, B * C), pgrid_t((H + 31) / 32, (W + 31) / 32, B * C);
    for (int i = 0; i < N; ++i) {
        // NC.cpp:194-197: float sigma_H_i = <double expression>; float box_radius = sqrt(3) * sigma_H_i
        const float sigma_i = (float)((double)sigma_s * std::sqrt(3.0) * std::pow(2.0, N - (i + 1)) / std::sqrt(std::pow(4.0, N) - 1.0));
        const float radius = (float)(std::sqrt(3.0) * (double)sigma_i);
        int rc = box_pass(ctx, out, ctH, tmp, B, C, H, W, radius);               // rows
        if (rc) return rc;
        hipLaunchKernelGGL(nc_transpose_kernel, pgrid, dim3(NT), 0, ctx->stream, tmp, out, H, W);
        rc = box_pass(ctx, out, ctVt, tmp, B, C, W, H, radius);                  // columns, as rows of the transpose
        if (rc) return rc;
        hipLaunchKernelGGL(nc_transpose_kernel, pgrid_t, dim3(NT), 0, ctx->stream, tmp, out, W, H);
        PB_LAUNCH_CHECK();
    }
    return PB_OK;
}
