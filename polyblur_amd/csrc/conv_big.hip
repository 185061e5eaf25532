// Reblurring pass for kernels LARGER than the 25 x 25 record: ker_size 26 .. 49.
//
// The reference builds its Gaussian on a ker_size x ker_size grid of any size (blur_estimation.py:211-232; the grid of an
// even size is off-centre, :222) and pads the image by ker_size / 2 (utils.py:48-53).  The engine's records, LDS tiles and
// window halos are laid out for the default 25 taps; sizes above that are rare and get a body of their own instead of
// growing every other one: the taps live in the context's scratch (big_taps_kernel: same formula, same normalisation, one
// 49 x 49 array per image, zero outside the support), and one Horner step  out = scale (K * in) + coef x  is a plain
// LDS-tiled stencil -- a 32 x 64 output tile per workgroup, the thread's taps of a kernel row arrive as scalar loads, its
// samples as 16-byte LDS reads into a sliding register window.  Same boundary models, operands and epilogue as the other
// bodies (filters.py:14-49, deblurring.py:122-138); the edgetaper's blends (edgetaper.py:26-33) with the weights of THIS kernel --
// autocorrelations of its projections at up to 49 lags, formed beside the taps -- since round 6.  Up to 2401 multiply-adds
// per sample: 12 (ker_size 27) to 40 ms (49) per 4K call of three iterations = 14.8 T multiply-adds per second.  The
// compiler packs the inner loop into v_pk_fma_f32 (64 per 16 taps) and the pass is then bound by its LDS reads -- 192 bytes
// per lane for those 64 instructions, every window pair being read twice, aligned and offset by one sample, 1.5 x the
// cycles of the arithmetic -- and the scalar tap loads in front of them: API completeness, not a tuned path.
#include "conv_common.h"

namespace {

constexpr int BK_R = 24;                 // largest half-size
constexpr int BK_P = 56;                 // taps per stored row: 49, then zeros (a row is read in groups of four, up to 2 R + 4 taps)
constexpr int BK_ROWS = 2 * BK_R + 1;
constexpr int BK_AC = 64;                // floats per stored autocorrelation (49 lags)
constexpr int BG_TW = 64, BG_TH = 32, BG_NT = 256;

__device__ float big_block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < BG_NT / 64; ++w) s += red[w];
    return s;
}

// taps[img][iy + 24][ix + 24] multiplies the sample (iy, ix) away from the output.  The formula, the off-centre grid of even
// sizes and the roll of the 'fft' method (shift) are estimate.hip's finish_record's, on a larger grid.
__global__ __launch_bounds__(BG_NT) void big_taps_kernel(const pb_blur_info *infos, float *taps, float *acorr, int ksize, int shift) {
    __shared__ float red[BG_NT / 64];
    const pb_blur_info *info = infos + blockIdx.x;
    float *out = taps + (long)blockIdx.x * (BK_ROWS * BK_P);
    const float th = -info->theta, sg = info->sigma, rh = info->rho;
    const float c = cosf(th), s = sinf(th);
    const float i1 = 1.f / (sg * sg), i2 = 1.f / (rh * rh);
    const float a00 = c * c * i1 + s * s * i2;
    const float a01 = s * c * (i1 - i2);
    const float a11 = c * c * i2 + s * s * i1;
    const int lo = (ksize & 1) ? -(ksize / 2) : -(ksize / 2) + 1, hi = ksize / 2;
    float part = 0.f;
    for (int idx = threadIdx.x; idx < BK_ROWS * BK_P; idx += BG_NT) {
        const int iy = idx / BK_P - BK_R, ix = idx % BK_P - BK_R;
        float e = 0.f;
        if (iy >= lo && iy <= hi && ix >= lo && ix <= hi) {
            const float Y = (float)(iy - shift), X = (float)(ix - shift);
            e = expf(-0.5f * ((X * a00 + Y * a01) * X + (X * a01 + Y * a11) * Y));
        }
        out[idx] = e;
        part += e;
    }
    const float total = big_block_sum(part, red);
    for (int idx = threadIdx.x; idx < BK_ROWS * BK_P; idx += BG_NT) out[idx] = out[idx] / total;     // (each thread its own entries)
    // The edgetaper's weights for this kernel (edgetaper.py:10-23): autocorrelations of its two projections at lags 0 .. 48 --
    // what estimate.hip's finish_record keeps for the 25 x 25 record, on the larger grid (an autocorrelation does not care where
    // the taps sit: even sizes as they are).  acorr[img][0][l]: of ky (rows), [1][l]: of kx (columns).
    __shared__ float proj[2][BK_ROWS];
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 2 * BK_ROWS) {
        const int ax = threadIdx.x / BK_ROWS, t = threadIdx.x - ax * BK_ROWS;
        float a = 0.f;
        for (int i = 0; i < BK_ROWS; ++i) a += ax ? out[i * BK_P + t] : out[t * BK_P + i];       // kx[t] = column sum, ky[t] = row sum
        proj[ax][t] = a;
    }
    __syncthreads();
    if (threadIdx.x < 2 * BK_ROWS) {
        const int ax = threadIdx.x / BK_ROWS, l = threadIdx.x - ax * BK_ROWS;
        float a = 0.f;
        for (int n = 0; n + l < BK_ROWS; ++n) a += proj[ax][n] * proj[ax][n + l];
        acorr[((long)blockIdx.x * 2 + (ax ? 1 : 0)) * BK_AC + l] = a;
    }
}

// taper_weight (conv_common.h) for autocorrelations of up to BK_ROWS lags
__device__ __forceinline__ float big_taper_weight(const float *ac, int p, int n) {
    const int q = n - 1 - p;
    const float z = ((p < BK_ROWS) ? ac[p] : 0.f) + ((q < BK_ROWS) ? ac[q] : 0.f);
    return 1.f - z * __frcp_rn(ac[0]);
}

template <typename T>
__device__ __forceinline__ float big_load(const ConvPass &a, const T *plane, int py, int px) {
    const int iy = map_axis(py, a.H, a.in_kind, a.boundary, a.pad), ix = map_axis(px, a.W, a.in_kind, a.boundary, a.pad);
    if ((iy | ix) < 0) return 0.f;
    return pb_ld(plane + (long)iy * a.in_pitch + ix);
}

template <typename TIn, typename TX, typename TOut>
__global__ __launch_bounds__(BG_NT) void conv_big_kernel(const ConvPass a, const float *__restrict__ taps, const float *__restrict__ acorr,
                                                        int R, int tiles_x, int tiles_per_plane) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int plane = blockIdx.x / tiles_per_plane, local = blockIdx.x - plane * tiles_per_plane;
    const int ty = local / tiles_x, tx = local - ty * tiles_x;
    const int img = plane / a.C;
    const OutRegion rg = out_region(a);
    const int oy0 = rg.y_lo + ty * BG_TH, ox0 = rg.x_lo + tx * BG_TW;
    const int nv4 = (2 * R + 4) / 4;                       // groups of four taps per kernel row (the last one zero-padded)
    const int rows = BG_TH + 2 * R, pitch = BG_TW + 4 * nv4 + 4;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    for (int r = threadIdx.x >> 6; r < rows; r += BG_NT / 64)
        for (int cidx = threadIdx.x & 63; cidx < pitch; cidx += 64)
            tile[r * pitch + cidx] = big_load(a, ipl, oy0 - R + r, ox0 - R + cidx);
    __syncthreads();
    const int gx = threadIdx.x & 15, gy = threadIdx.x >> 4;       // outputs: rows gy and gy + 16, columns 4 gx .. 4 gx + 3
    const PB_CONSTANT float *tp = as_constant(taps + (long)img * (BK_ROWS * BK_P) + (BK_R - R) * BK_P + (BK_R - R));
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u <= 2 * R; ++u) {
        const float *r0 = tile + (gy + u) * pitch + 4 * gx, *r1 = r0 + 16 * pitch;
        const PB_CONSTANT float *trow = tp + u * BK_P;
        float4 c0 = *reinterpret_cast<const float4 *>(r0), c1 = *reinterpret_cast<const float4 *>(r1);
        for (int v4 = 0; v4 < nv4; ++v4) {
            const float4 n0 = *reinterpret_cast<const float4 *>(r0 + 4 * v4 + 4), n1 = *reinterpret_cast<const float4 *>(r1 + 4 * v4 + 4);
            const float w0[8] = {c0.x, c0.y, c0.z, c0.w, n0.x, n0.y, n0.z, n0.w};
            const float w1[8] = {c1.x, c1.y, c1.z, c1.w, n1.x, n1.y, n1.z, n1.w};
            const float t[4] = {trow[4 * v4], trow[4 * v4 + 1], trow[4 * v4 + 2], trow[4 * v4 + 3]};     // uniform: scalar loads
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0[i] = fmaf(t[j], w0[i + j], acc0[i]);
                    acc1[i] = fmaf(t[j], w1[i + j], acc1[i]);
                }
            c0 = n0; c1 = n1;
        }
    }
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const pb_blur_info *info = a.info + img;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int py = oy0 + gy + 16 * h, px = ox0 + 4 * gx;
        if (py >= rg.y_hi || px >= rg.x_hi) continue;
        const float *acc = h ? acc1 : acc0;
        const float4 xq = load_x4<TX>(a, xpl, py, px);
        if (a.epilogue == EPI_TAPER) {
            // out = alpha x + (1 - alpha) (K * in), alpha = v1[py] v2[px] from THIS kernel's autocorrelations (edgetaper.py:26-33);
            // finish4 then stores it as it is (scale 1, no x term)
            const float *ac = acorr + (long)img * 2 * BK_AC;
            const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
            const float ty = big_taper_weight(ac, py, Hp);
            const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float al = ty * big_taper_weight(ac + BK_AC, min(px + i, Wp - 1), Wp);
                v[i] = al * xv[i] + (1.f - al) * acc[i];
            }
            ConvPass b = a;
            b.epilogue = EPI_HORNER; b.scale = 1.f; b.coef = 0.f;
            finish4<TOut>(b, info, opl, rg, py, px, make_float4(v[0], v[1], v[2], v[3]), make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        finish4<TOut>(a, info, opl, rg, py, px, make_float4(acc[0], acc[1], acc[2], acc[3]), xq);
    }
}

// (Loops compiled for fixed half-size classes, the whole tap row in scalar registers at once: measured no faster at the
// class sizes -- 38.5 vs 39.6 ms per 4K call at 49 -- and slower in between: 27.9 vs 20.1 ms at 35.)
template <typename TIn, typename TX, typename TOut>
int launch_big_typed(pb_ctx *ctx, const ConvPass &p, const float *taps, const float *acorr, int R) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    const int ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    const long tiles_x = (ow + BG_TW - 1) / BG_TW, tiles_y = (oh + BG_TH - 1) / BG_TH;
    const long tpp = tiles_x * tiles_y, blocks = tpp * p.P;
    if (blocks <= 0 || blocks > 0x7fffffffL) return pb_fail(ctx, PB_ERR_BADARG, "large-kernel pass: bad grid");
    const int nv4 = (2 * R + 4) / 4;
    const size_t lds = sizeof(float) * (size_t)(BG_TH + 2 * R) * (BG_TW + 4 * nv4 + 4);
    hipLaunchKernelGGL((conv_big_kernel<TIn, TX, TOut>), dim3((unsigned)blocks), dim3(BG_NT), lds, ctx->stream, p, taps, acorr, R,
                       (int)tiles_x, (int)tpp);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

// The taps of B estimated kernels on the ker_size x ker_size grid (context scratch; valid until the next call)
int pb_build_big_taps(pb_ctx *ctx, const pb_blur_info *dev_info, int B, int ksize, int shift, const float **taps) {
    // (the taps, then the two autocorrelations of every image: one scratch buffer, the pointer pb_launch_conv_big gets back)
    const size_t ntaps = (size_t)BK_ROWS * BK_P * (size_t)B;
    float *t = static_cast<float *>(pb_scratch(ctx, "big.taps", sizeof(float) * (ntaps + 2 * BK_AC * (size_t)B)));
    if (!t) return PB_ERR_NOMEM;
    ProfScope prof(ctx, PB_PROF_PARAMS);
    hipLaunchKernelGGL(big_taps_kernel, dim3((unsigned)B), dim3(BG_NT), 0, ctx->stream, dev_info, t, t + ntaps, ksize, shift);
    PB_LAUNCH_CHECK();
    *taps = t;
    return PB_OK;
}

int pb_launch_conv_big(pb_ctx *ctx, const ConvPass &p, const float *taps, int ksize) {
    ProfScope prof(ctx, PB_PROF_CONV);
    const int R = ksize / 2;
    const float *acorr = taps + (size_t)BK_ROWS * BK_P * (size_t)(p.P / p.C);      // (behind the taps: pb_build_big_taps)
    typedef unsigned char u8;
    switch (p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype) {
        case 0: return launch_big_typed<float, float, float>(ctx, p, taps, acorr, R);
        case 1: return launch_big_typed<float, float, __half>(ctx, p, taps, acorr, R);
        case 3: return launch_big_typed<float, __half, float>(ctx, p, taps, acorr, R);
        case 4: return launch_big_typed<float, __half, __half>(ctx, p, taps, acorr, R);
        case 9: return launch_big_typed<__half, float, float>(ctx, p, taps, acorr, R);
        case 10: return launch_big_typed<__half, float, __half>(ctx, p, taps, acorr, R);
        case 12: return launch_big_typed<__half, __half, float>(ctx, p, taps, acorr, R);
        case 13: return launch_big_typed<__half, __half, __half>(ctx, p, taps, acorr, R);
        case 24: return launch_big_typed<u8, u8, float>(ctx, p, taps, acorr, R);
        case 6: return launch_big_typed<float, u8, float>(ctx, p, taps, acorr, R);
        case 8: return launch_big_typed<float, u8, u8>(ctx, p, taps, acorr, R);
        case 2: return launch_big_typed<float, float, u8>(ctx, p, taps, acorr, R);
        default: return pb_fail(ctx, PB_ERR_UNSUPPORTED, "large-kernel pass: unsupported dtype combination");
    }
}
