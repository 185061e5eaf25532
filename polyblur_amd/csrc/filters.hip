// Optional stages of the hot path:
//   halo masking                     deblurring.py:172-208   (bug-compatible, see halo_kernel)
//   domain-transform recursive filter domain_transform.py:6-85, native twin RF.cpp:14-92
//   5x5 bilateral filter             filters.py:107-148
//   prefilter recombination          deblurring.py:84,88
#include "common.h"

int pb_fourier_gradients_impl(pb_ctx *ctx, const float *planes, int P, int H, int W, float *gx, float *gy);

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------
// halo masking
// ------------------------------------------------------------------------------------
// nM[plane] = sum_{H,W} gx^2 + gy^2          (deblurring.py:178-179,206)
// VEC: planes of whole groups of four samples on 16-byte (fp16: 8-byte) boundaries -- four samples per lane and trip (round 6:
// 78 -> 45 us on a 4K image's three planes); which samples a lane sums, and in which order, depends on the plane's size alone.
template <typename T> __device__ __forceinline__ float4 ld4v(const T *p);
template <typename TG, bool VEC>
__global__ __launch_bounds__(NT) void grad_energy_kernel(const TG *__restrict__ gx, const TG *__restrict__ gy,
                                                         float *__restrict__ partial, long HW, int blocks_per_plane) {
    const int plane = blockIdx.x / blocks_per_plane;
    const int blk = blockIdx.x - plane * blocks_per_plane;
    const TG *a = gx + (long)plane * HW, *b = gy + (long)plane * HW;
    float s = 0.f;
    if constexpr (VEC) {
        for (long i = 4 * ((long)blk * NT + threadIdx.x); i < HW; i += 4l * blocks_per_plane * NT) {
            const float4 u = ld4v(a + i), v = ld4v(b + i);
            s += u.x * u.x + v.x * v.x;
            s += u.y * u.y + v.y * v.y;
            s += u.z * u.z + v.z * v.z;
            s += u.w * u.w + v.w * v.w;
        }
    } else {
        for (long i = (long)blk * NT + threadIdx.x; i < HW; i += (long)blocks_per_plane * NT) {
            const float u = pb_ld(a + i), v = pb_ld(b + i);
            s += u * u + v * v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float red[NT / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// one wave per plane folds that plane's per-block partial sums in a fixed order: nM is bit-reproducible
__global__ __launch_bounds__(64) void grad_energy_fold_kernel(const float *__restrict__ partial, float *__restrict__ nM,
                                                              int blocks_per_plane) {
    const float *p = partial + (long)blockIdx.x * blocks_per_plane;
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks_per_plane; i += 64) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) nM[blockIdx.x] = s;
}

// M = -gx*ox - gy*gy  (sic: deblurring.py:174 multiplies grad_y by itself, not by gout_y)
// z = max(M / (nM + M), 0);  out = y + z (x - y)   [+ clamp to [0,1], deblurring.py:239]
// With a prefilter the masked result is recombined on the spot (cur != nullptr; recombine_kernel's arithmetic):
// out = clip(clip(out, 0, 1) + (cur - smooth), 0, 1) -- one pass over the batch instead of a store, a load and a pass.
// TG: the type the three gradient planes are kept in -- fp32, or fp16 where the call's images are fp16 (z = M / (nM + M) is a
// ratio of one sample's products to the whole plane's energy, ~1e-5 on an image: an fp16 rounding of its factors moves the
// output by ~1e-9, and the three planes are 12 of the 28 bytes this kernel moves per sample)
// four consecutive samples of a plane as floats (16 bytes of fp32, 8 of fp16): the streaming kernels' unit where rows allow it
template <> __device__ __forceinline__ float4 ld4v<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4v<__half>(const __half *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&u.x)), b = __half22float2(*reinterpret_cast<const __half2 *>(&u.y));
    return make_float4(a.x, a.y, b.x, b.y);
}
template <typename T> __device__ __forceinline__ void st4v(T *p, float4 v);
template <> __device__ __forceinline__ void st4v<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void st4v<__half>(__half *p, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned *>(&a); u.y = *reinterpret_cast<const unsigned *>(&b);
    *reinterpret_cast<uint2 *>(p) = u;
}

// VEC: every plane's rows start on a boundary of four samples (W, the x operand's pitch and plane stride multiples of 4): four
// samples per thread and trip -- the same operations per sample, 16-byte (fp16: 8-byte) accesses, no 64-bit division per sample
template <typename TX, typename TOut, typename TG, bool VEC>
__global__ __launch_bounds__(NT) void halo_kernel(const TX *__restrict__ x, int x_pitch, long x_plane,
                                                  const float *__restrict__ y, const TG *__restrict__ gx,
                                                  const TG *__restrict__ gy, const TG *__restrict__ ox,
                                                  const float *__restrict__ nM, TOut *__restrict__ out, int P, int H, int W,
                                                  int clamp01, const void *__restrict__ cur, int cur_is_half,
                                                  const float *smooth) {
    const long HW = (long)H * W;
    if constexpr (VEC) {
        const int W4 = W >> 2;
        const long n4 = HW >> 2;
        for (int plane = blockIdx.y; plane < P; plane += gridDim.y) {
            const float nm = nM[plane];
            for (long i4 = (long)blockIdx.x * NT + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * NT) {
                const int r = (int)((unsigned)i4 / (unsigned)W4), c = 4 * ((int)i4 - r * W4);       // (H W / 4 < 2^32: sides up to 65536)
                const long k = (long)plane * HW + 4 * i4;
                const float4 gxx = ld4v(gx + k), gyy = ld4v(gy + k), oxx = ld4v(ox + k), yv = ld4v(y + k);
                const float4 xv = ld4v(x + (long)plane * x_plane + (long)r * x_pitch + c);
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f), sm = cv;
                if (cur) {
                    cv = cur_is_half ? ld4v(static_cast<const __half *>(cur) + k) : ld4v(static_cast<const float *>(cur) + k);
                    sm = ld4v(smooth + k);
                }
                const float g1[4] = {gxx.x, gxx.y, gxx.z, gxx.w}, g2[4] = {gyy.x, gyy.y, gyy.z, gyy.w}, o1[4] = {oxx.x, oxx.y, oxx.z, oxx.w};
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
                const float cs[4] = {cv.x, cv.y, cv.z, cv.w}, ss[4] = {sm.x, sm.y, sm.z, sm.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float M = fmaf(-g1[e], o1[e], -__fmul_rn(g2[e], g2[e]));      // (roundings written out: both forms of the kernel give the same bits)
                    const float z = fmaxf(M / (nm + M), 0.f);
                    float v = fmaf(z, xs[e] - ys[e], ys[e]);
                    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                    if (cur) {
                        const float d = cs[e] - ss[e];
                        v = fminf(fmaxf(v, 0.f), 1.f) + d;
                        v = fminf(fmaxf(v, 0.f), 1.f);
                    }
                    o[e] = v;
                }
                st4v(out + k, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
        return;
    }
    for (int plane = blockIdx.y; plane < P; plane += gridDim.y) {      // (grid.y is capped at 65535)
    const float nm = nM[plane];
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        const int r = (int)(i / W), c = (int)(i - (long)r * W);
        const long k = (long)plane * HW + i;
        const float gxx = pb_ld(gx + k), gyy = pb_ld(gy + k);
        const float M = fmaf(-gxx, pb_ld(ox + k), -__fmul_rn(gyy, gyy));
        const float z = fmaxf(M / (nm + M), 0.f);
        const float xv = pb_ld(x + (long)plane * x_plane + (long)r * x_pitch + c);
        const float yv = y[k];
        float v = fmaf(z, xv - yv, yv);
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        if (cur) {
            const float cv = cur_is_half ? pb_ld(static_cast<const __half *>(cur) + k) : static_cast<const float *>(cur)[k];
            const float d = cv - smooth[k];
            v = fminf(fmaxf(v, 0.f), 1.f) + d;
            v = fminf(fmaxf(v, 0.f), 1.f);
        }
        pb_st(out + k, v);
    }
    }
}

// out = clip(clip(y,0,1) + (cur - smooth), 0, 1)          (deblurring.py:84,88,239)
template <typename TC, typename TO>
__global__ __launch_bounds__(NT) void recombine_kernel(const float *__restrict__ y, const TC *__restrict__ cur,
                                                       const float *__restrict__ smooth, TO *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float d = pb_ld(cur + i) - smooth[i];
        const float v = fminf(fmaxf(y[i], 0.f), 1.f) + d;
        pb_st(out + i, fminf(fmaxf(v, 0.f), 1.f));
    }
}

// ------------------------------------------------------------------------------------
// bilateral 5x5
// ------------------------------------------------------------------------------------
// Round 6: the weight of a tap is ONE exponential, exp2(d^2 kc + r^2 ks) with kc = -log2(e) / (2 sigma_color^2) and ks likewise
// for the spatial term (filters.py:111,134-136: exp(-d^2 / var2) * gw -- the same number; the library expf spent ~14
// instructions per tap on an accuracy a weight does not need: v_exp_f32 is within 1 ulp, and an error of the exponent's
// rounding, ~|x| 2^-24, weighs on taps whose weight is ~e^x), and a thread forms four vertically adjacent outputs from an
// 8 x 5 window held in registers: 10 LDS reads per output instead of 25.  370 -> ~150 us per 4K image.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(NT) void bilateral5_kernel(const TIn *__restrict__ in, TOut *__restrict__ out, int P, int H, int W,
                                                        float kc, float ks1) {
    constexpr int TWB = 64, THB = 16, R = 2, LW = TWB + 2 * R, LH = THB + 2 * R, PER = THB / (NT / TWB);
    __shared__ float s[LH * LW];
    const int x0 = blockIdx.x * TWB, y0 = blockIdx.y * THB;
    float ks[9];                                                     // by squared distance 0 .. 8
#pragma unroll
    for (int d = 0; d < 9; ++d) ks[d] = (float)d * ks1;
    for (int plane = blockIdx.z; plane < P; plane += gridDim.z) {      // (grid.z is capped at 65535)
    if (plane != (int)blockIdx.z) __syncthreads();
    const TIn *src = in + (long)plane * H * W;
    for (int e = threadIdx.x; e < LH * LW; e += NT) {
        const int r = e / LW, c = e - r * LW;
        const int yy = min(max(y0 + r - R, 0), H - 1), xx = min(max(x0 + c - R, 0), W - 1);   // replicate pad
        s[e] = pb_ld(src + (long)yy * W + xx);
    }
    __syncthreads();
    const int tx = threadIdx.x % TWB, ty = (threadIdx.x / TWB) * PER;     // outputs: rows ty .. ty + PER - 1 of the tile, column tx
    float win[PER + 2 * R][5];
#pragma unroll
    for (int r = 0; r < PER + 2 * R; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) win[r][j] = s[(ty + r) * LW + tx + j];
#pragma unroll
    for (int o = 0; o < PER; ++o) {
        const int yy = y0 + ty + o, xx = x0 + tx;
        if (yy >= H || xx >= W) continue;
        const float ctr = win[o + R][R];
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float v = win[o + i][j];
                const float d = v - ctr;
                const float w = __builtin_amdgcn_exp2f(fmaf(d * d, kc, ks[(i - 2) * (i - 2) + (j - 2) * (j - 2)]));
                num = fmaf(w, v, num);
                den += w;
            }
        pb_st(out + (long)plane * H * W + (long)yy * W + xx, num / (den + 1e-5f));
    }
    }
}

// ------------------------------------------------------------------------------------
// domain-transform recursive filter
// ------------------------------------------------------------------------------------
// The recurrence F[i] <- F[i] + V[i] (F[i-1] - F[i]) is the affine map F[i] = V[i] F[i-1] + (1-V[i]) x[i].
// Rows: one wave per (image, row); lanes own 64 consecutive columns and combine their maps with a
// wave-level inclusive scan (composition of affine maps), carrying the last value between chunks, so
// every global access is a coalesced 256-B row segment.  Columns: one thread per column walks down
// and up (adjacent lanes = adjacent columns, already coalesced).
//
// dom[i] = 1 + sigma_s/sigma_r * sum_c |J[c][i] - J[c][i-1]| (0 difference at i = 0);  V = a^dom.

template <typename T>
__global__ __launch_bounds__(NT) void dt_domain_kernel(const T *__restrict__ J, float *__restrict__ domx,
                                                       float *__restrict__ domy, int C, int H, int W, float ratio) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const T *src = J + (long)b * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        const int r = (int)(i / W), c = (int)(i - (long)r * W);
        float dx = 0.f, dy = 0.f;
        for (int ch = 0; ch < C; ++ch) {
            const float v = pb_ld(src + ch * HW + i);
            if (c > 0) dx += fabsf(v - pb_ld(src + ch * HW + i - 1));
            if (r > 0) dy += fabsf(v - pb_ld(src + ch * HW + i - W));
        }
        domx[(long)b * HW + i] = 1.f + ratio * dx;
        domy[(long)b * HW + i] = 1.f + ratio * dy;
    }
}

struct Affine { float a, b; };   // f(t) = a t + b
__device__ __forceinline__ Affine compose(Affine second, Affine first) {   // second(first(t))
    return Affine{second.a * first.a, fmaf(second.a, first.b, second.b)};
}

// One horizontal pass (left->right then right->left) over every row of every plane, in place on F.
__global__ __launch_bounds__(NT) void dt_rows_kernel(float *__restrict__ F, const float *__restrict__ domx, int C, int H,
                                                     int W, float log_a, long rows_total) {
    const int lane = threadIdx.x & 63;
    const long row_id = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);   // over B*C*H
    if (row_id >= rows_total) return;
    const long plane = row_id / H;
    const int r = (int)(row_id - plane * H);
    const long b = plane / C;
    float *f = F + row_id * W;
    const float *d = domx + (b * H + r) * (long)W;
    // ---- left -> right:  F[i] = V[i] F[i-1] + (1 - V[i]) F[i],  i >= 1
    float carry = 0.f;
    for (int base = 0; base < W; base += 64) {
        const int i = base + lane;
        float x = 0.f, v = 1.f;
        if (i < W) { x = f[i]; v = expf(d[i] * log_a); }
        Affine m = (i == 0 || i >= W) ? Affine{0.f, x} : Affine{v, (1.f - v) * x};
        if (i >= W) m = Affine{1.f, 0.f};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Affine prev{__shfl_up(m.a, o), __shfl_up(m.b, o)};
            if (lane >= o) m = compose(m, prev);
        }
        const float y = fmaf(m.a, carry, m.b);
        if (i < W) f[i] = y;
        carry = __shfl(y, 63);
        if (base + 63 >= W) carry = __shfl(y, (W - 1) - base);
    }
    // ---- right -> left:  F[i] = V[i+1] F[i+1] + (1 - V[i+1]) F[i],  i <= W-2
    const int last_base = ((W - 1) / 64) * 64;
    carry = 0.f;
    for (int base = last_base; base >= 0; base -= 64) {
        const int i = base + (63 - lane);            // lane 0 handles the right-most element of the chunk
        float x = 0.f, v = 1.f;
        if (i < W) x = f[i];
        if (i + 1 < W) v = expf(d[i + 1] * log_a);
        Affine m = (i >= W - 1) ? Affine{0.f, x} : Affine{v, (1.f - v) * x};
        if (i >= W) m = Affine{1.f, 0.f};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            Affine prev{__shfl_up(m.a, o), __shfl_up(m.b, o)};
            if (lane >= o) m = compose(m, prev);
        }
        const float y = fmaf(m.a, carry, m.b);
        if (i < W) f[i] = y;
        carry = __shfl(y, 63);                        // element `base`, the left-most of this chunk
    }
}

// One vertical pass (top->bottom then bottom->top) in place on F; thread = (plane, column).  The recurrence is
// sequential down a column, but its operands are not: the samples and domain weights of the next DT_U rows are
// fetched together before the U dependent steps run, so a column waits for memory once per U rows, not every row.
constexpr int DT_U = 16;
__global__ __launch_bounds__(NT) void dt_cols_kernel(float *__restrict__ F, const float *__restrict__ domy, int C, int H,
                                                     int W, float log_a, long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;   // over B*C*W
    if (id >= cols_total) return;
    const long plane = id / W;
    const int c = (int)(id - plane * W);
    const long b = plane / C;
    float *f = F + plane * (long)H * W + c;
    const float *d = domy + b * (long)H * W + c;
    float prev = f[0];
    for (int r0 = 1; r0 < H; r0 += DT_U) {
        float xs[DT_U], vs[DT_U];
#pragma unroll
        for (int u = 0; u < DT_U; ++u) {
            const int r = min(r0 + u, H - 1);
            xs[u] = f[(long)r * W];
            vs[u] = d[(long)r * W];
        }
#pragma unroll
        for (int u = 0; u < DT_U; ++u) {
            if (r0 + u < H) {
                const float v = expf(vs[u] * log_a);
                prev = xs[u] + v * (prev - xs[u]);
                f[(long)(r0 + u) * W] = prev;
            }
        }
    }
    for (int r0 = H - 2; r0 >= 0; r0 -= DT_U) {
        float xs[DT_U], vs[DT_U];
#pragma unroll
        for (int u = 0; u < DT_U; ++u) {
            const int r = max(r0 - u, 0);
            xs[u] = f[(long)r * W];
            vs[u] = d[(long)(r + 1) * W];
        }
#pragma unroll
        for (int u = 0; u < DT_U; ++u) {
            if (r0 - u >= 0) {
                const float v = expf(vs[u] * log_a);
                prev = xs[u] + v * (prev - xs[u]);
                f[(long)(r0 - u) * W] = prev;
            }
        }
    }
}

// ---- the same filter without the domain planes (C == 1 or 3) -----------------------------------------------------------
// dom is a function of two neighbouring samples of the joint image, which both passes have in hand or one cache line
// away: recomputing it where it is used removes dt_domain_kernel, its two fp32 planes per image (written once, read
// twice per pass) and the fp32 copy of the input -- 53 B per sample become 32 for fp32 images (44 -> 28 for fp16).
// Same expressions as the kernels above (the wave scan associates its products differently: last-bit differences).
// Wave-level inclusive scan of affine maps (shared slope ma, one intercept per channel) with DPP moves instead of LDS
// permutes: four row_shr steps inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into
// rows 2 and 3.  Lanes without a source take the identity map (1, 0), under which compose() returns its argument exactly.
// Round 6: a step is the instructions themselves -- v_fmac_f32_dpp mb, mb(lane - k), ma and v_mul_f32_dpp ma, ma(lane - k), ma, in
// place: a lane without a source, or in a masked row, is not written and keeps its map, which is what composing with the identity
// gave it (the builtin form spent a move of the identity and a move_dpp per operand on top of the arithmetic: 72 instructions
// per scan of three channels against 24; the row kernels are bound by instruction issue once the chip is full).  The products
// and sums are those of compose().  s_nop 1: a DPP operand wants two wait states behind the VALU write of its register, and the
// compiler's hazard pass does not look inside asm.
#define PB_DPP_SHR1 "row_shr:1 row_mask:0xf bank_mask:0xf"
#define PB_DPP_SHR2 "row_shr:2 row_mask:0xf bank_mask:0xf"
#define PB_DPP_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf"
#define PB_DPP_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf"
#define PB_DPP_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define PB_DPP_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
#define PB_SCAN_STEP1(CTRL)                                                                                          \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " CTRL "\n\tv_mul_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 1"            \
        : "+v"(mb[0]), "+v"(ma))
#define PB_SCAN_STEP3(CTRL)                                                                                          \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %3 " CTRL "\n\tv_fmac_f32_dpp %1, %1, %3 " CTRL                          \
        "\n\tv_fmac_f32_dpp %2, %2, %3 " CTRL "\n\tv_mul_f32_dpp %3, %3, %3 " CTRL                                  \
        : "+v"(mb[0]), "+v"(mb[1]), "+v"(mb[2]), "+v"(ma))
template <int C> __device__ __forceinline__ void scan_affine(float &ma, float (&mb)[C]);
template <> __device__ __forceinline__ void scan_affine<1>(float &ma, float (&mb)[1]) {
    PB_SCAN_STEP1(PB_DPP_SHR1); PB_SCAN_STEP1(PB_DPP_SHR2); PB_SCAN_STEP1(PB_DPP_SHR4); PB_SCAN_STEP1(PB_DPP_SHR8);
    PB_SCAN_STEP1(PB_DPP_BC15); PB_SCAN_STEP1(PB_DPP_BC31);
}
template <> __device__ __forceinline__ void scan_affine<3>(float &ma, float (&mb)[3]) {
    PB_SCAN_STEP3(PB_DPP_SHR1); PB_SCAN_STEP3(PB_DPP_SHR2); PB_SCAN_STEP3(PB_DPP_SHR4); PB_SCAN_STEP3(PB_DPP_SHR8);
    PB_SCAN_STEP3(PB_DPP_BC15); PB_SCAN_STEP3(PB_DPP_BC31);
    asm("s_nop 1" : "+v"(mb[0]), "+v"(mb[1]), "+v"(mb[2]), "+v"(ma));
}

template <typename TJ, typename TIN, int C>
__global__ __launch_bounds__(NT) void dt_rows_fused_kernel(const TJ *J, const TIN *in, float *F, int H, int W, float ratio,
                                                           float log_a, long rows_total) {
    const int lane = threadIdx.x & 63;
    const long row_id = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);   // over B*H: one wave = one row, all channels
    if (row_id >= rows_total) return;
    const long b = row_id / H;
    const int r = (int)(row_id - b * H);
    const long HW = (long)H * W, off = (b * C * H + r) * (long)W;
    const TJ *j = J + off;
    const TIN *x0 = in + off;
    float *f = F + off;
    // ---- left -> right
    float carry[C];
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = 0.f;
    for (int base = 0; base < W; base += 64) {
        const int i = base + lane;
        float dx = 0.f, x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            x[c] = 0.f;
            if (i < W) {
                x[c] = pb_ld(x0 + c * HW + i);
                if (i > 0) dx += fabsf(pb_ld(j + c * HW + i) - pb_ld(j + c * HW + i - 1));
            }
        }
        float v = 1.f;
        if (i < W) v = expf((1.f + ratio * dx) * log_a);
        float ma = (i == 0 || i >= W) ? 0.f : v, mb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mb[c] = (i == 0 || i >= W) ? x[c] : (1.f - v) * x[c];
        if (i >= W) {
            ma = 1.f;
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = 0.f;
        }
        scan_affine<C>(ma, mb);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = fmaf(ma, carry[c], mb[c]);
            if (i < W) f[c * HW + i] = y;
            carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), base + 63 >= W ? (W - 1) - base : 63));
        }
    }
    __threadfence_block();                               // the right-to-left pass reads what other lanes have just written
    // ---- right -> left
    const int last_base = ((W - 1) / 64) * 64;
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = 0.f;
    for (int base = last_base; base >= 0; base -= 64) {
        const int i = base + (63 - lane);                // lane 0 handles the right-most element of the chunk
        float dx = 0.f, x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            x[c] = 0.f;
            if (i < W) x[c] = f[c * HW + i];
            if (i + 1 < W) dx += fabsf(pb_ld(j + c * HW + i + 1) - pb_ld(j + c * HW + i));
        }
        float v = 1.f;
        if (i + 1 < W) v = expf((1.f + ratio * dx) * log_a);
        float ma = (i >= W - 1) ? 0.f : v, mb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mb[c] = (i >= W - 1) ? x[c] : (1.f - v) * x[c];
        if (i >= W) {
            ma = 1.f;
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = 0.f;
        }
        scan_affine<C>(ma, mb);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = fmaf(ma, carry[c], mb[c]);
            if (i < W) f[c * HW + i] = y;
            carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));  // element `base`, the left-most of this chunk
        }
    }
}

// ---- the row pass with the row in registers (round 5) -------------------------------------------------------------------
// A row of W <= 64 NCH samples of C channels is NCH x C values per lane: the wave reads its row ONCE (every load of the row in
// flight together), runs the left-to-right recurrence chunk by chunk, then the right-to-left one on the registers, and
// writes the row once -- where dt_rows_fused_kernel writes the intermediate row, reads it back and reads the joint image a
// second time (4.9 words per sample through HBM on 64 x 1080p against the 2 of one read and one write:
// profiles/r04_bench_cfg3_traffic.json).  Only for J == in (the pipeline's call, domain_transform.py:45-47 with joint=None):
// the neighbour differences |J[i] - J[i-1]| then come from the registers too (lane - 1, or lane 63 of the previous chunk).
// The recurrences, their operands and the order in which scan_affine composes them are those of dt_rows_fused_kernel: the
// right-to-left pass runs on lane-reversed values (lane l <-> sample base + 63 - l), exactly as that kernel lays them out --
// bit-identical results (tests/test_gpu_parity.py::test_dt_rows_register_form).
template <typename TIN, int C, int NCH>
__global__ __launch_bounds__(NT, NCH == 32 ? 3 : 1) void dt_rows_reg_kernel(const TIN *in, float *F, int H, int W, float ratio, float log_a, long rows_total) {   // (32 chunks: three waves per SIMD -- 168 registers)
    const int lane = threadIdx.x & 63;
    const long row_id = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);   // over B*H: one wave = one row, all channels
    if (row_id >= rows_total) return;
    const long b = row_id / H;
    const int r = (int)(row_id - b * H);
    const long HW = (long)H * W, off = (b * C * H + r) * (long)W;
    const TIN *x0 = in + off;
    float *f = F + off;
    const int nch = (W + 63) >> 6;
    float x[NCH][C], v[NCH];
    // (addresses clamped into the row, not loads under a lane mask: a masked fp16 load is followed by its conversion inside
    //  the masked block and by a wait for it, which put the 96 loads of a 1080p row one after the other -- 1.55 ms on
    //  64 x 1080p fp16 against 0.77 ms for the same row in fp32.
    //  Only the row's last chunk is ragged: every other chunk's address is a uniform base plus the lane -- one offset
    //  register for all of them; a clamp per chunk cost 35 registers and the third wave per SIMD.)
    TIN raw[NCH][C];
    const int klast = nch - 1, lane_last = min(lane, W - 1 - 64 * klast);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const TIN *xk = x0 + 64 * min(k, klast) + (k < klast ? lane : lane_last);      // (chunks past the row: the last chunk again)
#pragma unroll
        for (int c = 0; c < C; ++c) raw[k][c] = xk[c * HW];
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = 64 * k + lane;
#pragma unroll
        for (int c = 0; c < C; ++c) x[k][c] = (i < W) ? pb_ld(&raw[k][c]) : 0.f;
    }
    // ---- left -> right
    float carry[C], last[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { carry[c] = 0.f; last[c] = 0.f; }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (k < nch) {
            const int base = 64 * k, i = base + lane;
            float dx = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // sample i - 1: the lane before (wave_shr:1), lane 0 keeps the operand's old value -- lane 63 of the previous chunk
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(last[c]), __float_as_int(x[k][c]), 0x138, 0xf, 0xf, false));
                last[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[k][c]), 63));
                if (i < W && i > 0) dx += fabsf(x[k][c] - left);
            }
            float vv = 1.f;
            if (i < W) vv = expf((1.f + ratio * dx) * log_a);
            v[k] = vv;
            float ma = (i == 0 || i >= W) ? 0.f : vv, mb[C];
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = (i == 0 || i >= W) ? x[k][c] : (1.f - vv) * x[k][c];
            if (i >= W) {
                ma = 1.f;
#pragma unroll
                for (int c = 0; c < C; ++c) mb[c] = 0.f;
            }
            scan_affine<C>(ma, mb);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float y = fmaf(ma, carry[c], mb[c]);
                x[k][c] = y;
                carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), base + 63 >= W ? (W - 1) - base : 63));
            }
        } else {
            v[k] = 1.f;
        }
    }
    // ---- right -> left, on lane-reversed values: lane l <-> sample i = base + 63 - l; its weight is V[i + 1]
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = 0.f;
    float vnext0 = 1.f;                                                  // V[first sample of the chunk to the right]
#pragma unroll
    for (int k = NCH - 1; k >= 0; --k) {
        if (k < nch) {
            const int base = 64 * k, i = base + (63 - lane);
            float xr[C];
#pragma unroll
            for (int c = 0; c < C; ++c) xr[c] = __shfl(x[k][c], 63 - lane);
            float vr = __shfl(v[k], (64 - lane) & 63);                       // V[base + 64 - l], l >= 1
            if (lane == 0) vr = vnext0;
            vnext0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 0));
            const float vv = (i + 1 < W) ? vr : 1.f;
            float ma = (i >= W - 1) ? 0.f : vv, mb[C];
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = (i >= W - 1) ? xr[c] : (1.f - vv) * xr[c];
            if (i >= W) {
                ma = 1.f;
#pragma unroll
                for (int c = 0; c < C; ++c) mb[c] = 0.f;
            }
            scan_affine<C>(ma, mb);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float y = fmaf(ma, carry[c], mb[c]);
                if (i < W) f[c * HW + i] = y;
                carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));  // sample `base`, the left-most of this chunk
            }
        }
    }
}

// ---- the register row pass, a row over the four waves of a workgroup (round 6) ---------------------------------------------------
// One wave per row spends ~0.5 us per 64-sample chunk in the two scans whatever feeds it: a lone 4K image's 2160 rows are 2160
// waves of 60 chunks each -- 162 us for 200 MB.  The scans of a row's chunks do not depend on one another; only the carry does,
// and that is one multiply-add per chunk and channel.  Here wave w of the workgroup holds chunks w CPW .. w CPW + CPW - 1 of the
// row, scans them, leaves the map of each chunk's last sample in LDS, and every wave then walks the carries of the chunks before
// its own: fmaf(a, carry, b) on the last sample's map is what lane 63's y = fmaf(ma, carry, mb) was in dt_rows_reg_kernel, so
// the carries, and with them every sample, are the same bits.  The right-to-left pass likewise (chunk k's first weight comes from
// chunk k + 1: through LDS at the waves' seams).  Rows of up to 256 CPW samples: CPW = 32 takes an 8K row.
template <typename TIN, int C, int CPW>
__global__ __launch_bounds__(NT) void dt_rows_regw_kernel(const TIN *in, float *F, int H, int W, float ratio, float log_a) {
    constexpr int WPR = NT / 64, NCH = WPR * CPW;
    __shared__ float tail[2][NCH][C + 1];              // [pass][chunk]: the map (a, b[c]) of the chunk's last sample
    __shared__ float vfirst[NCH + 1];                  // the weight of each chunk's first sample
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, k0 = wv * CPW;
    const long row_id = blockIdx.x;                    // over B*H: one workgroup = one row, all channels
    const long b = row_id / H;
    const int r = (int)(row_id - b * H);
    const long HW = (long)H * W, off = (b * C * H + r) * (long)W;
    const TIN *x0 = in + off;
    float *f = F + off;
    const int nch = (W + 63) >> 6;
    float x[CPW][C], v[CPW], ma[CPW];
    TIN raw[CPW][C], rawl[C];
    const int klast = nch - 1, lane_last = min(lane, W - 1 - 64 * klast);
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int k = k0 + q;
        const TIN *xk = x0 + 64 * min(k, klast) + (k < klast ? lane : lane_last);      // (chunks past the row: the last chunk again)
#pragma unroll
        for (int c = 0; c < C; ++c) raw[q][c] = xk[c * HW];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) rawl[c] = x0[c * HW + min(max(64 * k0 - 1, 0), W - 1)];   // the sample left of this wave's first
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int i = 64 * (k0 + q) + lane;
#pragma unroll
        for (int c = 0; c < C; ++c) x[q][c] = (i < W) ? pb_ld(&raw[q][c]) : 0.f;
    }
    // ---- left -> right: the scans
    float last[C];
#pragma unroll
    for (int c = 0; c < C; ++c) last[c] = k0 > 0 ? pb_ld(&rawl[c]) : 0.f;
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int k = k0 + q, base = 64 * k, i = base + lane;
        float dx = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(last[c]), __float_as_int(x[q][c]), 0x138, 0xf, 0xf, false));
            last[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[q][c]), 63));
            if (i < W && i > 0) dx += fabsf(x[q][c] - left);
        }
        float vv = 1.f;
        if (i < W) vv = expf((1.f + ratio * dx) * log_a);
        v[q] = vv;
        float a = (i == 0 || i >= W) ? 0.f : vv, mb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mb[c] = (i == 0 || i >= W) ? x[q][c] : (1.f - vv) * x[q][c];
        if (i >= W) {
            a = 1.f;
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = 0.f;
        }
        scan_affine<C>(a, mb);
        ma[q] = a;
#pragma unroll
        for (int c = 0; c < C; ++c) x[q][c] = mb[c];
        if (lane == (base + 63 >= W ? max((W - 1) - base, 0) : 63)) {
            tail[0][k][0] = a;
#pragma unroll
            for (int c = 0; c < C; ++c) tail[0][k][1 + c] = mb[c];
        }
        if (lane == 0) vfirst[k] = vv;
    }
    if (threadIdx.x == 0) vfirst[NCH] = 1.f;
    __syncthreads();
    // ... the carries of the chunks before this wave's, then its own samples
    float carry[C];
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = 0.f;
    for (int k = 0; k < min(k0, nch); ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) carry[c] = fmaf(tail[0][k][0], carry[c], tail[0][k][1 + c]);
    }
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int base = 64 * (k0 + q);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = fmaf(ma[q], carry[c], x[q][c]);
            x[q][c] = y;
            carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), base + 63 >= W ? max((W - 1) - base, 0) : 63));
        }
    }
    // ---- right -> left, on lane-reversed values: lane l <-> sample i = base + 63 - l; its weight is V[i + 1]
#pragma unroll
    for (int q = CPW - 1; q >= 0; --q) {
        const int k = k0 + q, base = 64 * k, i = base + (63 - lane);
        float xr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xr[c] = __shfl(x[q][c], 63 - lane);
        float vr = __shfl(v[q], (64 - lane) & 63);                           // V[base + 64 - l], l >= 1
        const float vnext0 = q + 1 < CPW ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q + 1 < CPW ? q + 1 : q]), 0)) : vfirst[k + 1];
        if (lane == 0) vr = k + 1 < nch ? vnext0 : 1.f;
        const float vv = (i + 1 < W) ? vr : 1.f;
        float a = (i >= W - 1) ? 0.f : vv, mb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) mb[c] = (i >= W - 1) ? xr[c] : (1.f - vv) * xr[c];
        if (i >= W) {
            a = 1.f;
#pragma unroll
            for (int c = 0; c < C; ++c) mb[c] = 0.f;
        }
        scan_affine<C>(a, mb);
        ma[q] = a;
#pragma unroll
        for (int c = 0; c < C; ++c) x[q][c] = mb[c];
        if (lane == 63) {
            tail[1][k][0] = a;
#pragma unroll
            for (int c = 0; c < C; ++c) tail[1][k][1 + c] = mb[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = 0.f;
    for (int k = nch - 1; k >= k0 + CPW; --k) {
#pragma unroll
        for (int c = 0; c < C; ++c) carry[c] = fmaf(tail[1][k][0], carry[c], tail[1][k][1 + c]);
    }
#pragma unroll
    for (int q = CPW - 1; q >= 0; --q) {
        const int k = k0 + q, i = 64 * k + (63 - lane);
        if (k < nch) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float y = fmaf(ma[q], carry[c], x[q][c]);
                if (i < W) f[c * HW + i] = y;
                carry[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));  // sample `base`, the left-most of this chunk
            }
        }
    }
}

constexpr int DT_UF = 8;
constexpr long DT_ROWSW_MAX = 8192;                // rows in the batch up to which a row is spread over four waves (tools/bench_dt.py)
template <typename TJ, int C>
__global__ __launch_bounds__(NT) void dt_cols_fused_kernel(const TJ *__restrict__ J, float *__restrict__ F, int H, int W,
                                                           float ratio, float log_a, long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;   // over B*W: one thread = one column, all channels
    if (id >= cols_total) return;
    const long b = id / W;
    const int col = (int)(id - b * W);
    const long HW = (long)H * W;
    float *f = F + b * C * HW + col;
    const TJ *j = J + b * C * HW + col;
    float prev[C], pj[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev[c] = f[c * HW]; pj[c] = pb_ld(j + c * HW); }
    for (int r0 = 1; r0 < H; r0 += DT_UF) {
        float xs[DT_UF][C], js[DT_UF][C];
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            const int r = min(r0 + u, H - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) { xs[u][c] = f[c * HW + (long)r * W]; js[u][c] = pb_ld(j + c * HW + (long)r * W); }
        }
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            if (r0 + u < H) {
                float dy = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) { dy += fabsf(js[u][c] - pj[c]); pj[c] = js[u][c]; }
                const float v = expf((1.f + ratio * dy) * log_a);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    prev[c] = xs[u][c] + v * (prev[c] - xs[u][c]);
                    f[c * HW + (long)(r0 + u) * W] = prev[c];
                }
            }
        }
    }
    // pj = J[H-1], prev = F[H-1]
    for (int r0 = H - 2; r0 >= 0; r0 -= DT_UF) {
        float xs[DT_UF][C], js[DT_UF][C];
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            const int r = max(r0 - u, 0);
#pragma unroll
            for (int c = 0; c < C; ++c) { xs[u][c] = f[c * HW + (long)r * W]; js[u][c] = pb_ld(j + c * HW + (long)r * W); }
        }
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            if (r0 - u >= 0) {
                float dy = 0.f;                                      // dom of row r+1: |J[r+1] - J[r]|
#pragma unroll
                for (int c = 0; c < C; ++c) { dy += fabsf(pj[c] - js[u][c]); pj[c] = js[u][c]; }
                const float v = expf((1.f + ratio * dy) * log_a);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    prev[c] = xs[u][c] + v * (prev[c] - xs[u][c]);
                    f[c * HW + (long)(r0 - u) * W] = prev[c];
                }
            }
        }
    }
}

// ---- the column pass in strips (round 5) -----------------------------------------------------------------------------------
// dt_cols_fused_kernel moves 6 words per sample: the down sweep reads F and J and writes F, the up sweep reads both again
// and writes F again.  The up sweep at row r needs the down sweep's value of row r and the up sweep's of row r + 1, so the
// down sweep has to reach the bottom first -- but not to WRITE anything on the way except what lets its values be formed
// again: one carry per strip of S rows (dt_cols_down_kernel: 2 words per sample read, 1 / S written).  dt_cols_up_kernel
// then takes the strips bottom-up: the strip's rows of F and J into registers, the down sweep's values of the strip formed
// again from the carry above it -- the same operations on the same operands in the same order: the same bits --, the up
// sweep over them, one store per sample: 3 words.  5 words instead of 6, and one exponential per sample and column instead
// of two (the up sweep's weight of row r + 1 is the down sweep's, domain_transform.py:62-85: |J[r+1] - J[r]| either way).
// Bit-identical to dt_cols_fused_kernel (tests/test_gpu_round5_forms.py).
// Round 6: where J is C fp32 channels the up sweep reads C words per pixel only to form one weight again, so the down sweep
// stores the weight it has (WV: one word per pixel, 1 / C per sample) and dt_cols_upw_kernel reads that instead of J: 2 / C words
// against 1, 4.67 + 2 / S per sample at C = 3 -- and, no J in its registers, strips of 32 rows.  The weight is the value the
// down sweep used, the operations on it are the same: the same bits again.  (fp16 J, or one channel: J costs no more than the
// weight would -- those keep dt_cols_up_kernel.)
template <int C> __device__ __forceinline__ float dt_weight(const float (&jr)[C], const float (&jp)[C], float ratio, float log_a) {
    float dy = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) dy += fabsf(jr[c] - jp[c]);
    return expf((1.f + ratio * dy) * log_a);
}
constexpr int DT_STRIP = 16, DT_STRIP_W = 32;
template <typename TJ, int C, int S, bool WV>
__global__ __launch_bounds__(NT) void dt_cols_down_kernel(const TJ *__restrict__ J, const float *__restrict__ F, float *__restrict__ carry,
                                                          float *__restrict__ wts, int H, int W, float ratio, float log_a, long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;   // over B*W: one thread = one column, all channels
    if (id >= cols_total) return;
    const long b = id / W;
    const int col = (int)(id - b * W);
    const long HW = (long)H * W;
    const float *f = F + b * C * HW + col;
    const TJ *j = J + b * C * HW + col;
    float *wt = WV ? wts + b * HW + col : nullptr;         // wts[b][r][col] = the weight of row r (rows 1 .. H - 1)
    float prev[C], pj[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev[c] = f[c * HW]; pj[c] = pb_ld(j + c * HW); }
    // carry[s][c][id] = the down sweep's value of the last row of strip s (rows S s .. S s + S - 1), for every strip but the last
    for (int r0 = 1; r0 < H; r0 += DT_UF) {
        float xs[DT_UF][C], js[DT_UF][C];
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            const int r = min(r0 + u, H - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) { xs[u][c] = f[c * HW + (long)r * W]; js[u][c] = pb_ld(j + c * HW + (long)r * W); }
        }
#pragma unroll
        for (int u = 0; u < DT_UF; ++u) {
            const int r = r0 + u;
            if (r < H) {
                const float v = dt_weight<C>(js[u], pj, ratio, log_a);
                if constexpr (WV) wt[(long)r * W] = v;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    pj[c] = js[u][c];
                    prev[c] = xs[u][c] + v * (prev[c] - xs[u][c]);
                }
                if ((r & (S - 1)) == S - 1 && r + 1 < H) {
#pragma unroll
                    for (int c = 0; c < C; ++c) carry[((long)(r / S) * C + c) * cols_total + id] = prev[c];
                }
            }
        }
    }
}
// the up sweep over strips of S rows with the down sweep's weights read back (see above); F only, no J
template <int C, int S>
__global__ __launch_bounds__(NT) void dt_cols_upw_kernel(float *__restrict__ F, const float *__restrict__ carry, const float *__restrict__ wts,
                                                         int H, int W, long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;
    if (id >= cols_total) return;
    const long b = id / W;
    const int col = (int)(id - b * W);
    const long HW = (long)H * W;
    float *f = F + b * C * HW + col;
    const float *wt = wts + b * HW + col;
    const int nstrips = (H + S - 1) / S;
    float fnext[C], vnext = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) fnext[c] = 0.f;
    for (int s = nstrips - 1; s >= 0; --s) {
        const int a = s * S;
        float xs[S][C], prev[C], v[S];
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int r = min(a + u, H - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) xs[u][c] = f[c * HW + (long)r * W];
            v[u] = wt[(long)max(r, 1) * W];                  // (row 0 has no weight: its slot is never written, never used)
        }
        if (s > 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) prev[c] = carry[((long)(s - 1) * C + c) * cols_total + id];
        }
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int r = a + u;
            if (r == 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) prev[c] = xs[u][c];
            } else if (r < H) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    prev[c] = xs[u][c] + v[u] * (prev[c] - xs[u][c]);
                    xs[u][c] = prev[c];
                }
            }
        }
#pragma unroll
        for (int u = S - 1; u >= 0; --u) {
            const int r = a + u;
            if (r < H) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (r < H - 1) fnext[c] = xs[u][c] + vnext * (fnext[c] - xs[u][c]);
                    else fnext[c] = xs[u][c];
                    f[c * HW + (long)r * W] = fnext[c];
                }
                vnext = v[u];
            }
        }
    }
}
template <typename TJ, int C>
__global__ __launch_bounds__(NT) void dt_cols_up_kernel(const TJ *__restrict__ J, float *__restrict__ F, const float *__restrict__ carry,
                                                        int H, int W, float ratio, float log_a, long cols_total) {
    const long id = (long)blockIdx.x * NT + threadIdx.x;
    if (id >= cols_total) return;
    const long b = id / W;
    const int col = (int)(id - b * W);
    const long HW = (long)H * W;
    float *f = F + b * C * HW + col;
    const TJ *j = J + b * C * HW + col;
    const int nstrips = (H + DT_STRIP - 1) / DT_STRIP;
    float fnext[C], vnext = 0.f;                            // the up sweep's value of the row below the strip, and that row's weight
#pragma unroll
    for (int c = 0; c < C; ++c) fnext[c] = 0.f;
    for (int s = nstrips - 1; s >= 0; --s) {
        const int a = s * DT_STRIP;
        float xs[DT_STRIP][C], js[DT_STRIP][C], pj[C], prev[C], v[DT_STRIP];
        // (rows past the plane's end: the last row again -- loaded, never used)
#pragma unroll
        for (int u = 0; u < DT_STRIP; ++u) {
            const int r = min(a + u, H - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) { xs[u][c] = f[c * HW + (long)r * W]; js[u][c] = pb_ld(j + c * HW + (long)r * W); }
        }
        if (s > 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) { prev[c] = carry[((long)(s - 1) * C + c) * cols_total + id]; pj[c] = pb_ld(j + c * HW + (long)(a - 1) * W); }
        }
        // the down sweep's values of the strip, formed again (row 0 keeps its value: domain_transform.py:62-72)
#pragma unroll
        for (int u = 0; u < DT_STRIP; ++u) {
            const int r = a + u;
            if (r == 0) {
                v[u] = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) { prev[c] = xs[u][c]; pj[c] = js[u][c]; }
            } else if (r < H) {
                v[u] = dt_weight<C>(js[u], pj, ratio, log_a);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    pj[c] = js[u][c];
                    prev[c] = xs[u][c] + v[u] * (prev[c] - xs[u][c]);
                    xs[u][c] = prev[c];
                }
            } else {
                v[u] = 0.f;
            }
        }
        // the up sweep over the strip (the last row of the plane keeps the down sweep's value)
#pragma unroll
        for (int u = DT_STRIP - 1; u >= 0; --u) {
            const int r = a + u;
            if (r < H) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (r < H - 1) fnext[c] = xs[u][c] + vnext * (fnext[c] - xs[u][c]);
                    else fnext[c] = xs[u][c];
                    f[c * HW + (long)r * W] = fnext[c];
                }
                vnext = v[u];
            }
        }
    }
}

// ---- the column pass of few columns (round 6) -------------------------------------------------------------------------------
// One thread per column is B W threads: a single 1080p image is 30 waves, each allowed 63 requests in flight -- 16 KB -- against
// a latency of ~1.3 us: 0.37 TB/s however the loop is written (measured: down + up sweep 176 + 206 us at B = 1 and at B = 4
// alike, where the planes' 250 MB would take 50 us).  Here a workgroup owns CG adjacent columns (16 of three channels, 64 of one:
// 64 bytes per row and channel) and ALL its lanes fetch: blocks of DTC_R3 (one channel: DTC_R1) rows of F and J straight into LDS (16 bytes
// per lane, waves 1 - 3; the next block in flight under the present one's recurrence), all lanes form the block's weights, and wave 0
// -- lane = (channel, column) -- runs the recurrence down the block out of LDS; carries per block; then the blocks bottom-up: the
// down sweep's values of the block formed again from the carry above it, the up sweep over them, the rows stored 16 bytes per lane
// one block late, under the next block's recurrence.  Rows 0 and H - 1 need no case of their own: their weights are 0 and
// x + 0 (p - x) = x (a sample that is -0 comes out as +0: the only difference in bits there can be).  The same operations on the same operands in the same order as dt_cols_fused_kernel: the same bits
// (tests/test_gpu_round5_forms.py::test_dt_columns_in_strips).  5 words per sample like the strips; the launch is taken where the
// per-column form cannot fill the chip (dt_filter_fused: cols_total up to DTC_MAX_COLS).
typedef __amdgpu_buffer_rsrc_t dt_rsrc;
typedef __attribute__((address_space(3))) void dt_lds_void;
constexpr unsigned DTC_NO_ACCESS = 0x80000000u;        // (an image's planes are smaller than 2 GiB: checked on the host)
constexpr int DTC_R3 = 96, DTC_R1 = 64;            // rows of a block: three channels (78 KB of LDS, two workgroups per CU), one channel
constexpr int DTC_CH = 32;                          // rows the recurrence's wave holds in registers at a time
constexpr long DTC_MAX_COLS = 24576;              // up to here the per-column forms leave CUs idle (tools/bench_dt.py)
__device__ __forceinline__ void dtc_dma16(dt_rsrc r, void *dst, unsigned voffset) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dt_lds_void *)dst, 16, (int)voffset, 0, 0, 0);
#pragma clang diagnostic pop
}
template <typename TJ, int C, int R> struct DtCoop {
    static constexpr int CG = C == 1 ? 64 : 16, L = C * CG;
    static constexpr int FB = R * L * 4;                                                     // bytes of a block of F: whole KB
    static constexpr int JB = ((R + 1) * L * (int)sizeof(TJ) + 1023) / 1024 * 1024;         // of J: the row above the block too
    static constexpr int LDS = 2 * FB + 2 * JB + R * CG * 4;
    static_assert(FB % 1024 == 0 && R % DTC_CH == 0, "a block of F is whole wave requests and whole chunks of the chain");
};
template <typename TJ, int C, int R>
__global__ __launch_bounds__(NT) void dt_cols_coop_kernel(const TJ *__restrict__ J, float *__restrict__ F, float *__restrict__ carry,
                                                          int H, int W, float ratio, float log_a, int groups) {
    using G = DtCoop<TJ, C, R>;
    constexpr int CG = G::CG, L = G::L, FB = G::FB, JB = G::JB, CH = DTC_CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char dtc_smem[];
    unsigned char *smem = dtc_smem;
    float *vb = reinterpret_cast<float *>(smem + 2 * FB + 2 * JB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / groups, col0 = (blockIdx.x - b * groups) * CG;
    const long HW = (long)H * W;
    const int nb = (H + R - 1) / R;
    float *Fi = F + (long)b * C * HW;
    const dt_rsrc rF = __builtin_amdgcn_make_buffer_rsrc(Fi, 0, (int)(C * HW * 4), 0x00020000);
    const dt_rsrc rJ = __builtin_amdgcn_make_buffer_rsrc(const_cast<TJ *>(J + (long)b * C * HW), 0, (int)(C * HW * (long)sizeof(TJ)), 0x00020000);
    float *cw = carry + (long)blockIdx.x * nb * L;
    auto fbuf = [&](int k) { return reinterpret_cast<float *>(smem + (k & 1) * FB); };
    auto jbuf = [&](int k) { return reinterpret_cast<TJ *>(smem + 2 * FB + (k & 1) * JB); };
    // (waves 1 .. 3 fetch; wave 0 runs the recurrence and has no request of its own to wait for in the middle of it)
    auto request = [&](int k) {
        if (wave == 0) return;
        const int r0 = k * R;
        for (int i = wave - 1; i < FB / 1024; i += NT / 64 - 1) {
            const int e = (i * 1024 + lane * 16) / 4;
            const int u = e / L, idx = e - u * L, c = idx / CG, col = idx - c * CG;
            const bool ok = r0 + u < H && col0 + col < W;
            dtc_dma16(rF, reinterpret_cast<unsigned char *>(fbuf(k)) + i * 1024, ok ? (unsigned)((c * HW + (long)(r0 + u) * W + col0 + col) * 4) : DTC_NO_ACCESS);
        }
        for (int i = wave - 1; i < JB / 1024; i += NT / 64 - 1) {
            const int e = (i * 1024 + lane * 16) / (int)sizeof(TJ);
            const int u = e / L, idx = e - u * L, c = idx / CG, col = idx - c * CG, row = r0 - 1 + u;
            const bool ok = u <= R && row >= 0 && row < H && col0 + col < W;
            dtc_dma16(rJ, reinterpret_cast<unsigned char *>(jbuf(k)) + i * 1024, ok ? (unsigned)((c * HW + (long)row * W + col0 + col) * (long)sizeof(TJ)) : DTC_NO_ACCESS);
        }
    };
    // the weights of the block's rows (row 0 of the plane has none)
    auto weights = [&](int k) {
        const TJ *jb = jbuf(k);
        for (int p = tid; p < R * CG; p += NT) {
            const int u = p / CG, col = p - u * CG, r = k * R + u;
            float jr[C], jp[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { jr[c] = pb_ld(jb + (u + 1) * L + c * CG + col); jp[c] = pb_ld(jb + u * L + c * CG + col); }
            vb[p] = r >= 1 && r < H ? dt_weight<C>(jr, jp, ratio, log_a) : 0.f;
        }
    };
    // Order of a block's steps: the requests go out AFTER the weights (the compiler waits for every outstanding request before an
    // LDS read that follows one: behind the weights a request flies under the recurrence instead of being waited for), and the
    // up sweep's rows are stored one step late, by the fetching waves, while wave 0 is in the next block's recurrence.
    // Rows 0 and H - 1 need no case of their own: the weight of row 0, and of rows past the end, is 0 (weights()) and
    // x + 0 (p - x) = x -- the row keeps its value --; rows past the end hold the zeros the requests returned.
    const int ccol = lane % CG;
    auto store_rows = [&](int k) {
        const float *fb = fbuf(k);
        for (int i = wave - 1; i < FB / 1024; i += NT / 64 - 1) {    // (a wave stores the KB it fetches: its next request into them follows its own reads)
            const int e = i * 256 + lane * 4, u = e / L, idx = e - u * L, c = idx / CG, col = idx - c * CG, r = k * R + u;
            if (r < H && col0 + col < W) *reinterpret_cast<float4 *>(Fi + c * HW + (long)r * W + col0 + col) = *reinterpret_cast<const float4 *>(fb + e);
        }
    };
    float prev = 0.f;
    request(0);
    for (int k = 0; k < nb; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        weights(k);
        __syncthreads();
        if (k + 1 < nb) {                                        // (the last block's down values are formed by the up sweep below)
            if (wave > 0) {
                request(k + 1);
            } else if (lane < L) {
                const float *fb = fbuf(k);
                for (int q = 0; q < R; q += CH) {
                    float x[CH], v[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) { x[u] = fb[(q + u) * L + lane]; v[u] = vb[(q + u) * CG + ccol]; }
#pragma unroll
                    for (int u = 0; u < CH; ++u) prev = x[u] + v[u] * (prev - x[u]);
                }
                cw[k * L + lane] = prev;
            }
        }
    }
    float fnext = 0.f, vnext = 0.f;
    for (int k = nb - 1; k >= 0; --k) {
        if (k != nb - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            weights(k);
            __syncthreads();
        }
        if (wave > 0) {
            if (k + 1 < nb) store_rows(k + 1);
            if (k > 0) request(k - 1);                           // (into the buffer of block k + 1, read out just above)
        } else if (lane < L) {
            float *fb = fbuf(k);
            // the down sweep's values of the block, into LDS in place, ...
            prev = k > 0 ? cw[(k - 1) * L + lane] : 0.f;
            for (int q = 0; q < R; q += CH) {
                float x[CH], v[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) { x[u] = fb[(q + u) * L + lane]; v[u] = vb[(q + u) * CG + ccol]; }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    prev = x[u] + v[u] * (prev - x[u]);
                    fb[(q + u) * L + lane] = prev;
                }
            }
            // ... and the up sweep over them
            for (int q = R - CH; q >= 0; q -= CH) {
                float x[CH], v[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) { x[u] = fb[(q + u) * L + lane]; v[u] = vb[(q + u) * CG + ccol]; }
#pragma unroll
                for (int u = CH - 1; u >= 0; --u) {
                    fnext = x[u] + vnext * (fnext - x[u]);
                    fb[(q + u) * L + lane] = fnext;
                    vnext = v[u];
                }
            }
        }
    }
    __syncthreads();
    if (wave > 0) store_rows(0);
}

template <typename T> __global__ void to_float_kernel(const T *__restrict__ in, float *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = pb_ld(in + i);
}
template <typename T> __global__ void from_float_kernel(const float *__restrict__ in, T *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) pb_st(out + i, in[i]);
}

unsigned grid_for(long n, int per_block = NT, int cap = 8192) {
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

// ---- internal entry points used by api.hip --------------------------------------------------
int pb_grad_energy(pb_ctx *ctx, const void *gx, const void *gy, float *nM, int P, long HW, int g_dtype) {
    ProfScope prof(ctx, PB_PROF_HALO);
    int bpp = (int)((HW + NT * 16 - 1) / (NT * 16));
    if (bpp > 256) bpp = 256;
    if (bpp < 1) bpp = 1;
    float *partial = static_cast<float *>(pb_scratch(ctx, "halo.partial", sizeof(float) * (size_t)P * bpp));
    if (!partial) return PB_ERR_NOMEM;
    const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gy)) % 16 == 0;
#define PB_GE(TG, VEC)                                                                                                             \
    hipLaunchKernelGGL((grad_energy_kernel<TG, VEC>), dim3(P * bpp), dim3(NT), 0, ctx->stream, static_cast<const TG *>(gx),       \
                       static_cast<const TG *>(gy), partial, HW, bpp)
    if (g_dtype == PB_F16) { if (vec) PB_GE(__half, true); else PB_GE(__half, false); }
    else { if (vec) PB_GE(float, true); else PB_GE(float, false); }
#undef PB_GE
    hipLaunchKernelGGL(grad_energy_fold_kernel, dim3(P), dim3(64), 0, ctx->stream, partial, nM, bpp);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_halo_apply(pb_ctx *ctx, const void *x, int x_dtype, int x_pitch, long x_plane, const float *y, const void *gx,
                  const void *gy, const void *ox, const float *nM, void *out, int out_dtype, int P, int H, int W,
                  int clamp01, const void *recomb_cur, int recomb_cur_dtype, const float *recomb_smooth, int g_dtype) {
    // four samples per thread where every row of every plane starts on a boundary of four samples (and x's first sample does:
    // the interior of a padded plane starts pad (pp + 1) samples in)
    const size_t xb = x_dtype == PB_F16 ? 2 : 4;
    const bool vec = (W & 3) == 0 && (x_pitch & 3) == 0 && (x_plane & 3) == 0 && (reinterpret_cast<size_t>(x) % (4 * xb)) == 0 &&
                     (long)H * W / 4 < 0xffffffffL;
    dim3 grid(grid_for((long)H * W / (vec ? 4 : 1), NT, 2048), P < 65535 ? P : 65535);
    ProfScope prof(ctx, PB_PROF_HALO);
    if (recomb_cur && recomb_cur_dtype != PB_F32 && recomb_cur_dtype != PB_F16)
        return pb_fail(ctx, PB_ERR_BADARG, "halo: the recombined image must be fp32 or fp16");
    const int cur_is_half = recomb_cur_dtype == PB_F16;
#define PB_HALO_V(TX, TO, TG, VEC)                                                                                 \
    hipLaunchKernelGGL((halo_kernel<TX, TO, TG, VEC>), grid, dim3(NT), 0, ctx->stream, static_cast<const TX *>(x), x_pitch, \
                       x_plane, y, static_cast<const TG *>(gx), static_cast<const TG *>(gy), static_cast<const TG *>(ox), nM, \
                       static_cast<TO *>(out), P, H, W, clamp01, recomb_cur, cur_is_half, recomb_smooth)
#define PB_HALO(TX, TO, TG) do { if (vec) PB_HALO_V(TX, TO, TG, true); else PB_HALO_V(TX, TO, TG, false); } while (0)
#define PB_HALO_G(TX, TO) do { if (g_dtype == PB_F16) PB_HALO(TX, TO, __half); else PB_HALO(TX, TO, float); } while (0)
    if (x_dtype == PB_F32 && out_dtype == PB_F32) PB_HALO_G(float, float);
    else if (x_dtype == PB_F32 && out_dtype == PB_F16) PB_HALO_G(float, __half);
    else if (x_dtype == PB_F16 && out_dtype == PB_F32) PB_HALO_G(__half, float);
    else PB_HALO_G(__half, __half);
#undef PB_HALO_G
#undef PB_HALO
#undef PB_HALO_V
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_recombine(pb_ctx *ctx, const float *y, const void *cur, int cur_dtype, const float *smooth, void *out, int out_dtype,
                 long n) {
    ProfScope prof(ctx, PB_PROF_PREFILTER);
#define PB_REC(TC, TO)                                                                                           \
    hipLaunchKernelGGL((recombine_kernel<TC, TO>), dim3(grid_for(n)), dim3(NT), 0, ctx->stream, y,                \
                       static_cast<const TC *>(cur), smooth, static_cast<TO *>(out), n)
    if (cur_dtype == PB_F32 && out_dtype == PB_F32) PB_REC(float, float);
    else if (cur_dtype == PB_F32) PB_REC(float, __half);
    else if (out_dtype == PB_F32) PB_REC(__half, float);
    else PB_REC(__half, __half);
#undef PB_REC
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_bilateral5_impl(pb_ctx *ctx, const void *in, int in_dtype, void *out, int out_dtype, int P, int H, int W) {
    const double sigma_color = 0.1, sigma_space = 5.0, log2e = 1.4426950408889634;
    const float ivc = (float)(-log2e / (2. * sigma_color * sigma_color)), ivs = (float)(-log2e / (2. * sigma_space * sigma_space));   // (exp2's arguments)
    dim3 grid((W + 63) / 64, (H + 15) / 16, P < 65535 ? P : 65535);
    ProfScope prof(ctx, PB_PROF_PREFILTER);
#define PB_BIL(TI, TO)                                                                                           \
    hipLaunchKernelGGL((bilateral5_kernel<TI, TO>), grid, dim3(NT), 0, ctx->stream, static_cast<const TI *>(in),  \
                       static_cast<TO *>(out), P, H, W, ivc, ivs)
    if (in_dtype == PB_F32 && out_dtype == PB_F32) PB_BIL(float, float);
    else if (in_dtype == PB_F16 && out_dtype == PB_F32) PB_BIL(__half, float);
    else if (in_dtype == PB_F16 && out_dtype == PB_F16) PB_BIL(__half, __half);
    else PB_BIL(float, __half);
#undef PB_BIL
    PB_LAUNCH_CHECK();
    return PB_OK;
}

template <typename T, int C>
static int dt_filter_fused(pb_ctx *ctx, const T *in, const T *J, float *out, int B, int H, int W, float ratio, int N, float sigma_s) {
    const long rows_total = (long)B * H, cols_total = (long)B * W;
    for (int i = 0; i < N; ++i) {
        // domain_transform.py:50,53
        const double sigma_i = (double)sigma_s * std::sqrt(3.0) * std::pow(2.0, N - (i + 1)) / std::sqrt(std::pow(4.0, N) - 1.0);
        const float a = (float)std::exp(-std::sqrt(2.0) / sigma_i);
        const float log_a = std::log(a);
        const dim3 rgrid((unsigned)((rows_total + 3) / 4)), cgrid((unsigned)((cols_total + NT - 1) / NT));
        // (the first iteration of a filter guided by its own input, rows of up to 4096 samples: the row lives in registers)
        const bool reg_rows = i == 0 && J == in && W <= 4096 && ctx->dt_rows_reg && ctx->dt_rows_reg != 3;
        // (few rows: a row over the four waves of a workgroup, dt_rows_regw_kernel -- rows of up to 8192 samples)
        const bool regw_rows = i == 0 && J == in && W > 256 && W <= 8192 && ctx->dt_rows_reg && ctx->dt_rows_reg != 2 &&
                               (ctx->dt_rows_reg == 3 || rows_total <= DT_ROWSW_MAX);
        const dim3 wgrid((unsigned)rows_total);
        if (regw_rows && W <= 1024)
            hipLaunchKernelGGL((dt_rows_regw_kernel<T, C, 4>), wgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a);
        else if (regw_rows && W <= 2048)
            hipLaunchKernelGGL((dt_rows_regw_kernel<T, C, 8>), wgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a);
        else if (regw_rows && W <= 4096)
            hipLaunchKernelGGL((dt_rows_regw_kernel<T, C, 16>), wgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a);
        else if (regw_rows)
            hipLaunchKernelGGL((dt_rows_regw_kernel<T, C, 32>), wgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a);
        else if (reg_rows && W <= 1024)
            hipLaunchKernelGGL((dt_rows_reg_kernel<T, C, 16>), rgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a, rows_total);
        else if (reg_rows && W <= 2048)
            hipLaunchKernelGGL((dt_rows_reg_kernel<T, C, 32>), rgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a, rows_total);
        else if (reg_rows)                                       // (a 4K row: one wave per SIMD, ~300 registers)
            hipLaunchKernelGGL((dt_rows_reg_kernel<T, C, 64>), rgrid, dim3(NT), 0, ctx->stream, in, out, H, W, ratio, log_a, rows_total);
        else if (i == 0)
            hipLaunchKernelGGL((dt_rows_fused_kernel<T, T, C>), rgrid, dim3(NT), 0, ctx->stream, J, in, out, H, W, ratio, log_a, rows_total);
        else
            hipLaunchKernelGGL((dt_rows_fused_kernel<T, float, C>), rgrid, dim3(NT), 0, ctx->stream, J, out, out, H, W, ratio, log_a, rows_total);
        // few columns: workgroups of CG columns whose lanes all fetch (dt_cols_coop_kernel); 16-byte requests want rows, planes and
        // operands on 16-byte boundaries
        {
            constexpr int CG = C == 1 ? 64 : 16;
            const int per16 = 16 / (int)sizeof(T);
            const bool aligned = W % 4 == 0 && W % per16 == 0 && (reinterpret_cast<uintptr_t>(J) | reinterpret_cast<uintptr_t>(out)) % 16 == 0 &&
                                 (long)C * H * W * 4 < (1l << 31);
            if (ctx->dt_cols_coop && aligned && (ctx->dt_cols_coop == 2 || cols_total <= DTC_MAX_COLS)) {
                constexpr int R = C == 1 ? DTC_R1 : DTC_R3;
                using G = DtCoop<T, C, R>;
                const int groups = (W + CG - 1) / CG, nb = (H + R - 1) / R;
                float *cc = static_cast<float *>(pb_scratch(ctx, "dt.carry", sizeof(float) * (size_t)B * groups * nb * C * CG));
                if (!cc) return PB_ERR_NOMEM;
                auto k = dt_cols_coop_kernel<T, C, R>;
                if (G::LDS > 48 * 1024) PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                hipLaunchKernelGGL(k, dim3((unsigned)(B * groups)), dim3(NT), G::LDS, ctx->stream, J, out, cc, H, W, ratio, log_a, groups);
                PB_LAUNCH_CHECK();
                continue;
            }
        }
        float *carry = nullptr, *wts = nullptr;
        // (the weights pay where J costs the up sweep more than a word to write and a word to read per pixel: fp32 J of 3 channels;
        //  PB_DT_COLS_STRIP=2: always J again)
        constexpr bool weights_pay = C * sizeof(T) > 2 * sizeof(float);
        const bool weights = weights_pay && ctx->dt_cols_strip == 1 && H >= 4 * DT_STRIP_W;
        const int strip = weights ? DT_STRIP_W : DT_STRIP;
        if (ctx->dt_cols_strip && H >= 4 * DT_STRIP) {
            carry = static_cast<float *>(pb_scratch(ctx, "dt.carry", sizeof(float) * (size_t)((H + strip - 1) / strip) * C * (size_t)cols_total));
            if (!carry) return PB_ERR_NOMEM;
            if (weights) {
                wts = static_cast<float *>(pb_scratch(ctx, "dt.weights", sizeof(float) * (size_t)H * (size_t)cols_total));
                if (!wts) return PB_ERR_NOMEM;
            }
        }
        if (wts) {
            if constexpr (weights_pay) {
                hipLaunchKernelGGL((dt_cols_down_kernel<T, C, DT_STRIP_W, true>), cgrid, dim3(NT), 0, ctx->stream, J, out, carry, wts, H, W, ratio, log_a, cols_total);
                hipLaunchKernelGGL((dt_cols_upw_kernel<C, DT_STRIP_W>), cgrid, dim3(NT), 0, ctx->stream, out, carry, wts, H, W, cols_total);
            }
        } else if (carry) {
            hipLaunchKernelGGL((dt_cols_down_kernel<T, C, DT_STRIP, false>), cgrid, dim3(NT), 0, ctx->stream, J, out, carry, static_cast<float *>(nullptr), H, W, ratio, log_a, cols_total);
            hipLaunchKernelGGL((dt_cols_up_kernel<T, C>), cgrid, dim3(NT), 0, ctx->stream, J, out, carry, H, W, ratio, log_a, cols_total);
        } else {
            hipLaunchKernelGGL((dt_cols_fused_kernel<T, C>), cgrid, dim3(NT), 0, ctx->stream, J, out, H, W, ratio, log_a, cols_total);
        }
        PB_LAUNCH_CHECK();
    }
    return PB_OK;
}

// out (float32, B*C*H*W) = recursive_filter(in, joint)
int pb_dt_filter_impl(pb_ctx *ctx, const void *in, const void *joint, int dtype, float *out, int B, int C, int H, int W,
                      float sigma_s, float sigma_r, int num_iterations) {
    const long HW = (long)H * W, n = (long)B * C * HW;
    ProfScope prof(ctx, PB_PROF_PREFILTER);
    const void *J = joint ? joint : in;
    const float ratio = sigma_s / sigma_r;
    const int N = num_iterations;
    if (C == 1 || C == 3) {                       // gray and colour images: no domain planes (see dt_rows_fused_kernel)
        if (dtype == PB_F32) {
            const float *i32 = static_cast<const float *>(in), *j32 = static_cast<const float *>(J);
            return C == 3 ? dt_filter_fused<float, 3>(ctx, i32, j32, out, B, H, W, ratio, N, sigma_s)
                          : dt_filter_fused<float, 1>(ctx, i32, j32, out, B, H, W, ratio, N, sigma_s);
        }
        const __half *i16 = static_cast<const __half *>(in), *j16 = static_cast<const __half *>(J);
        return C == 3 ? dt_filter_fused<__half, 3>(ctx, i16, j16, out, B, H, W, ratio, N, sigma_s)
                      : dt_filter_fused<__half, 1>(ctx, i16, j16, out, B, H, W, ratio, N, sigma_s);
    }
    float *domx = static_cast<float *>(pb_scratch(ctx, "dt.domx", sizeof(float) * B * HW));
    float *domy = static_cast<float *>(pb_scratch(ctx, "dt.domy", sizeof(float) * B * HW));
    if (!domx || !domy) return PB_ERR_NOMEM;
    dim3 dgrid(grid_for(HW, NT, 2048), B);
    if (dtype == PB_F32) {
        hipLaunchKernelGGL(dt_domain_kernel<float>, dgrid, dim3(NT), 0, ctx->stream, static_cast<const float *>(J), domx, domy, C, H, W, ratio);
        hipLaunchKernelGGL(to_float_kernel<float>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, static_cast<const float *>(in), out, n);
    } else {
        hipLaunchKernelGGL(dt_domain_kernel<__half>, dgrid, dim3(NT), 0, ctx->stream, static_cast<const __half *>(J), domx, domy, C, H, W, ratio);
        hipLaunchKernelGGL(to_float_kernel<__half>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, static_cast<const __half *>(in), out, n);
    }
    PB_LAUNCH_CHECK();
    const long rows_total = (long)B * C * H, cols_total = (long)B * C * W;
    for (int i = 0; i < N; ++i) {
        // domain_transform.py:50,53
        const double sigma_i = (double)sigma_s * std::sqrt(3.0) * std::pow(2.0, N - (i + 1)) / std::sqrt(std::pow(4.0, N) - 1.0);
        const float a = (float)std::exp(-std::sqrt(2.0) / sigma_i);
        const float log_a = std::log(a);
        hipLaunchKernelGGL(dt_rows_kernel, dim3((unsigned)((rows_total + 3) / 4)), dim3(NT), 0, ctx->stream, out, domx, C, H, W, log_a, rows_total);
        hipLaunchKernelGGL(dt_cols_kernel, dim3((unsigned)((cols_total + NT - 1) / NT)), dim3(NT), 0, ctx->stream, out, domy, C, H, W, log_a, cols_total);
        PB_LAUNCH_CHECK();
    }
    return PB_OK;
}

// (H,W,C) interleaved bytes <-> (C,H,W) planar bytes.  One thread per pixel: the interleaved side is touched
// as C consecutive bytes per lane (a wavefront covers one contiguous 64*C-byte span), the planar side as
// one byte per lane per plane.
template <bool TO_PLANAR>
__global__ void u8_layout_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int C, long HW, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
        const long inter = (b * HW + p) * C, planar = b * C * HW + p;
        for (int c = 0; c < C; ++c) {
            if (TO_PLANAR) out[planar + c * HW] = in[inter + c];
            else out[inter + c] = in[planar + c * HW];
        }
    }
}

int pb_u8_layout(pb_ctx *ctx, const unsigned char *in, unsigned char *out, int B, int C, int H, int W, int to_planar) {
    const long HW = (long)H * W, total = (long)B * HW;
    if (to_planar) hipLaunchKernelGGL(u8_layout_kernel<true>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream, in, out, C, HW, total);
    else hipLaunchKernelGGL(u8_layout_kernel<false>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream, in, out, C, HW, total);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_convert_to_float(pb_ctx *ctx, const void *in, int dtype, float *out, long n) {
    if (dtype == PB_F32) PB_HIP(hipMemcpyAsync(out, in, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
    else if (dtype == PB_U8) {
        hipLaunchKernelGGL(to_float_kernel<unsigned char>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, static_cast<const unsigned char *>(in), out, n);
        PB_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(to_float_kernel<__half>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, static_cast<const __half *>(in), out, n);
        PB_LAUNCH_CHECK();
    }
    return PB_OK;
}

int pb_convert_from_float(pb_ctx *ctx, const float *in, void *out, int dtype, long n) {
    if (dtype == PB_F32) PB_HIP(hipMemcpyAsync(out, in, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
    else if (dtype == PB_U8) {
        hipLaunchKernelGGL(from_float_kernel<unsigned char>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, in, static_cast<unsigned char *>(out), n);
        PB_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(from_float_kernel<__half>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, in, static_cast<__half *>(out), n);
        PB_LAUNCH_CHECK();
    }
    return PB_OK;
}

// ---------------------------------------------------------------------------------------------
// patch decomposition with windowed overlap-add (reference deblurring.py:269-340, fix-forward:
// the undefined `handling_saturation` branch is dropped and patches of a batch are indexed
// [n*B, (n+1)*B) instead of the reference's [n::batch_size], which is only right for B == 1)
// ---------------------------------------------------------------------------------------------
namespace {

// patches[(n*B + b), c, y, x] = img[b, c, clamp(i0 + y - pad_top), clamp(j0 + x - pad_left)],
// n = first + local patch index, (i0, j0) = (n / n_j * step_h, n % n_j * step_w): pad_with_new_size
// (replicate, deblurring.py:368-377) and the slicing (:312-313) in one pass.
template <typename T>
__global__ __launch_bounds__(NT) void extract_patches_kernel(const T *__restrict__ img, T *__restrict__ patches, int B, int C,
                                                             int H, int W, int ph, int pw, int step_h, int step_w, int n_j,
                                                             int pad_top, int pad_left, int first, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int x = (int)(i % pw);
        long r = i / pw;
        const int y = (int)(r % ph); r /= ph;
        const int c = (int)(r % C); r /= C;
        const int b = (int)(r % B);
        const int n = first + (int)(r / B);
        const int i0 = (n / n_j) * step_h, j0 = (n % n_j) * step_w;
        const int sy = min(max(i0 + y - pad_top, 0), H - 1), sx = min(max(j0 + x - pad_left, 0), W - 1);
        patches[i] = img[(((long)b * C + c) * H + sy) * W + sx];
    }
}

// out[b, c, Y, X] = clamp( sum_n restored_n * w / (sum_n w + 1e-8), 0, 1 ) over the patches that cover the
// (padded) pixel, written straight into the cropped H x W result (deblurring.py:332-340): a gather, so
// no atomics and no window_sum buffer.
template <typename T>
__global__ __launch_bounds__(NT) void overlap_add_kernel(const T *__restrict__ patches, T *__restrict__ out, int B, int C, int H,
                                                         int W, int ph, int pw, int step_h, int step_w, int n_i, int n_j,
                                                         int pad_top, int pad_left, const float *__restrict__ win_y,
                                                         const float *__restrict__ win_x, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int X = (int)(i % W);
        long r = i / W;
        const int Y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int py = Y + pad_top, px = X + pad_left;            // position in the padded image
        int ki_lo = (py - ph + step_h) / step_h; if (py - ph + 1 <= 0) ki_lo = 0;
        int kj_lo = (px - pw + step_w) / step_w; if (px - pw + 1 <= 0) kj_lo = 0;
        const int ki_hi = min(py / step_h, n_i - 1), kj_hi = min(px / step_w, n_j - 1);
        float num = 0.f, den = 0.f;
        for (int ki = ki_lo; ki <= ki_hi; ++ki) {
            const int y = py - ki * step_h;
            if (y < 0 || y >= ph) continue;
            for (int kj = kj_lo; kj <= kj_hi; ++kj) {
                const int x = px - kj * step_w;
                if (x < 0 || x >= pw) continue;
                const float w = win_y[y] * win_x[x];
                const long n = (long)ki * n_j + kj;
                num += w * pb_ld(patches + (((n * B + b) * C + c) * ph + y) * pw + x);
                den += w;
            }
        }
        pb_st(out + i, fminf(fmaxf(num / (den + 1e-8f), 0.f), 1.f));
    }
}

}  // namespace

int pb_extract_patches_impl(pb_ctx *ctx, const void *img, void *patches, int dtype, int B, int C, int H, int W, int ph, int pw,
                            int step_h, int step_w, int n_j, int pad_top, int pad_left, int first, int count) {
    const long total = (long)count * B * C * ph * pw;
    ProfScope prof(ctx, PB_PROF_OTHER);
    if (dtype == PB_F32)
        hipLaunchKernelGGL(extract_patches_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream,
                           static_cast<const float *>(img), static_cast<float *>(patches), B, C, H, W, ph, pw, step_h, step_w, n_j,
                           pad_top, pad_left, first, total);
    else
        hipLaunchKernelGGL(extract_patches_kernel<__half>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream,
                           static_cast<const __half *>(img), static_cast<__half *>(patches), B, C, H, W, ph, pw, step_h, step_w,
                           n_j, pad_top, pad_left, first, total);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_overlap_add_impl(pb_ctx *ctx, const void *patches, void *out, int dtype, int B, int C, int H, int W, int ph, int pw,
                        int step_h, int step_w, int n_i, int n_j, int pad_top, int pad_left, const float *win_y,
                        const float *win_x) {
    const long total = (long)B * C * H * W;
    ProfScope prof(ctx, PB_PROF_OTHER);
    if (dtype == PB_F32)
        hipLaunchKernelGGL(overlap_add_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream,
                           static_cast<const float *>(patches), static_cast<float *>(out), B, C, H, W, ph, pw, step_h, step_w, n_i,
                           n_j, pad_top, pad_left, win_y, win_x, total);
    else
        hipLaunchKernelGGL(overlap_add_kernel<__half>, dim3(grid_for(total)), dim3(NT), 0, ctx->stream,
                           static_cast<const __half *>(patches), static_cast<__half *>(out), B, C, H, W, ph, pw, step_h, step_w,
                           n_i, n_j, pad_top, pad_left, win_y, win_x, total);
    PB_LAUNCH_CHECK();
    return PB_OK;
}
