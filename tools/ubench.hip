// Micro-benchmarks that size the stencil pass against the machine (run on the GPU box):
//   1. fp32 FMA issue rate: v_fma_f32 vs v_pk_fma_f32 (is packed math worth forcing on CDNA4?)
//   2. streaming "Horner step" ceiling: out = a*in + b*x over 3 x 100 MB (2 reads + 1 write),
//      as a function of workgroups per CU and 16-byte loads in flight per lane.
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int ILP> __global__ void fma_scalar(float *out, float a, float b, int iters) {
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v[i]) : "s"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < ILP; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> __global__ void fma_packed(float *out, float a, float b, int iters) {
    float2v v[ILP];
    float2v aa = {a, a * 1.0001f}, bb = {b, b};
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = (float2v){threadIdx.x * 1e-3f + i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(aa), "v"(bb));
    }
    float s = 0; for (int i = 0; i < ILP; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int UNROLL> __global__ void horner_stream(const float4 *__restrict__ in, const float4 *__restrict__ x,
                                                     float4 *__restrict__ out, long n4, float a, float b) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 u[UNROLL], w[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { u[k] = in[i + k * stride]; w[k] = x[i + k * stride]; }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            float4 r; r.x = a * u[k].x + b * w[k].x; r.y = a * u[k].y + b * w[k].y; r.z = a * u[k].z + b * w[k].z; r.w = a * u[k].w + b * w[k].w;
            out[i + k * stride] = r;
        }
    }
    for (; i < n4; i += stride) {
        float4 u = in[i], w = x[i], r; r.x = a * u.x + b * w.x; r.y = a * u.y + b * w.y; r.z = a * u.z + b * w.z; r.w = a * u.w + b * w.w; out[i] = r;
    }
}

template <typename F> float time_ms(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
    float *out; CK(hipMalloc(&out, 256 * 2048 * 4 * 8));
    const int iters = 4096;
    for (int wpb : {4}) for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc, threads = 64 * wpb;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_scalar<16>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); }, 5);
        double fl = 2.0 * 16 * iters * (double)blocks * threads;
        float ms2 = time_ms([&] { hipLaunchKernelGGL(fma_packed<16>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); }, 5);
        printf("fma  waves/SIMD=%d : v_fma_f32 %.1f TFLOP/s   v_pk_fma_f32 %.1f TFLOP/s\n", bpc, fl / ms / 1e9, 2 * fl / ms2 / 1e9);
    }
    const long n = 3L * 2160 * 3840;     // one 4K fp32 image = 99.5 MB
    float *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    std::vector<float> h(n); for (long i = 0; i < n; ++i) h[i] = (float)(i % 977) * 1e-3f;
    CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice));
    const long n4 = n / 4;
    for (int bpc : {2, 4, 8, 16}) {
#define RUN(U) { float ms = time_ms([&] { hipLaunchKernelGGL(horner_stream<U>, dim3(256 * bpc), dim3(256), 0, 0, (const float4 *)a, (const float4 *)b, (float4 *)c, n4, 0.5f, 0.25f); }, 20); \
        printf("stream 2r+1w  blocks/CU=%2d unroll=%d : %.4f ms  %.0f GB/s\n", bpc, U, ms, 3.0 * n * 4 / ms / 1e6); }
        RUN(1) RUN(2) RUN(4)
    }
    // the three-step chain in/x/t1/t2/y (400 MB working set > 256 MB Infinity Cache)
    {
        float ms = time_ms([&] {
            hipLaunchKernelGGL(horner_stream<2>, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)a, (const float4 *)a, (float4 *)b, n4, 0.5f, 0.25f);
            hipLaunchKernelGGL(horner_stream<2>, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)b, (const float4 *)a, (float4 *)c, n4, 0.5f, 0.25f);
            hipLaunchKernelGGL(horner_stream<2>, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)c, (const float4 *)a, (float4 *)d, n4, 0.5f, 0.25f); }, 20);
        printf("3-step chain (x,t1,t2,y = 4 x 99.5 MB): %.4f ms per polynomial, algorithmic 8s/sample -> %.0f GB/s\n", ms, 8.0 * n * 4 / ms / 1e6);
    }
    return 0;
}
