// Why is gray+min/max at 1.9 TB/s?  Same access pattern in isolation, varying grid and atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int ATOM> __global__ __launch_bounds__(256) void gray3(const float *__restrict__ in, float *__restrict__ g, unsigned *mm, long HW, int nb) {
    const long n4 = HW >> 2;
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)nb * 256) {
        const float4 a = *(const float4 *)(in + 4 * i), b = *(const float4 *)(in + HW + 4 * i), c = *(const float4 *)(in + 2 * HW + 4 * i);
        float4 r; r.x = (a.x + b.x + c.x) / 3.f; r.y = (a.y + b.y + c.y) / 3.f; r.z = (a.z + b.z + c.z) / 3.f; r.w = (a.w + b.w + c.w) / 3.f;
        *(float4 *)(g + 4 * i) = r;
        lo = fminf(lo, fminf(fminf(r.x, r.y), fminf(r.z, r.w))); hi = fmaxf(hi, fmaxf(fmaxf(r.x, r.y), fmaxf(r.z, r.w)));
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (ATOM == 1 && (threadIdx.x & 63) == 0) { atomicMin(mm, __float_as_uint(lo)); atomicMax(mm + 1, __float_as_uint(hi)); }
    if (ATOM == 2 && (threadIdx.x & 63) == 0) { mm[2 + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = __float_as_uint(lo); }
}
int main() {
    const long HW = 2160L * 3840;
    float *in, *g; unsigned *mm;
    hipMalloc(&in, HW * 12); hipMalloc(&g, HW * 4); hipMalloc(&mm, 1 << 20);
    hipMemset(in, 0x3c, HW * 12);
    for (int nb : {256, 512, 1024, 2025, 4096, 8100}) {
        for (int atom = 0; atom < 3; ++atom) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto run = [&] { if (atom == 0) hipLaunchKernelGGL(gray3<0>, dim3(nb), dim3(256), 0, 0, in, g, mm, HW, nb);
                             if (atom == 1) hipLaunchKernelGGL(gray3<1>, dim3(nb), dim3(256), 0, 0, in, g, mm, HW, nb);
                             if (atom == 2) hipLaunchKernelGGL(gray3<2>, dim3(nb), dim3(256), 0, 0, in, g, mm, HW, nb); };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 20; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("blocks %5d atomics=%d : %.1f us  %.0f GB/s\n", nb, atom, ms * 1e3, HW * 16.0 / ms / 1e6);
        }
    }
    return 0;
}
