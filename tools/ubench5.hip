// What a kernel boundary costs on MI355X: empty kernels of the estimation's launch shapes, alone and behind a kernel that
// leaves 66 MB of dirty lines in the L2s.   hipcc --offload-arch=gfx950 -O3 tools/ubench5.hip -o tools/ubench5 && tools/ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void empty_big(int *p, int n) { extern __shared__ char s[]; if (n < 0) p[threadIdx.x] = s[threadIdx.x]; }
__global__ __launch_bounds__(256) void empty_small(int *p, int n) { extern __shared__ char s[]; if (n < 0) p[threadIdx.x] = s[threadIdx.x]; }
__global__ __launch_bounds__(256) void dirty(float4 *p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void reader(const float4 *p, long n, float *out) {
    float a = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float4 v = p[i]; a += v.x + v.w; }
    if (a == 123.456f) out[0] = a;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int *d; float4 *buf; float *o;
    const long n = 66L * 1024 * 1024 / 16;
    CK(hipMalloc(&d, 4096)); CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&o, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(empty_big), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto fn, int reps) {
        for (int i = 0; i < 5; ++i) fn();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < reps; ++i) fn();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-70s %.2f us per iteration\n", name, 1e3 * ms / reps);
    };
    time("empty 240 x 1024 threads, 155 KB LDS", [&] { hipLaunchKernelGGL(empty_big, dim3(240), dim3(1024), 155 * 1024, 0, d, 0); }, 200);
    time("empty 1080 x 256 threads, 31.7 KB LDS", [&] { hipLaunchKernelGGL(empty_small, dim3(1080), dim3(256), 31744, 0, d, 0); }, 200);
    time("empty 64 x 256 threads", [&] { hipLaunchKernelGGL(empty_small, dim3(64), dim3(256), 0, 0, d, 0); }, 200);
    time("dirty 66 MB", [&] { hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, buf, n); }, 100);
    time("dirty 66 MB + empty big", [&] { hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(empty_big, dim3(240), dim3(1024), 155 * 1024, 0, d, 0); }, 100);
    time("dirty 66 MB + read it back (other kernel)", [&] { hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, buf, n, o); }, 100);
    time("read 66 MB", [&] { hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, buf, n, o); }, 100);
    return 0;
}
