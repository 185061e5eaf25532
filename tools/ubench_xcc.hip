// Which XCD does a workgroup run on?  s_getreg HW_REG_XCC_ID against blockIdx % 8 (tools; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);
}
int main() {
    const int n = 4096;
    int *d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    int h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0}, agree = 0;
    for (int i = 0; i < n; ++i) { hist[h[i] & 15]++; agree += (h[i] & 7) == (i & 7); }
    printf("first 16:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\nhist:");
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\nagree with blockIdx%%8: %d of %d\n", agree, n);
    return 0;
}
