// fp32 VALU issue-rate probe: per-wave issue interval vs waves per SIMD, instruction form and ILP.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int ILP, int FORM> __global__ void k(float *out, float a, float b, int iters) {
    float v[ILP]; float2v p[ILP / 2];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < ILP / 2; ++i) p[i] = (float2v){threadIdx.x * 1e-3f + i, 1.f};
    float2v aa = {a, a}; float bv = b + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (FORM == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "s"(a), "v"(bv));      // VOP3, sgpr tap
            if (FORM == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(bv));         // VOP2, sgpr tap
            if (FORM == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(bv), "v"(bv));        // VOP2, vgprs
        }
        if (FORM >= 3) {
#pragma unroll
            for (int i = 0; i < ILP / 2; ++i) {
                if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(aa), "v"(p[(i + 1) % (ILP / 2)]));
                if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(aa), "v"(aa));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += v[i];
    for (int i = 0; i < ILP / 2; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP, int FORM> void run(float *out, const char *name) {
    const int iters = 2048;
    printf("%-28s ILP=%2d :", name, ILP);
    for (int w : {1, 2, 3, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<ILP, FORM>), dim3(256 * w), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<ILP, FORM>), dim3(256 * w), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double fmas = (double)ILP * iters * 256.0 * w * 256;   // lane-FMAs (pk forms: ILP/2 instr x 2)
        printf("  w%d %6.1fTF", w, 2 * fmas / ms / 1e9);
    }
    printf("\n");
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<16, 0>(out, "v_fma_f32 vop3 sgpr"); run<16, 1>(out, "v_fmac_f32 vop2 sgpr"); run<16, 2>(out, "v_fmac_f32 vop2 vgpr");
    run<16, 3>(out, "v_pk_fma sgpr opsel"); run<16, 4>(out, "v_pk_fma vgpr");
    run<32, 1>(out, "v_fmac_f32 vop2 sgpr"); run<32, 3>(out, "v_pk_fma sgpr opsel"); run<64, 3>(out, "v_pk_fma sgpr opsel");
    return 0;
}
