// What does the 2-D tile access pattern of the stencil pass cost against a linear stream?
// Each workgroup loads a (TH+2R) x (TW+2R) window of `in` into LDS with 16-byte loads, loads the TW x TH centre of
// `x`, and stores out = in + x for the centre: the memory behaviour of one rank-1 Horner step with the arithmetic
// removed.  Planes are 3 x (2160+24) x (3840+24) fp32 like a 4K pass.  hipcc --offload-arch=gfx950 -O3 tools/ubench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TW, int TH, int R, int NT>
__global__ __launch_bounds__(NT) void tile_copy(const float *__restrict__ in, const float *__restrict__ x, float *__restrict__ out,
                                                int H, int W, int pitch, int tiles_x, int tiles_y, int total, int xcd_order) {
    extern __shared__ float s[];
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R, C4 = LW / 4;
    int tile = blockIdx.x;
    if (xcd_order) { const int chunk = gridDim.x >> 3; tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3); }
    if (tile >= total) return;
    const int plane = tile / (tiles_x * tiles_y);
    const int local = tile - plane * tiles_x * tiles_y;
    const int ty = local / tiles_x, tx = local - ty * tiles_x;
    const int y0 = R + ty * TH, x0 = R + tx * TW;              // centre origin; window starts R before
    const float *ip = in + (long)plane * H * pitch;
    const float *xp = x + (long)plane * H * pitch;
    float *op = out + (long)plane * H * pitch;
    for (int e = threadIdx.x; e < LH * C4; e += NT) {
        const int r = e / C4, c = e - r * C4;
        const int yy = min(y0 - R + r, H - 1), xx = min(x0 - R + 4 * c, W - 4);
        *reinterpret_cast<float4 *>(s + r * LW + 4 * c) = *reinterpret_cast<const float4 *>(ip + (long)yy * pitch + xx);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TH * (TW / 4); e += NT) {
        const int r = e / (TW / 4), c = e - r * (TW / 4);
        const int yy = y0 + r, xx = x0 + 4 * c;
        if (yy < H && xx + 3 < W) {
            const float4 a = *reinterpret_cast<const float4 *>(s + (r + R) * LW + R + 4 * c);
            const float4 b = *reinterpret_cast<const float4 *>(xp + (long)yy * pitch + xx);
            *reinterpret_cast<float4 *>(op + (long)yy * pitch + xx) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
    }
}

template <typename F> float time_ms(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

template <int TW, int TH, int R, int NT> int run(const float *a, const float *b, float *c, int P, int H, int W, int pitch) {
    const int tiles_x = (W - 2 * R + TW - 1) / TW, tiles_y = (H - 2 * R + TH - 1) / TH;
    const int total = tiles_x * tiles_y * P;
    const int grid = (total + 7) / 8 * 8;
    const size_t lds = sizeof(float) * (TW + 2 * R) * (TH + 2 * R);
    for (int xo = 0; xo < 2; ++xo) {
        float ms = time_ms([&] { hipLaunchKernelGGL((tile_copy<TW, TH, R, NT>), dim3(grid), dim3(NT), lds, 0, a, b, c, H, W, pitch, tiles_x, tiles_y, total, xo); }, 20);
        const double alg = 3.0 * 4 * P * (double)(H - 2 * R) * (W - 2 * R);
        printf("tile %3dx%-3d R=%2d NT=%4d lds=%6zu xcd=%d : %.1f us  -> %.0f GB/s algorithmic (3 words/sample)\n", TW, TH, R, NT, lds, xo, ms * 1e3, alg / ms / 1e6);
    }
    return 0;
}

int main() {
    const int P = 3, H = 2160 + 24, W = 3840 + 24, pitch = W;
    const long n = (long)P * H * pitch;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    run<64, 64, 12, 256>(a, b, c, P, H, W, pitch);
    run<64, 64, 0, 256>(a, b, c, P, H, W, pitch);
    run<128, 64, 12, 512>(a, b, c, P, H, W, pitch);
    run<128, 64, 0, 512>(a, b, c, P, H, W, pitch);
    run<128, 48, 12, 384>(a, b, c, P, H, W, pitch);
    run<128, 32, 12, 512>(a, b, c, P, H, W, pitch);
    run<96, 64, 12, 384>(a, b, c, P, H, W, pitch);
    run<192, 64, 12, 768>(a, b, c, P, H, W, pitch);
    run<256, 64, 12, 1024>(a, b, c, P, H, W, pitch);
    run<128, 96, 12, 768>(a, b, c, P, H, W, pitch);
    run<256, 32, 12, 512>(a, b, c, P, H, W, pitch);
    run<256, 32, 12, 1024>(a, b, c, P, H, W, pitch);
    run<64, 64, 12, 512>(a, b, c, P, H, W, pitch);
    run<64, 32, 12, 256>(a, b, c, P, H, W, pitch);
    return 0;
}
