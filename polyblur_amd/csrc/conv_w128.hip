// Tile-spectrum body of the one-pass polynomial on 128 x 128 windows: FOUR waves per window pair.
//
// Same pass, same operands, same result as the one-pass form of conv_wfft.hip (the whole polynomial a3 K^3 + a2 K^2 + a1 K + b
// of the deconvolution as ONE circular correlation per window -- the reference's own 'fft' form, deblurring.py:139-169 -- with
// the boundary models of filters.py:14-49 applied when the window is loaded): forward 2-D DFT, product with the polynomial's
// real spectrum (khat128_body, khat.h), inverse DFT, of which the samples at least (hx, hy) from the window's edge are
// kept (overlap-save); two horizontally adjacent real windows ride one complex transform, z = A + iB.
//
// Why a second window size.  The composite filter's halo is sqrt(3) times the kernel's: 20 x 24 samples for the headline's
// first estimate (sigma 2.1 / rho 1.3 at 66 degrees).  A 64 x 64 window keeps 24 x 16 of its 4096 samples then -- three
// Horner passes with the kernel's own 12-sample halo are cheaper --, a 128 x 128 window keeps 88 x 80 of 16384: 43 % in ONE
// pass against 39 % in each of three.
//
// Who does it.  A lane still holds 64 complex values, so a 128-point line is shared by the two halves of a wave: lanes l and
// l + 32 hold its two halves, exchange them with v_permlane32_swap for ONE radix-2 step (decimation in frequency forward,
// in time backward; twiddles are compile-time constants: the step's index is the register number) and run the 64-point
// transform of conv_wave_common.h on what they then hold.  A wave therefore owns 32 lines, a workgroup of four waves the 128:
//
//   load      wave w, lane (c, h): window column 32 w + c, rows 64 h .. 64 h + 63 in the registers
//   columns   radix-2 across the wave's halves, fft64 in registers: the lower lanes hold the even, the upper the odd frequencies
//   transpose through the workgroup's LDS matrix: lane (row slot, x half) holds 64 consecutive columns of one transformed row
//   rows      radix-2, fft64, x real spectrum, inverse fft64, inverse radix-2
//   transpose back, inverse column transform, epilogue (clamp, store)
//
// The transposes move 128 KB through 66 KB of LDS (two workgroups per CU: eight waves, two per SIMD, as in conv_wfft.hip) in
// two rounds: every lane sends half its registers, the waves whose rows those are receive theirs -- 64 values per lane, 32
// into the registers just freed and 32 into a spare set, which is why the window pair's 128 registers leave room for
// this at all --, then the other half.  Four workgroup barriers per transpose.
// No MFMA, no library FFT.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_wave_common.h"

namespace {

constexpr int W_N = 128;            // window side
constexpr int W_P = 129;            // LDS row pitch in complex values (rows of 128 plus one: conflict-free both ways)
constexpr size_t kW128Lds = sizeof(float2) * 64 * W_P;

// cos / sin (2 pi m / 128), m = 0 .. 63: indexed with compile-time constants only
static __device__ const float kC128[64] = {
    1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f, 0.9807852506637573f, 0.9700312614440918f,
    0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f, 0.8819212913513184f, 0.8577286005020142f,
    0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f, 0.7071067690849304f, 0.6715589761734009f,
    0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f, 0.4713967442512512f, 0.4275550842285156f,
    0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f, 0.19509032368659973f, 0.1467304676771164f,
    0.0980171412229538f, 0.049067676067352295f, 6.123234262925839e-17f, -0.049067676067352295f, -0.0980171412229538f, -0.1467304676771164f,
    -0.19509032368659973f, -0.24298018217086792f, -0.290284663438797f, -0.3368898630142212f, -0.3826834261417389f, -0.4275550842285156f,
    -0.4713967442512512f, -0.5141027569770813f, -0.5555702447891235f, -0.5956993103027344f, -0.6343932747840881f, -0.6715589761734009f,
    -0.7071067690849304f, -0.7409511208534241f, -0.7730104327201843f, -0.803207516670227f, -0.8314695954322815f, -0.8577286005020142f,
    -0.8819212913513184f, -0.903989315032959f, -0.9238795042037964f, -0.9415440559387207f, -0.9569403529167175f, -0.9700312614440918f,
    -0.9807852506637573f, -0.9891765117645264f, -0.9951847195625305f, -0.9987954497337341f};
static __device__ const float kS128[64] = {
    0.0f, 0.049067676067352295f, 0.0980171412229538f, 0.1467304676771164f, 0.19509032368659973f, 0.24298018217086792f,
    0.290284663438797f, 0.3368898630142212f, 0.3826834261417389f, 0.4275550842285156f, 0.4713967442512512f, 0.5141027569770813f,
    0.5555702447891235f, 0.5956993103027344f, 0.6343932747840881f, 0.6715589761734009f, 0.7071067690849304f, 0.7409511208534241f,
    0.7730104327201843f, 0.803207516670227f, 0.8314695954322815f, 0.8577286005020142f, 0.8819212913513184f, 0.903989315032959f,
    0.9238795042037964f, 0.9415440559387207f, 0.9569403529167175f, 0.9700312614440918f, 0.9807852506637573f, 0.9891765117645264f,
    0.9951847195625305f, 0.9987954497337341f, 1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f,
    0.9807852506637573f, 0.9700312614440918f, 0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f,
    0.8819212913513184f, 0.8577286005020142f, 0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f,
    0.7071067690849304f, 0.6715589761734009f, 0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f,
    0.4713967442512512f, 0.4275550842285156f, 0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f,
    0.19509032368659973f, 0.1467304676771164f, 0.0980171412229538f, 0.049067676067352295f};

// One radix-2 step between the halves of a wave.  Forward (decimation in frequency): lanes l < 32 hold a[r], lanes l + 32
// hold b[r] = the sample 64 further on (r = register); afterwards the lower lanes hold a + b -- the input of the even
// frequencies' 64-point transform -- and the upper (a - b) W128^r, that of the odd ones.  sg = +1 in the lower lanes, -1 in
// the upper.
__device__ __forceinline__ void r2_twiddle_fwd(cf (&v)[64], bool upper) {
    if (upper) {
#pragma unroll
        for (int r = 1; r < 64; ++r) v[r] = cmul_s(v[r], (cf){kC128[r], -kS128[r]});
    }
}
__device__ __forceinline__ void r2_twiddle_inv(cf (&v)[64], bool upper) {
    if (upper) {
#pragma unroll
        for (int r = 1; r < 64; ++r) v[r] = cmul_conj_s(v[r], (cf){kC128[r], -kS128[r]});
    }
}
__device__ __forceinline__ void r2_exchange(cf (&v)[64], float sg) {
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        cf a = v[r], b = v[r];
        swap_halves(a, b);                                     // a: the lower lanes' value in every lane, b: the upper lanes'
        v[r] = a + b * sg;
    }
}
__device__ __forceinline__ void r2_fwd(cf (&v)[64], bool upper, float sg) { r2_exchange(v, sg); r2_twiddle_fwd(v, upper); }
__device__ __forceinline__ void r2_inv(cf (&v)[64], bool upper, float sg) { r2_twiddle_inv(v, upper); r2_exchange(v, sg); }

// Column layout -> row layout, WITH the rows' forward radix-2 step.  Before: wave w, lane (c, h) holds, in register g, the
// element (column x = 32 w + c, row slot 2 g + h).  After: wave w, lane (j, hh) holds, in register c, for row slot 32 w + j:
// hh = 0: a[c] + b[c], hh = 1: (a[c] - b[c]) W128^c, with a[c] = column c and b[c] = column 64 + c of that row -- what r2_fwd
// would make of the plain transpose, but both lanes of a pair read BOTH halves of their row from the matrix (64 more LDS
// reads per lane) instead of exchanging them through v_permlane32_swap (320 vector instructions).
// Z: the workgroup's matrix, 64 slots x W_P.  Round A: every lane sends its registers 0 .. 31 (slots 0 .. 63: the rows of waves
// 0 and 1, which receive 64 values per lane -- 32 into the registers they have just sent, 32 into `s`); round B: registers
// 32 .. 63 (waves 2 and 3 receive into all 64; waves 0 and 1 move the spare set into the registers they have just sent).
__device__ __forceinline__ void transpose_c2r_r2(cf (&v)[64], float2 *Z, int w, int lane, bool upper, float sg) {
    const int c = lane & 31, h = lane >> 5, x = 32 * w + c;
    const float2 *rd = Z + (32 * (w & 1) + c) * W_P;            // (as receiver: slot 32 (w & 1) + j, j = lane & 31)
    cf s[32];
    auto both = [&](int i) -> cf { return pbfft::to_cf(rd[i]) + pbfft::to_cf(rd[64 + i]) * sg; };
#pragma unroll
    for (int g = 0; g < 32; ++g) Z[(2 * g + h) * W_P + x] = pbfft::to_f2(v[g]);
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { v[i] = both(i); s[i] = both(32 + i); }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 32; ++g) Z[(2 * g + h) * W_P + x] = pbfft::to_f2(v[32 + g]);
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[32 + i] = s[i];
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = both(i);
    }
    __syncthreads();
    r2_twiddle_fwd(v, upper);
}

// Row layout -> column layout (the way back), WITH the rows' inverse radix-2 step.  Before: wave w, lane (j, hh), register c:
// hh = 0: e[c], hh = 1: o[c] of row slot 32 w + j (the two 64-point inverse transforms); after: wave w, lane (c, h), register
// g = element (column 32 w + c, slot 2 g + h) of e + o conj(W128^c) (columns 0 .. 63) and e - o conj(W128^c) (columns
// 64 .. 127): the upper lanes multiply by the twiddles before they send, and the receivers of columns x and x + 64 both read
// e[x] and o'[x] from the matrix and add or subtract.  Z: 64 columns x W_P slots.  Round A: registers 0 .. 31 -> columns
// 0 .. 31 and 64 .. 95: waves 0 and 2; round B: the rest, waves 1 and 3.
__device__ __forceinline__ void transpose_r2c_r2(cf (&v)[64], float2 *Z, int w, int lane, bool upper) {
    const int j = lane & 31, hh = lane >> 5, slot = 32 * w + j;
    const int c = lane & 31, h = lane >> 5;                    // (as receiver)
    const float2 *rd = Z + c * W_P + h;                         // e[x] in matrix row c, o'[x] in row 32 + c; slots 2 g + h
    const float sg = (w >> 1) ? -1.f : 1.f;                     // columns 64 .. 127 (waves 2, 3): e - o'
    cf s[32];
    auto both = [&](int g) -> cf { return pbfft::to_cf(rd[2 * g]) + pbfft::to_cf(rd[32 * W_P + 2 * g]) * sg; };
    r2_twiddle_inv(v, upper);
#pragma unroll
    for (int i = 0; i < 32; ++i) Z[(32 * hh + i) * W_P + slot] = pbfft::to_f2(v[i]);
    __syncthreads();
    if (!(w & 1)) {
#pragma unroll
        for (int g = 0; g < 32; ++g) { v[g] = both(g); s[g] = both(32 + g); }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) Z[(32 * hh + i) * W_P + slot] = pbfft::to_f2(v[32 + i]);
    __syncthreads();
    if (!(w & 1)) {
#pragma unroll
        for (int g = 0; g < 32; ++g) v[32 + g] = s[g];
    } else {
#pragma unroll
        for (int g = 0; g < 64; ++g) v[g] = both(g);
    }
    __syncthreads();
}

#if defined(PB_EXPERIMENTAL) && defined(PB_W128_TRACE)
// Lab build only (tools/build_variant.sh w128trace "-DPB_EXPERIMENTAL -DPB_W128_TRACE" conv_w128.hip): shader-clock stamps of the
// phases of wave 0 of the first workgroups, read back with pb_debug_w128_trace (tools/w128_trace.py).
constexpr int kTraceGroups = 8192, kTraceStamps = 12;
__device__ unsigned long long g_w128_trace[kTraceGroups * kTraceStamps];
#define PB_WT(i) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PB_WRT(i) do { if (tr) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PB_WT_ARG , unsigned long long *tr
#define PB_WT_PASS , tr
#else
#define PB_WT(i)
#define PB_WRT(i)
#define PB_WT_ARG
#define PB_WT_PASS
#endif

struct W128Geom { int ow, oh; };
struct W128Jobs { int pairs_x, njobs, per; float inv_pairs_x; };
__device__ __forceinline__ int div_rcp128(int n, float rcp_d) { return (int)(((float)n + 0.5f) * rcp_d); }   // exact for 0 <= n < 2^21
__device__ __forceinline__ W128Jobs jobs128_of(const W128Geom &g, int hx, int hy) {
    const int Tx = W_N - 2 * hx, Ty = W_N - 2 * hy;
    const int tiles_x = div_rcp128(g.ow + Tx - 1, __builtin_amdgcn_rcpf((float)Tx)), tiles_y = div_rcp128(g.oh + Ty - 1, __builtin_amdgcn_rcpf((float)Ty));
    W128Jobs j;
    j.pairs_x = (tiles_x + 1) >> 1;
    j.njobs = j.pairs_x * tiles_y;
    j.per = (j.njobs + 7) >> 3;
    j.inv_pairs_x = __builtin_amdgcn_rcpf((float)j.pairs_x);
    return j;
}

// One window pair, by the four waves of a workgroup.  Z: the workgroup's LDS matrix; kp: the image's spectrum,
// [wave][register][lane] (khat128_body).  hx a multiple of 4, hy even; a tile is 128 - 2 hx by 128 - 2 hy outputs.
// ZERO: the pass's boundary model is PB_ZERO (method='direct') -- a compile-time fact of the instantiation: the circular
// domain's kernel is instruction for instruction what it was before the zero boundary's loaders existed (with the model a
// run-time branch inside the loaders the rank-1 inner loop took 0.1026 ms against 0.0988 on the same box).
template <typename TIn, typename TOut, bool ZERO>
__device__ __forceinline__ void w128_pair(const ConvPass &a, int plane, int ty, int pxi, int hx, int hy, float2 *Z, const float *kp PB_WT_ARG) {
    constexpr int kBoundary = ZERO ? PB_ZERO : PB_WRAP;
    const int Tx = W_N - 2 * hx, Ty = W_N - 2 * hy;
    const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), c = lane & 31, h = lane >> 5;
    const bool upper = h != 0;
    const float sg = upper ? -1.f : 1.f;
    const OutRegion rg = out_region(a);
    const int oy0 = rg.y_lo + ty * Ty, wy0 = oy0 - hy;              // first output row / first window row, padded coordinates
    const int wxA = rg.x_lo + 2 * pxi * Tx - hx, wxB = wxA + Tx;
    const bool hasB = wxB + hx < rg.x_hi;
    const int x = 32 * w + c;                                       // this lane's window column
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    cf v[64];
    PB_WT(1);

    // ---- the window: lane = (column, half), register r = row 64 h + r ----
    {
        const brsrc rin = plane_rsrc(ipl, a.in_plane);
        const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
        const int pitchb = a.in_pitch * (int)sizeof(TIn);
        const bool x_inside = wxA >= lo && wxB + W_N <= Wp - lo && hasB;
        const bool y_inside = wy0 >= lo && wy0 + W_N <= Hp - lo;
        const bool inside = x_inside && y_inside;
        // (rows beyond the image -- the first and the last row of tiles -- go through the boundary model per lane: one
        // correction suffices for planes of at least a window's height)
        // 16-byte pieces: both windows inside the source along x, on 16-byte boundaries; else (fp32, circular domain) a
        // four-byte gather per lane through the boundary model -- either way global -> LDS without touching a register
        const bool pieces = x_inside && (y_inside || Hp >= 2 * W_N) && ((a.in_pitch | (wxA - lo)) & 3) == 0;
        if (sizeof(TIn) == 4) {                                     // (either boundary model: rows and columns outside the zero boundary's domain are out-of-range offsets, which write zeros)
            // fp32 windows inside the source on 16-byte boundaries: every wave brings ITS 32 columns of both windows global ->
            // LDS in 16-byte pieces (four rows per wave instruction: 8 pieces of window A and 8 of window B per row), through
            // its quarter of the workgroup's LDS: four chunks of 32 rows -- 16 for the lower lanes, 16 for the upper --
            // through two 8 KB buffers, two chunks requested before the first is waited for (as in conv_wfft.hip; the LDS
            // reads are issued behind the compiler's back for the same reason).  128 four-byte loads per lane become 32 wave
            // instructions of 1 KB.
            char *zw = reinterpret_cast<char *>(Z) + w * (int)(kW128Lds / 4);
            lds_char *zl = lds_ptr(zw);
            const int pc = lane & 15;
            const unsigned colb = (unsigned)(((pc < 8 ? wxA : wxB - 32) - lo + 32 * w + 4 * pc) * 4);
            const unsigned vo = (unsigned)((lane >> 4) * pitchb) + colb;
            const bool virt = a.in_kind == SRC_VIRTUAL;
            constexpr bool wrapb = !ZERO;
            auto request = [&](int k, int buf) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {     // LDS rows 4 j .. 4 j + 3 of the chunk: lane half j >> 2, its rows 16 k + 4 (j & 3) ..
                    const int p0 = wy0 + 64 * (j >> 2) + 16 * k + 4 * (j & 3);      // (padded coordinates, first of the four rows)
                    if (y_inside) {
                        dma16<0>(rin, zl + buf * 8192 + j * 1024, vo, (p0 - lo) * pitchb);
                    } else {
                        int pr = p0 + (lane >> 4);
                        if (wrapb) {                                                    // the circular domain (PB_WRAP) ...
                            pr = pr < 0 ? pr + Hp : (pr >= Hp ? pr - Hp : pr);
                            const int row = virt ? min(max(pr - a.pad, 0), a.H - 1) : pr;   // ... of the replicate-padded plane
                            dma16<0>(rin, zl + buf * 8192 + j * 1024, (unsigned)(row * pitchb) + colb, 0);
                        } else {                                                        // (PB_ZERO: zeros outside the padded domain; a branch of its own, the circular path pays nothing for it)
                            const bool ok = pr >= 0 && pr < Hp;
                            const int row = virt ? min(max(pr - a.pad, 0), a.H - 1) : pr;
                            dma16<0>(rin, zl + buf * 8192 + j * 1024, ok ? (unsigned)(row * pitchb) + colb : kNoAccess, 0);
                        }
                    }
                }
            };
            // Windows that cross the plane's left or right border (the first and the last pair of a row of tiles: 9 % of the
            // pairs at 4K, and they took 2.5 x the time of the others sample by sample): lane l gathers column l of window A
            // (l < 32) or B through the boundary model, one LDS row of 256 bytes per wave instruction, its row mapped on the
            // scalar side; the chunk then looks exactly like one that arrived in 16-byte pieces.
            const int gx = (lane < 32 || hasB ? (lane < 32 ? wxA : wxB) : wxA) + 32 * w + (lane & 31);   // (no window B: A's samples again -- finite, never stored)
            const int gix = map_axis(gx, a.W, a.in_kind, kBoundary, a.pad);
            const unsigned gcol = gix >= 0 ? (unsigned)(gix * (int)sizeof(TIn)) : kNoAccess;
            auto gather = [&](int k, int buf) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {    // LDS row i of the chunk: window row 16 k + i (i < 16) or 64 + 16 k + i - 16
                    int pr = wy0 + 16 * k + (i < 16 ? i : 48 + i);
                    if (wrapb) {
                        while (pr < 0) pr += Hp;
                        while (pr >= Hp) pr -= Hp;
                        const int row = virt ? min(max(pr - a.pad, 0), a.H - 1) : pr;
                        dma4<0>(rin, zl + buf * 8192 + i * 256, gcol, row * pitchb);
                    } else {
                        const bool ok = pr >= 0 && pr < Hp;
                        const int row = ok ? (virt ? min(max(pr - a.pad, 0), a.H - 1) : pr) : 0;
                        dma4<0>(rin, zl + buf * 8192 + i * 256, ok ? gcol : kNoAccess, row * pitchb);
                    }
                }
            };
            // Both lanes of a pair read BOTH halves' rows of the chunk (LDS rows i and 16 + i of a buffer: 256 bytes each, A then
            // B) and form a + b (lower lanes) or a - b (upper lanes) themselves: the columns' forward radix-2 step without its
            // lane exchange (64 more LDS reads per lane instead of 320 vector instructions).
            const unsigned la = lds_addr(zw) + (unsigned)(c * 4);
            auto pick = [&](int k, int buf) {
                cf tb[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const unsigned ad = la + (unsigned)(buf * 8192 + g4 * 1024), ad2 = ad + 4096u;
                    asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(v[16 * k + 4 * g4]) : "v"(ad));
                    asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(v[16 * k + 4 * g4 + 1]) : "v"(ad));
                    asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:160" : "=v"(v[16 * k + 4 * g4 + 2]) : "v"(ad));
                    asm volatile("ds_read2_b32 %0, %1 offset0:192 offset1:224" : "=v"(v[16 * k + 4 * g4 + 3]) : "v"(ad));
                    asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(tb[4 * g4]) : "v"(ad2));
                    asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(tb[4 * g4 + 1]) : "v"(ad2));
                    asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:160" : "=v"(tb[4 * g4 + 2]) : "v"(ad2));
                    asm volatile("ds_read2_b32 %0, %1 offset0:192 offset1:224" : "=v"(tb[4 * g4 + 3]) : "v"(ad2));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[16 * k]), "+v"(v[16 * k + 1]), "+v"(v[16 * k + 2]), "+v"(v[16 * k + 3]), "+v"(v[16 * k + 4]),
                             "+v"(v[16 * k + 5]), "+v"(v[16 * k + 6]), "+v"(v[16 * k + 7]), "+v"(v[16 * k + 8]), "+v"(v[16 * k + 9]), "+v"(v[16 * k + 10]),
                             "+v"(v[16 * k + 11]), "+v"(v[16 * k + 12]), "+v"(v[16 * k + 13]), "+v"(v[16 * k + 14]), "+v"(v[16 * k + 15]) :: "memory");
                asm volatile("" : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(tb[4]), "+v"(tb[5]), "+v"(tb[6]), "+v"(tb[7]), "+v"(tb[8]),
                             "+v"(tb[9]), "+v"(tb[10]), "+v"(tb[11]), "+v"(tb[12]), "+v"(tb[13]), "+v"(tb[14]), "+v"(tb[15]) :: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) v[16 * k + i] = v[16 * k + i] + tb[i] * sg;
            };
            if (pieces) {
                request(0, 0); request(1, 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                pick(0, 0);
                request(2, 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                pick(1, 1);
                request(3, 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                pick(2, 0);
                wait_vm0();
                pick(3, 1);
            } else {
                gather(0, 0); gather(1, 1);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                pick(0, 0);
                gather(2, 0);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                pick(1, 1);
                gather(3, 1);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                pick(2, 0);
                wait_vm0();
                pick(3, 1);
            }
            __syncthreads();                                        // (the transposes reuse every wave's quarter)
            r2_twiddle_fwd(v, upper);
        } else if (inside) {
            // both windows inside the source: one offset per lane, the row in the scalar offset
            const unsigned colA = (unsigned)((wy0 - lo + 64 * h) * pitchb + (wxA - lo + x) * (int)sizeof(TIn));
            const unsigned colB = colA + (unsigned)(Tx * (int)sizeof(TIn));
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const int r = 8 * (q & 7) + (q >> 3);
                v[r] = (cf){BufIO<TIn>::ld(rin, colA, r * pitchb), BufIO<TIn>::ld(rin, colB, r * pitchb)};
            }
            r2_fwd(v, upper, sg);
        } else {
            // border window: columns mapped through the boundary model once per lane, rows on the scalar side (one per half)
            const int ixa = map_axis(wxA + x, a.W, a.in_kind, kBoundary, a.pad);
            // (no window B: window A's samples again -- finite, never stored --, as the LDS-DMA loaders of fp32 planes have it: what the
            // imaginary half holds reaches the real half's ROUNDING, and an 8-bit or fp16 image must get bit for bit what its float
            // copy gets -- tests/test_gpu_parity.py::test_uint8_edge)
            const int ixb = map_axis((hasB ? wxB : wxA) + x, a.W, a.in_kind, kBoundary, a.pad);
            const unsigned colA = ixa >= 0 ? (unsigned)ixa * (unsigned)sizeof(TIn) : kNoAccess;
            const unsigned colB = ixb >= 0 ? (unsigned)ixb * (unsigned)sizeof(TIn) : kNoAccess;
            constexpr bool wrap = !ZERO;
            const int base = wrap ? __builtin_amdgcn_readfirstlane(wrap_idx(wy0, Hp)) : wy0;
            // (planes at least a window tall: one conditional step brings a row into the circular domain -- straight-line code,
            // the 128 loads in flight together; the loops of a shorter plane end a basic block per row, and every pair of loads
            // is then waited for before the next is issued)
            auto rows = [&](auto tall) {
                auto rowmap = [&](int p, bool &ok) -> int {
                    if (wrap) {
                        if (decltype(tall)::value) p -= p >= Hp ? Hp : 0;           // (base in [0, Hp), 0 <= r < 128)
                        else { while (p < 0) p += Hp; while (p >= Hp) p -= Hp; }
                    }
                    ok = wrap || (p >= 0 && p < Hp);
                    return ok ? (a.in_kind == SRC_VIRTUAL ? min(max(p - a.pad, 0), a.H - 1) : p) : 0;
                };
#pragma unroll
                for (int q = 0; q < 64; ++q) {
                    const int r = 8 * (q & 7) + (q >> 3);
                    bool ok0, ok1;
                    const int i0 = rowmap(base + r, ok0), i1 = rowmap(base + 64 + r, ok1);
                    const bool ok = upper ? ok1 : ok0;
                    const unsigned ro_ = (unsigned)((upper ? i1 : i0) * pitchb);
                    v[r] = (cf){BufIO<TIn>::ld(rin, ok && colA != kNoAccess ? colA + ro_ : kNoAccess, 0),
                                BufIO<TIn>::ld(rin, ok && colB != kNoAccess ? colB + ro_ : kNoAccess, 0)};
                }
            };
            if (Hp >= W_N) rows(std::true_type()); else rows(std::false_type());
            r2_fwd(v, upper, sg);
        }
    }
    PB_WT(2);
    fft64_fwd(v);                                                   // columns (their radix-2 step: above)
    PB_WT(3);
    transpose_c2r_r2(v, Z, w, lane, upper, sg);                      // ... and the rows' radix-2 step
    PB_WT(4);
    {
        // rows: (radix-2 across the halves -- columns x and x + 64 -- inside the transposes,) 64-point transform, x spectrum,
        // and back.  The spectrum's 64 values per lane travel in a ring of four groups of eight, as in conv_wfft.hip.
        const brsrc rk = plane_rsrc(kp, (long)W_N * W_N);
        float kh[4][8];
        auto khload = [&](int grp) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) kh[grp & 3][k2] = BufIO<float>::ld(rk, (unsigned)lane * 4u, ((w * 64 + 8 * grp + k2) * 64) * 4);
        };
        khload(0); khload(1); khload(2); khload(3);
        __builtin_amdgcn_sched_barrier(0);
        fft64_fwd_stage1(v);
        centre_stage<0>(v, kh[0]); khload(4); __builtin_amdgcn_sched_barrier(0);
        centre_stage<1>(v, kh[1]); khload(5); __builtin_amdgcn_sched_barrier(0);
        centre_stage<2>(v, kh[2]); khload(6); __builtin_amdgcn_sched_barrier(0);
        centre_stage<3>(v, kh[3]); khload(7); __builtin_amdgcn_sched_barrier(0);
        centre_stage<4>(v, kh[0]); centre_stage<5>(v, kh[1]); centre_stage<6>(v, kh[2]); centre_stage<7>(v, kh[3]);
        fft64_inv_stage1(v);
    }
    PB_WT(5);
    transpose_r2c_r2(v, Z, w, lane, upper);                         // ... with the rows' inverse radix-2 step
    PB_WT(6);
    fft64_inv_stage2(v);                                            // columns
    fft64_inv_stage1(v);
    r2_twiddle_inv(v, upper);                                       // (the halves' exchange: in the epilogue)
    PB_WT(7);

    // ---- epilogue: lane = (column, half); after the exchange register r = window row 64 h + r; the polynomial carries its
    // own b x ----
    {
        const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
        const int opitchb = a.out_pitch * (int)sizeof(TOut);
        const brsrc ro = plane_rsrc(opl, a.out_plane);
        const float clo = a.clamp01 ? 0.f : -INFINITY, chi = a.clamp01 ? 1.f : INFINITY;
        const int rmax = min(W_N - hy, rg.y_hi - wy0);              // window rows hy .. rmax - 1 are the tile's rows inside the region
        if (((a.out_pitch | (wxA - oo) | (rg.x_hi - wxA)) & 3) == 0) {
            // 16-byte boundaries (tiles, plane rows and the region's end): every wave sends ITS 32 columns of both windows through its
            // quarter of the LDS (written by columns, read back as 16-byte row pieces: 8 of window A, 8 of window B per row,
            // four rows per wave instruction), registers 0 .. 31 of both lane halves first, then 32 .. 63 -- 32 stores of 1 KB
            // per wave instead of 128 of 256 bytes.  The columns' inverse radix-2 step rides along: the lanes write e (lower
            // half) and o conj(W) (upper half) and the row pieces read back are e + o' (window rows 0 .. 63) and e - o' (64 ..).
            __syncthreads();                                        // (the last transpose's reads of the other waves' quarters)
            float *zw = reinterpret_cast<float *>(reinterpret_cast<char *>(Z) + w * (int)(kW128Lds / 4));
            float *zt = zw + (32 * h) * 64 + c;
            const int pc = lane & 15, lr = lane >> 4;
            const int xcol = 32 * w + 4 * (pc & 7);                 // first window column of this lane's piece
            // (the last pair of a row of tiles: a narrower last tile, or no window B at all -- whole pieces fall away, the
            // region ends on a piece boundary)
            const bool colok = xcol >= hx && xcol < W_N - hx && (pc < 8 || hasB) && (pc < 8 ? wxA : wxB) + xcol + 4 <= rg.x_hi;
            const int cb = ((pc < 8 ? wxA : wxB) + xcol - oo) * (int)sizeof(TOut), rb = (wy0 - oo + lr) * opitchb;   // (rb < 0 above the plane: only for rows outside the tile)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < 32; ++i) { zt[i * 64] = v[32 * p + i].x; zt[i * 64 + 32] = v[32 * p + i].y; }
                wave_lds_fence();
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int row = 32 * p + 4 * t;                                 // window row of the instruction's first LDS row (lower half)
                    const f4v qe = *reinterpret_cast<const f4v *>(zw + (4 * t + lr) * 64 + 4 * pc);
                    const f4v qo = *reinterpret_cast<const f4v *>(zw + (32 + 4 * t + lr) * 64 + 4 * pc);
                    f4v o0, o1;
                    o0.x = __builtin_amdgcn_fmed3f(qe.x + qo.x, clo, chi); o0.y = __builtin_amdgcn_fmed3f(qe.y + qo.y, clo, chi);
                    o0.z = __builtin_amdgcn_fmed3f(qe.z + qo.z, clo, chi); o0.w = __builtin_amdgcn_fmed3f(qe.w + qo.w, clo, chi);
                    o1.x = __builtin_amdgcn_fmed3f(qe.x - qo.x, clo, chi); o1.y = __builtin_amdgcn_fmed3f(qe.y - qo.y, clo, chi);
                    o1.z = __builtin_amdgcn_fmed3f(qe.z - qo.z, clo, chi); o1.w = __builtin_amdgcn_fmed3f(qe.w - qo.w, clo, chi);
                    const bool ok0 = colok && row + lr >= hy && row + lr < rmax, ok1 = colok && 64 + row + lr >= hy && 64 + row + lr < rmax;
                    Piece4<TOut>::st(ro, ok0 ? (unsigned)(rb + row * opitchb + cb) : kNoAccess, 0, o0);
                    Piece4<TOut>::st(ro, ok1 ? (unsigned)(rb + (64 + row) * opitchb + cb) : kNoAccess, 0, o1);
                }
                wave_lds_fence();
            }
        } else {
            const bool colin = x >= hx && x < W_N - hx;
            const int pxA = wxA + x, pxB = wxB + x;
            const bool okA = colin && pxA < rg.x_hi, okB = colin && hasB && pxB < rg.x_hi;
            const int row0 = 64 * h;                                // this lane's first window row
            // (the lane's offset is that of its FIRST row of the tile -- the window's first rows lie above the output plane for
            // the first tiles -- and a row's offset relative to it is added per row)
            const int rlo = max(hy - row0, 0), rhi = min(rmax - row0, 64);
            const unsigned baseA = okA && rlo < rhi ? (unsigned)((wy0 + row0 + rlo - oo) * opitchb + (pxA - oo) * (int)sizeof(TOut)) : kNoAccess;
            const unsigned baseB = okB && rlo < rhi ? (unsigned)((wy0 + row0 + rlo - oo) * opitchb + (pxB - oo) * (int)sizeof(TOut)) : kNoAccess;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const bool rok = r >= rlo && r < rhi;
                cf ea = v[r], eb = v[r];
                swap_halves(ea, eb);                               // (the columns' inverse radix-2 step, register by register)
                const cf er = ea + eb * sg;
                const float ra = __builtin_amdgcn_fmed3f(er.x, clo, chi), rb = __builtin_amdgcn_fmed3f(er.y, clo, chi);
                BufIO<TOut>::st(ro, rok ? baseA + (unsigned)((r - rlo) * opitchb) : kNoAccess, 0, ra);
                BufIO<TOut>::st(ro, rok ? baseB + (unsigned)((r - rlo) * opitchb) : kNoAccess, 0, rb);
            }
        }
    }
    PB_WT(8);
    PB_WRT(10);
}

// One workgroup of four waves per window pair; the GRID is the job list, as in conv_wfft.hip: workgroup b belongs to list
// b % 8 (the XCD it is observed to run on) at position b / 8, every list owns the same contiguous eighth of every plane's
// pairs, images in order; the jobs are the window pairs of the images whose record says "one pass on 128 x 128 windows"
// (pb_fft_sel.poly == 2), each with its own halos.
template <typename TIn, typename TOut, bool ZERO>
__global__ __launch_bounds__(256, 2) void conv_w128_kernel(const ConvPass a, const W128Geom g) {
    extern __shared__ __attribute__((aligned(16))) float2 Zw[];
    const int lane = threadIdx.x & 63;
#if defined(PB_EXPERIMENTAL) && defined(PB_W128_TRACE)
    unsigned long long *tr = (blockIdx.x < kTraceGroups && threadIdx.x < 64) ? g_w128_trace + (long)blockIdx.x * kTraceStamps : nullptr;
    PB_WT(0);
    PB_WRT(9);
#endif
    const int C = a.C, B = a.P / C;
    const int q = (int)(blockIdx.x & 7u);
    int rem = (int)(blockIdx.x >> 3);
    int img = 0, hx = 0, hy = 0;
    if (B == 1) {
        const PB_CONSTANT pb_fft_sel *s0 = as_constant(a.fsel);
        if (!s0->use_fft || s0->poly != 2) return;
        hx = s0->hx; hy = s0->hy;
    } else {
        auto share_of = [&](int i) -> int {
            if (i >= B) return 0;
            const pb_fft_sel s = a.fsel[i];
            if (!s.use_fft || s.poly != 2) return 0;
            return jobs128_of(g, s.hx, s.hy).per * C;
        };
        bool work = false;
        int base = 0;
        for (int c0 = 0; c0 < B; c0 += 64) {
            const int n = share_of(c0 + lane), incl = wave_scan(n, lane);
            const unsigned long long m = __ballot(base + incl > rem);
            if (m) {
                const int l = __builtin_ctzll(m);
                img = c0 + l;
                rem -= base + (__builtin_amdgcn_readlane(incl, l) - __builtin_amdgcn_readlane(n, l));
                work = true;
                break;
            }
            base += __builtin_amdgcn_readlane(incl, 63);
        }
        if (!work) return;
        img = __builtin_amdgcn_readfirstlane(img); rem = __builtin_amdgcn_readfirstlane(rem);
        hx = as_constant(a.fsel + img)->hx; hy = as_constant(a.fsel + img)->hy;
    }
    const W128Jobs j = jobs128_of(g, hx, hy);
    const int pl = __builtin_amdgcn_readfirstlane(div_rcp128(rem, __builtin_amdgcn_rcpf((float)j.per)));
    if (pl >= C) return;
    const int pair = q * j.per + (rem - pl * j.per);
    if (pair >= j.njobs) return;
    const int ty = __builtin_amdgcn_readfirstlane(div_rcp128(pair, j.inv_pairs_x)), pxi = pair - ty * j.pairs_x;
    w128_pair<TIn, TOut, ZERO>(a, img * C + pl, ty, pxi, hx, hy, Zw, a.khat + (long)img * PB_KHAT_STRIDE PB_WT_PASS);
}

template <typename TIn, typename TOut>
int launch_w128_typed(pb_ctx *ctx, const ConvPass &p, const W128Geom &g, long groups) {
    if (p.boundary == PB_ZERO)
        hipLaunchKernelGGL((conv_w128_kernel<TIn, TOut, true>), dim3((unsigned)groups), dim3(256), kW128Lds, ctx->stream, p, g);
    else
        hipLaunchKernelGGL((conv_w128_kernel<TIn, TOut, false>), dim3((unsigned)groups), dim3(256), kW128Lds, ctx->stream, p, g);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

#if defined(PB_EXPERIMENTAL) && defined(PB_W128_TRACE)
extern "C" int pb_debug_w128_trace(unsigned long long *host) {
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w128_trace), sizeof(g_w128_trace));
    static unsigned long long zeros[kTraceGroups * kTraceStamps];
    if (!rc) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w128_trace), zeros, sizeof(zeros));
    return rc;
}
#endif

bool pb_conv_w128_types(int in_dtype, int out_dtype) { return in_dtype >= 0 && in_dtype <= 2 && out_dtype >= 0 && out_dtype <= 2; }

// Whether the job list of a launch sized for the smallest tiles (records the host has not read) stays within the grid
// (pb_poly_spec_mode asks before it admits 128 x 128 windows: an oversized batch keeps the other forms instead of failing).
bool pb_conv_w128_feasible(const ConvPass &p) {
    const int oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad, ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    const int t = PB_POLY128_MIN_T;
    const long nj = (long)(((ow + t - 1) / t + 1) / 2) * ((oh + t - 1) / t);
    const long groups = 8L * ((nj + 7) / 8) * p.P;
    return groups > 0 && groups <= (1L << 23);
}

// The one-pass polynomial of the images whose record selects 128 x 128 windows (pb_fft_sel.poly == 2): from the first step's
// input to the last step's output (p: the composite pass -- scale 1, no x operand, the last step's clamp).
int pb_launch_conv_w128(pb_ctx *ctx, const ConvPass &p) {
    W128Geom g;
    g.oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    g.ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    // the job list: exactly the images' where the host has their records, else the largest their halos may ask for
    // (tiles of at least PB_POLY128_MIN_T samples per side)
    long groups = 0;
    const int B = p.P / p.C;
    auto per_of = [&](int hx, int hy) -> long {
        const int tx = W_N - 2 * hx, ty = W_N - 2 * hy;
        const long nj = (long)(((g.ow + tx - 1) / tx + 1) / 2) * ((g.oh + ty - 1) / ty);
        return (nj + 7) / 8;
    };
    if (ctx->known_sel) {
        long per_sum = 0;
        for (int b = 0; b < B && b < (int)ctx->known_sel->size(); ++b) {
            const pb_fft_sel &e = (*ctx->known_sel)[(size_t)b];
            if (e.use_fft && e.poly == 2) per_sum += per_of(e.hx, e.hy);
        }
        if (!per_sum) return PB_OK;
        groups = 8L * per_sum * p.C;
    } else {
        const int hmax = (W_N - PB_POLY128_MIN_T) / 2;
        groups = 8L * per_of(hmax, hmax) * p.P;
    }
    if (groups <= 0 || groups > (1L << 23)) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: too many 128 x 128 windows");
    ProfScope prof(ctx, PB_PROF_CONV_FFT);
    switch (p.in_dtype * 3 + p.out_dtype) {
        case 0: return launch_w128_typed<float, float>(ctx, p, g, groups);
        case 1: return launch_w128_typed<float, __half>(ctx, p, g, groups);
        case 3: return launch_w128_typed<__half, float>(ctx, p, g, groups);
        case 4: return launch_w128_typed<__half, __half>(ctx, p, g, groups);
        case 2: return launch_w128_typed<float, unsigned char>(ctx, p, g, groups);
        case 5: return launch_w128_typed<__half, unsigned char>(ctx, p, g, groups);
        case 6: return launch_w128_typed<unsigned char, float>(ctx, p, g, groups);
        case 7: return launch_w128_typed<unsigned char, __half>(ctx, p, g, groups);
        case 8: return launch_w128_typed<unsigned char, unsigned char>(ctx, p, g, groups);
        default: return PB_ERR_UNSUPPORTED;
    }
}
