// Tile-spectrum body of the reblurring pass, one WAVE per window pair.
//
// Same pass, same operands, same result as conv_fft.hip (one Horner step  t <- K*t + coef*x  of the polynomial
// deconvolution, reference deblurring.py:122-138 / :141-169, or one edgetaper blend, edgetaper.py:30-32, with the
// boundary models of filters.py:14-49): the exact 2-D stencil of a 64 x 64 window is evaluated as a circular correlation
// -- forward 2-D DFT, product with the kernel's real 64 x 64 spectrum (khat_kernel, conv_fft.hip), inverse DFT, of which
// the samples at least R from the window's edge are kept (overlap-save); two horizontally adjacent real windows ride
// one complex transform, z = A + iB.
//
// What differs is who does it.  conv_fft.hip spreads a window pair over 512 threads: eight 8-point butterfly stages,
// six LDS round trips and seven workgroup barriers per pair, every stage exposing an LDS or barrier latency.  Here a
// window pair belongs to ONE wave and a whole 64-point line to one lane (64 complex values = 128 registers):
//
//   load      lane = window column x, register y = window row: 2 x 64 row-segment loads of 256 bytes per wave
//   columns   fft64 in registers (two radix-8 stages, compile-time twiddles in scalar registers), no LDS, no barrier
//   transpose through the wave's private LDS tile: lane = transformed row, register = column
//   rows      fft64, x real spectrum (64 values per lane, [x position][y position] layout: coalesced), inverse fft64
//   transpose back
//   columns   inverse fft64; lane = window column again: the epilogue (scale, + coef * x, clamp) and the stores walk
//             rows with 256-byte segments per wave instruction
//
// Two LDS round trips per pair instead of six, no workgroup barrier at all (a wave's DS operations execute in order;
// only the compiler has to be told, wave_lds_fence), rows wave-uniform -- every row offset, boundary mapping and
// validity test is scalar work -- and columns one map per lane and window.  Border windows, the taper epilogue and
// ragged ends therefore run the same code as interior ones with different offsets (an out-of-range buffer offset
// loads 0 / drops the store).
//
// The transposes keep only HALF a window pair in LDS at a time (16.6 KB per wave, eight waves per CU): a 64 x 64
// transpose is the swap of the two off-diagonal 32 x 32 quadrants -- v_permlane32_swap between the wave's halves --
// followed by a transpose inside each quadrant, and the quadrant pairs go through the same 32 x 64 LDS tile one after
// the other.
//
// Work items: window pairs of the images whose record selects this body, every image with its own tile size (prefix sum
// over the batch's pb_fft_sel records inside the kernel: no idle slots for images with larger tiles, no host read-back);
// workgroup b runs on XCD b % 8 (observed, used for speed only) and every XCD takes the same contiguous eighth of every
// plane's pairs, so neighbouring windows share their halos in that XCD's L2.
// No MFMA, no library FFT.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_wave_common.h"

namespace {

// Output extent of a pass; the window counts follow from each image's halos on the device (the wave form's halos are per
// axis -- hx a multiple of 4, hy even -- and the host never learns them).
struct WGeom { int ow, oh; };
struct WJobs { int pairs_x, njobs, per; float inv_pairs_x; };      // of one image: window pairs per row, per plane, per plane and XCD
// n / d for 0 <= n < 2^21 with the hardware's reciprocal (1 ulp): exact -- (n + 1/2) / d is at least 1 / (2 d) away from an integer
__device__ __forceinline__ int div_rcp(int n, float rcp_d) { return (int)(((float)n + 0.5f) * rcp_d); }
__device__ __forceinline__ WJobs jobs_of(const WGeom &g, int hx, int hy) {
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    const int tiles_x = div_rcp(g.ow + Tx - 1, __builtin_amdgcn_rcpf((float)Tx)), tiles_y = div_rcp(g.oh + Ty - 1, __builtin_amdgcn_rcpf((float)Ty));
    WJobs j;
    j.pairs_x = (tiles_x + 1) >> 1;
    j.njobs = j.pairs_x * tiles_y;
    j.per = (j.njobs + 7) >> 3;
    j.inv_pairs_x = __builtin_amdgcn_rcpf((float)j.pairs_x);
    return j;
}

#ifdef PB_WF_TRACE
// Debug build only (python -m polyblur_amd.build --experimental with PB_EXTRA_FLAGS=-DPB_WF_TRACE): shader-clock stamps of the
// first waves' phases, read back with pb_debug_wf_trace (tools/wf_trace.py).
constexpr int kTraceWaves = 8192, kTraceStamps = 14;
__device__ unsigned long long g_wf_trace[kTraceWaves * kTraceStamps];
#define PB_T(i) do { if (tr) { __builtin_amdgcn_sched_barrier(0); tr[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PB_TWAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PB_TRT(i) do { if (tr) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PB_T(i)
#define PB_TWAIT()
#define PB_TRT(i)
#endif

// One window pair.  zb: the wave's LDS region (kWfLdsWave bytes); kp: the image's spectrum, [x position][y position].
// hx, hy: the window halo along x (a multiple of 4: windows stay on 16-byte boundaries) and along y (even); a tile is
// Tx = 64 - 2 hx by Ty = 64 - 2 hy outputs.
//
// Rows are ROTATED in the registers: register r holds window row (r + hy) mod 64 -- the tile's rows sit in registers
// 0 .. Ty - 1, the hy halo rows above it in registers 64 - hy .. 63.  A circular correlation commutes with a circular
// shift of its input, so the transforms do not notice, and the epilogue walks registers 0 .. Ty - 1 whatever hy is: the row
// halo is a run-time value (every row offset is scalar work); only register numbers have to be compile-time constants.
// Columns are lanes: their halo is a per-lane predicate.
// MODE 1: interior pairs on 16-byte boundaries (pair_is_fast); MODE 2: the same structure for the border pairs of the
// circular domain (pair_is_gen: fp32 windows gathered through the boundary model, tiles cut by the region's end, an x
// operand that needs the replicate clamp); MODE 0: everything else, sample by sample.
// ZERO: the pass's boundary model is PB_ZERO -- a compile-time fact of the instantiation, as in conv_w128.hip: with the model a
// run-time branch inside the loaders every pass of the circular domain was 3 - 5 % slower (same box: three steps 0.2442 ->
// 0.2525 ms, one pass on 64 x 64 windows 0.0774 -> 0.0815).
template <int MODE, typename TIn, typename TX, typename TOut, bool ZERO>
__device__ __forceinline__ void wave_pair(const ConvPass &a, const pb_blur_info *info, int plane, int ty, int pxi, int hx, int hy,
                                          char *zb, const float *kp, unsigned long long *tr) {
    constexpr int kBoundary = ZERO ? PB_ZERO : PB_WRAP;
    constexpr bool FAST = MODE != 0, GEN = MODE == 2;
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    PB_T(1);
    float2 *Z = reinterpret_cast<float2 *>(zb);
    float *Zf = reinterpret_cast<float *>(zb);
    const OutRegion rg = out_region(a);
    const int oy0 = rg.y_lo + ty * Ty;                              // the tile's first row, padded coordinates
    const int wxA = rg.x_lo + 2 * pxi * Tx - hx, wxB = wxA + Tx;    // window origins along x
    const int wrap_r = FT_N - hy;                                   // registers wrap_r .. 63 hold the rows above the tile
    const bool hasB = wxB + hx < rg.x_hi;
    const int lane = threadIdx.x & 63;
    const TIn *ipl = static_cast<const TIn *>(a.in) + (long)plane * a.in_plane;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    cf v[64];

    // ---- the window: lane = column, register = (rotated) row ----
    {
        const brsrc rin = plane_rsrc(ipl, a.in_plane);
        const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
        const int pitchb = a.in_pitch * (int)sizeof(TIn);
        const int wy0 = oy0 - hy;
        // byte offset of register r's row in the source plane (windows inside the source only)
        auto rowoff = [&](int r) -> int { return (oy0 - lo + r - (r >= wrap_r ? FT_N : 0)) * pitchb; };
        if constexpr (FAST) {
            // Interior fp32 pair on 16-byte boundaries: both windows go global -> LDS in 16-byte pieces (four-byte loads
            // straight into the registers cost one vector memory instruction per row and window, 128 per pair instead of 32:
            // the pass was bound by their issue).  An LDS row is 128 floats: pieces 0 .. 15 = window A's 64 columns, 16 .. 31
            // = window B's (the 2 hx columns the two share in memory are fetched twice -- the same cache lines, and B's
            // samples sit at a fixed distance from A's whatever the halo), one wave instruction fills two rows, sixteen rows at
            // a time through two LDS buffers: chunk k holds the registers 8 n1 + 2k, 8 n1 + 2k + 1 of the first-stage groups
            // n2 = 2k, 2k + 1, so the column transform starts on what has arrived while the rest is in flight, and every lane
            // picks its column's two samples per row with one ds_read2_b32.  (hy is even: the two rows of an instruction
            // never straddle the rotation's wrap.)  Two chunks are requested before the first is waited for and the next as
            // soon as a buffer has been read: one memory latency per pair.
            lds_char *zl = lds_ptr(zb);
            if constexpr (sizeof(TIn) == 4) {
                const int c = lane & 31;
                const unsigned vo = (unsigned)((lane >> 5) * pitchb + ((c < 16 ? wxA : wxB - FT_N) - lo + 4 * c) * 4);
                auto request = [&](int k, int buf) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dma16<0>(rin, zl + buf * 8192 + j * 1024, vo, rowoff(8 * j + 2 * k));
                };
                // (the LDS reads are issued behind the compiler's back: it would make every read of either buffer wait for ALL
                // outstanding LDS-DMA; the waits for the right chunk are placed by hand, and the wait that follows a chunk's
                // reads names their destinations, so that nothing using them can be scheduled above it)
                const unsigned la = lds_addr(zb) + (unsigned)lane * 4u;
                auto pick = [&](int k, int buf) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned ad = la + (unsigned)(buf * 8192 + j * 1024);
                        asm volatile("ds_read2_b32 %0, %1 offset1:64" : "=v"(v[8 * j + 2 * k]) : "v"(ad));
                        asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:192" : "=v"(v[8 * j + 2 * k + 1]) : "v"(ad));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[2 * k]), "+v"(v[2 * k + 1]), "+v"(v[8 + 2 * k]), "+v"(v[9 + 2 * k]), "+v"(v[16 + 2 * k]),
                                 "+v"(v[17 + 2 * k]), "+v"(v[24 + 2 * k]), "+v"(v[25 + 2 * k]), "+v"(v[32 + 2 * k]), "+v"(v[33 + 2 * k]), "+v"(v[40 + 2 * k]),
                                 "+v"(v[41 + 2 * k]), "+v"(v[48 + 2 * k]), "+v"(v[49 + 2 * k]), "+v"(v[56 + 2 * k]), "+v"(v[57 + 2 * k]) :: "memory");
                };
                bool pieces = true;
                if constexpr (GEN) pieces = wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && hasB && ((a.in_pitch | (wxA - lo)) & 3) == 0;
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
                } else if constexpr (GEN) {
                    // Border pairs of the circular domain: the same chunks through the same two LDS buffers, but gathered four
                    // bytes per lane through the boundary model -- lane = column of window A (one wave instruction = the A half
                    // of an LDS row) or of window B (its other half), the row mapped on the scalar side: 128 wave instructions
                    // that touch no register, then the same LDS reads.
                    // (the zero boundary: a column or a row outside the padded domain is an out-of-range offset -- the request
                    // returns zeros, the bounds check being on the lane's offset)
                    constexpr bool wrapb = !ZERO;
                    const int mxa = map_axis(wxA + lane, a.W, a.in_kind, kBoundary, a.pad);
                    const int mxb = map_axis((hasB ? wxB : wxA) + lane, a.W, a.in_kind, kBoundary, a.pad);   // (no window B: A's samples again -- finite, never stored)
                    const unsigned gcolA = mxa >= 0 ? (unsigned)mxa * 4u : kNoAccess, gcolB = mxb >= 0 ? (unsigned)mxb * 4u : kNoAccess;
                    const int base = wrapb ? __builtin_amdgcn_readfirstlane(wrap_idx(oy0, Hp)) : oy0;
                    const bool virt_in = a.in_kind == SRC_VIRTUAL;
                    auto gather = [&](int k, int buf) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {     // LDS row i of the chunk = register 8 (i >> 1) + 2 k + (i & 1)
                            const int r = 8 * (i >> 1) + 2 * k + (i & 1);
                            int pr = base + r - (r >= wrap_r ? FT_N : 0);
                            if (wrapb) {                      // (a branch of its own for each boundary model: the circular path pays nothing for the other)
                                while (pr < 0) pr += Hp;
                                while (pr >= Hp) pr -= Hp;
                                const int so = (virt_in ? min(max(pr - a.pad, 0), a.H - 1) : pr) * pitchb;
                                dma4<0>(rin, zl + buf * 8192 + i * 512, gcolA, so);
                                dma4<0>(rin, zl + buf * 8192 + i * 512 + 256, gcolB, so);
                            } else {
                                const bool ok = pr >= 0 && pr < Hp;
                                const int so = ok ? (virt_in ? min(max(pr - a.pad, 0), a.H - 1) : pr) * pitchb : 0;
                                dma4<0>(rin, zl + buf * 8192 + i * 512, ok ? gcolA : kNoAccess, so);
                                dma4<0>(rin, zl + buf * 8192 + i * 512 + 256, ok ? gcolB : kNoAccess, so);
                            }
                        }
                    };
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
            } else {
                // fp16 window (the first step, or the one-pass polynomial, of an fp16 image): a 16-byte piece is eight samples
                // and windows start on multiples of four, so each window is fetched from the 16-byte boundary at or before
                // its first column -- nine pieces; pieces 0 .. 15 of an LDS row belong to window A, 16 .. 31 to window B, lanes
                // past a window's ninth piece repeat it -- and every lane picks its two samples (two bytes each) at its
                // window's offset from that boundary.  Same chunks, same waits.
                const int c = lane & 31, cc = min(c & 15, 8);
                const int eA = wxA - lo, eB = wxB - lo;             // first column of each window, in samples from the row start
                const unsigned vo = (unsigned)((lane >> 5) * pitchb + (((c < 16 ? eA : eB) & ~7) + 8 * cc) * 2);
                auto request = [&](int k, int buf) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dma16<0>(rin, zl + buf * 8192 + j * 1024, vo, rowoff(8 * j + 2 * k));
                };
                const unsigned la = lds_addr(zb) + (unsigned)(((eA & 7) + lane) * 2), lb = lds_addr(zb) + 256u + (unsigned)(((eB & 7) + lane) * 2);
                // (plain 16-bit reads, one register per sample: the d16 forms that fill half a register clear the other half
                // on this hardware)
                auto pick = [&](int k, int buf) {
                    unsigned ra[16], rb[16];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned oa = la + (unsigned)(buf * 8192 + j * 1024), ob = lb + (unsigned)(buf * 8192 + j * 1024);
                        asm volatile("ds_read_u16 %0, %1" : "=v"(ra[2 * j]) : "v"(oa));
                        asm volatile("ds_read_u16 %0, %1" : "=v"(rb[2 * j]) : "v"(ob));
                        asm volatile("ds_read_u16 %0, %1 offset:512" : "=v"(ra[2 * j + 1]) : "v"(oa));
                        asm volatile("ds_read_u16 %0, %1 offset:512" : "=v"(rb[2 * j + 1]) : "v"(ob));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]),
                                 "+v"(ra[7]), "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]),
                                 "+v"(ra[15]) :: "memory");
                    asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]), "+v"(rb[8]),
                                 "+v"(rb[9]), "+v"(rb[10]), "+v"(rb[11]), "+v"(rb[12]), "+v"(rb[13]), "+v"(rb[14]), "+v"(rb[15]) :: "memory");
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int r = 8 * (j >> 1) + 2 * k + (j & 1);
                        v[r] = (cf){__half2float(__builtin_bit_cast(__half, (unsigned short)ra[j])), __half2float(__builtin_bit_cast(__half, (unsigned short)rb[j]))};
                    }
                };
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
            }
        } else if (sizeof(TIn) == 4) {
            // Border pairs (the circular domain, or the zero boundary's), fp32: the same chunks through the same two LDS buffers, but gathered four
            // bytes per lane through the boundary model -- lane = column of window A (one wave instruction = the A half of an
            // LDS row) or of window B (its other half), the row mapped on the scalar side: 128 wave instructions that touch no
            // register, then the loader's own LDS reads (sample by sample into the registers this took 128 loads per lane
            // and made the border pairs -- 8 % of the pairs at 4K, 15 % at 1080p -- the stragglers of every launch).
            lds_char *zl = lds_ptr(zb);
            constexpr bool wrapb = !ZERO;
            const int mxa = map_axis(wxA + lane, a.W, a.in_kind, kBoundary, a.pad);
            const int mxb = map_axis((hasB ? wxB : wxA) + lane, a.W, a.in_kind, kBoundary, a.pad);   // (no window B: A's samples again -- finite, never stored)
            const unsigned gcolA = mxa >= 0 ? (unsigned)mxa * 4u : kNoAccess, gcolB = mxb >= 0 ? (unsigned)mxb * 4u : kNoAccess;
            const int base = wrapb ? __builtin_amdgcn_readfirstlane(wrap_idx(oy0, Hp)) : oy0;
            const bool virt_in = a.in_kind == SRC_VIRTUAL;
            auto gather = [&](int k, int buf) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {     // LDS row i of the chunk = register 8 (i >> 1) + 2 k + (i & 1)
                    const int r = 8 * (i >> 1) + 2 * k + (i & 1);
                    int pr = base + r - (r >= wrap_r ? FT_N : 0);
                    if (wrapb) {
                        while (pr < 0) pr += Hp;
                        while (pr >= Hp) pr -= Hp;
                        const int so = (virt_in ? min(max(pr - a.pad, 0), a.H - 1) : pr) * pitchb;
                        dma4<0>(rin, zl + buf * 8192 + i * 512, gcolA, so);
                        dma4<0>(rin, zl + buf * 8192 + i * 512 + 256, gcolB, so);
                    } else {
                        const bool ok = pr >= 0 && pr < Hp;
                        const int so = ok ? (virt_in ? min(max(pr - a.pad, 0), a.H - 1) : pr) * pitchb : 0;
                        dma4<0>(rin, zl + buf * 8192 + i * 512, ok ? gcolA : kNoAccess, so);
                        dma4<0>(rin, zl + buf * 8192 + i * 512 + 256, ok ? gcolB : kNoAccess, so);
                    }
                }
            };
            const unsigned la = lds_addr(zb) + (unsigned)lane * 4u;
            auto pick = [&](int k, int buf) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned ad = la + (unsigned)(buf * 8192 + j * 1024);
                    asm volatile("ds_read2_b32 %0, %1 offset1:64" : "=v"(v[8 * j + 2 * k]) : "v"(ad));
                    asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:192" : "=v"(v[8 * j + 2 * k + 1]) : "v"(ad));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[2 * k]), "+v"(v[2 * k + 1]), "+v"(v[8 + 2 * k]), "+v"(v[9 + 2 * k]), "+v"(v[16 + 2 * k]),
                             "+v"(v[17 + 2 * k]), "+v"(v[24 + 2 * k]), "+v"(v[25 + 2 * k]), "+v"(v[32 + 2 * k]), "+v"(v[33 + 2 * k]), "+v"(v[40 + 2 * k]),
                             "+v"(v[41 + 2 * k]), "+v"(v[48 + 2 * k]), "+v"(v[49 + 2 * k]), "+v"(v[56 + 2 * k]), "+v"(v[57 + 2 * k]) :: "memory");
            };
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
        } else if (wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && hasB) {
            const unsigned colA = (unsigned)(wxA - lo + lane) * (unsigned)sizeof(TIn), colB = colA + (unsigned)Tx * (unsigned)sizeof(TIn);
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const int y = 8 * (q & 7) + (q >> 3);
                const int so = rowoff(y);
                v[y] = (cf){BufIO<TIn>::ld(rin, colA, so), BufIO<TIn>::ld(rin, colB, so)};
            }
        } else {
            // border window: columns mapped through the boundary model once per lane, rows on the scalar side
            const int ixa = map_axis(wxA + lane, a.W, a.in_kind, kBoundary, a.pad);
            // (no window B: window A's samples again -- finite, never stored --, as the LDS-DMA loaders of fp32 planes have it: what the
            // imaginary half holds reaches the real half's ROUNDING, and an 8-bit or fp16 image must get bit for bit what its float
            // copy gets -- tests/test_gpu_parity.py::test_uint8_edge)
            const int ixb = map_axis((hasB ? wxB : wxA) + lane, a.W, a.in_kind, kBoundary, a.pad);
            const unsigned colA = ixa >= 0 ? (unsigned)ixa * (unsigned)sizeof(TIn) : kNoAccess;
            const unsigned colB = ixb >= 0 ? (unsigned)ixb * (unsigned)sizeof(TIn) : kNoAccess;
            constexpr bool wrap = !ZERO;
            const int base = wrap ? __builtin_amdgcn_readfirstlane(wrap_idx(oy0, Hp)) : oy0;
            // (planes at least a window tall: one conditional step brings a row into the circular domain -- straight-line code,
            // the 128 loads in flight together; the loops of a shorter plane end a basic block per row, and every pair of loads
            // is then waited for before the next is issued)
            auto rows = [&](auto tall) {
#pragma unroll
                for (int q = 0; q < 64; ++q) {
                    const int y = 8 * (q & 7) + (q >> 3);
                    int p = base + y - (y >= wrap_r ? FT_N : 0);
                    if (wrap) {
                        if (decltype(tall)::value) { p += p < 0 ? Hp : 0; p -= p >= Hp ? Hp : 0; }
                        else { while (p < 0) p += Hp; while (p >= Hp) p -= Hp; }
                    }
                    const bool ok = wrap || (p >= 0 && p < Hp);
                    const int iy = ok ? (a.in_kind == SRC_VIRTUAL ? min(max(p - a.pad, 0), a.H - 1) : p) : 0;
                    const int so = iy * pitchb;
                    v[y] = (cf){BufIO<TIn>::ld(rin, ok ? colA : kNoAccess, so), BufIO<TIn>::ld(rin, ok ? colB : kNoAccess, so)};
                }
            };
            if (Hp >= FT_N) rows(std::true_type()); else rows(std::false_type());
        }
    }
    PB_T(2);
    fft64_fwd(v);                                               // columns
    PB_T(3);
    transpose64(v, Z, lane);
    PB_T(4);
    {
        // the image's spectrum (16 KB, resident in L2), [x position][y position]: lane = transformed row py reads kh[px][py]
        // for every px -- 64 coalesced requests that travel while the first row stage runs (through LDS they cost a
        // DMA pass, 64 LDS reads and the wait for both)
        const brsrc rk = plane_rsrc(kp, (long)FT_N * FT_N);
        // The 64 values travel in a ring of four groups of eight (the centre stage of group k1 multiplies by the values
        // 8 k1 .. 8 k1 + 7): four groups are requested before the first row stage, group k1 + 4 when group k1 is done.  All 64
        // at once -- 64 registers beside the window pair's 128 and the butterflies' -- leaves the scheduler short of
        // registers: it then either sinks the requests into the centre stage seven at a time (every batch exposing an L2
        // latency: 25 k cycles per pair for these two stages instead of 7 k) or spills.  The sched_barriers pin the
        // requests (nothing is scheduled across them; with a mask that lets arithmetic pass, the requests sank all the same).
        float kh[4][8];
        auto khload = [&](int grp) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) kh[grp & 3][k2] = BufIO<float>::ld(rk, (unsigned)lane * 4u, (8 * grp + k2) * (FT_N * 4));
        };
        khload(0); khload(1); khload(2); khload(3);
        __builtin_amdgcn_sched_barrier(0);
        fft64_fwd_stage1(v);                                    // rows
        PB_T(5);
        // stage 2, x spectrum, inverse stage 2
        centre_stage<0>(v, kh[0]); khload(4); __builtin_amdgcn_sched_barrier(0);
        centre_stage<1>(v, kh[1]); khload(5); __builtin_amdgcn_sched_barrier(0);
        centre_stage<2>(v, kh[2]); khload(6); __builtin_amdgcn_sched_barrier(0);
        centre_stage<3>(v, kh[3]); khload(7); __builtin_amdgcn_sched_barrier(0);
        centre_stage<4>(v, kh[0]); centre_stage<5>(v, kh[1]); centre_stage<6>(v, kh[2]); centre_stage<7>(v, kh[3]);
    }
    fft64_inv_stage1(v);
    PB_T(6);
    transpose64(v, Z, lane);
    PB_T(7);

    // ---- epilogue: lane = window column again, register r = tile row r ----
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const int xmax = virt ? a.W - 1 : Wp - 1, ymax = virt ? a.H - 1 : Hp - 1, xsh = virt ? a.pad : 0;
    const int xpitchb = a.x_pitch * (int)sizeof(TX), opitchb = a.out_pitch * (int)sizeof(TOut);
    const brsrc rx = plane_rsrc(xpl, a.x_plane);
    const brsrc ro = plane_rsrc(opl, a.out_plane);
    const bool colin = lane >= hx && lane < FT_N - hx;
    const float sc = a.scale, cfx = a.coef;
    const bool cl = a.clamp01 != 0;
    const bool taper = a.epilogue == EPI_TAPER;
    const bool usex = taper || cfx != 0.f;                      // (the one-pass polynomial has no x operand: beta sits in its spectrum)
    const int oxA = wxA + hx;
    if constexpr (FAST) {
        // Complete interior pair, plain Horner epilogue, everything on 16-byte boundaries: the 2 Tx-wide block of outputs goes
        // through an LDS tile (written by columns, read back as 16-byte row pieces) so that the x operand arrives and the
        // result leaves in 16-byte accesses: registers 0 .. 31 first, then 32 .. 63 (the tile holds 32 rows), each half read
        // back in rounds of eight rows.  C pieces per row; lane -> (row of the round, piece) for each of the four vector
        // memory instructions of a round.  The x operand travels four rounds ahead (memory latency under load is thousands
        // of cycles).
        //   The window pair never lives across a branch: everything up to the second half's writes is straight-line code and
        // every vector memory operation is issued unconditionally (a piece that does not exist gets an out-of-range offset).
        // With one branch per round the compiler shuffled forty register pairs per round between its blocks and -- it counts
        // outstanding requests per path and assumes the fewest at a join -- waited for younger requests than the round needed;
        // with the rounds in a switch, or a copy of the epilogue per kind of pass, it spilled a hundred registers around the
        // last transform.  Only the rounds of the second half, when the pair is in LDS, are conditional.
        const int C = Tx >> 1;
        const float invC = __builtin_amdgcn_rcpf((float)C);
        constexpr int XP = 4 * sizeof(TX), OP = 4 * sizeof(TOut);                 // bytes per piece of x / of the output
        int rlk[4], chk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = 64 * k + lane;
            rlk[k] = div_small(e, invC); chk[k] = e - rlk[k] * C;                 // (e < 256, C = 4 .. 28: exact)
        }
        const int xso = (oy0 - xsh) * xpitchb + (oxA - xsh) * (int)sizeof(TX);
        const int oso = (oy0 - oo) * opitchb + (oxA - oo) * (int)sizeof(TOut);
        // MODE 2 (border pairs): rows beyond the region's end fall away (tyr), whole pieces beyond its right end too (the end
        // lies on a piece boundary: bit k of `cut`); the x operand of a tile in the pad ring is the replicate-clamped image --
        // rows clamped per piece, and a piece left (right) of the image is the image's first (last) sample four times (the
        // image starts and ends on piece boundaries: bits 4 + k and 8 + k of `cut`).  One register of flags per lane; the
        // interior pairs' code (MODE 1) is unchanged.
        const int tyr = GEN ? min(Ty, rg.y_hi - oy0) : Ty;
        const int xw = virt ? a.W : Wp, xh = virt ? a.H : Hp;
        int cut = 0, cxb[4] = {0, 0, 0, 0};
        if constexpr (GEN) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cx = oxA - xsh + 4 * chk[k];              // the piece's first column in the x plane
                if (oxA + 4 * chk[k] + 4 > rg.x_hi) cut |= 1 << k;
                if (cx < 0) cut |= 16 << k;
                if (cx > xw - 4) cut |= 256 << k;
                cxb[k] = min(max(cx, 0), xw - 4) * (int)sizeof(TX);
            }
        }
        typename Piece4<TX>::raw xq[4][4];
        auto request = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int left = tyr - 8 * q;                       // rows of this round inside the tile (<= 0: none)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (GEN) {
                    const bool ok = usex && rlk[k] < min(left, 8) && !(cut & (1 << k));
                    const int xr = min(max(oy0 - xsh + 8 * q + rlk[k], 0), xh - 1);
                    xq[q & 3][k] = Piece4<TX>::ld(rx, ok ? (unsigned)(xr * xpitchb + cxb[k]) : kNoAccess, 0);
                } else {
                    const bool ok = usex && rlk[k] < min(left, 8);
                    xq[q & 3][k] = Piece4<TX>::ld(rx, ok ? (unsigned)(rlk[k] * xpitchb + chk[k] * XP) : kNoAccess, xso + 8 * q * xpitchb);
                }
            }
        };
        // (the final clamp without a branch per piece: the bounds are infinite where the pass does not clamp)
        const float clo = cl ? 0.f : -INFINITY, chi = cl ? 1.f : INFINITY;
        auto round = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int left = tyr - 8 * q;
            f4v acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = *reinterpret_cast<const f4v *>(Zf + (8 * (q & 3) + min(rlk[k], 7)) * 128 + 4 * chk[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f4v x4 = Piece4<TX>::to_f(xq[q & 3][k]);
                if constexpr (GEN) {
                    if (cut & (16 << k)) { x4.y = x4.x; x4.z = x4.x; x4.w = x4.x; }
                    if (cut & (256 << k)) { x4.x = x4.w; x4.y = x4.w; x4.z = x4.w; }
                }
                f4v o;
                o.x = fmaf(sc, acc[k].x, cfx * x4.x); o.y = fmaf(sc, acc[k].y, cfx * x4.y);
                o.z = fmaf(sc, acc[k].z, cfx * x4.z); o.w = fmaf(sc, acc[k].w, cfx * x4.w);
                o.x = __builtin_amdgcn_fmed3f(o.x, clo, chi); o.y = __builtin_amdgcn_fmed3f(o.y, clo, chi);
                o.z = __builtin_amdgcn_fmed3f(o.z, clo, chi); o.w = __builtin_amdgcn_fmed3f(o.w, clo, chi);
                const bool ok = rlk[k] < min(left, 8) && !(GEN && (cut & (1 << k)));
                Piece4<TOut>::st(ro, ok ? (unsigned)(rlk[k] * opitchb + chk[k] * OP) : kNoAccess, oso + 8 * q * opitchb, o);
            }
        };
        float *zt = Zf + (lane - hx);
        auto put_half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if (colin) {
#pragma unroll
                for (int i = 0; i < 32; ++i) { zt[i * 128] = v[32 * h + i].x; zt[i * 128 + Tx] = v[32 * h + i].y; }
            }
        };
        typedef std::integral_constant<int, 0> Q0; typedef std::integral_constant<int, 1> Q1; typedef std::integral_constant<int, 2> Q2;
        typedef std::integral_constant<int, 3> Q3; typedef std::integral_constant<int, 4> Q4; typedef std::integral_constant<int, 5> Q5;
        typedef std::integral_constant<int, 6> Q6; typedef std::integral_constant<int, 7> Q7;
        // (sixteen registers a round of the x operand: one round's worth fits beside each stage of the last transform; the
        // barriers keep the scheduler from hoisting the later requests into the transform)
        request(Q0{});
        __builtin_amdgcn_sched_barrier(0);
        fft64_inv_stage2(v);                                    // columns
        __builtin_amdgcn_sched_barrier(0);
        request(Q1{});
        __builtin_amdgcn_sched_barrier(0);
        fft64_inv_stage1(v);
        __builtin_amdgcn_sched_barrier(0);
        PB_T(8);
        put_half(Q0{});
        request(Q2{}); request(Q3{});
        wave_lds_fence();
        round(Q0{}); request(Q4{});
        round(Q1{}); request(Q5{});
        round(Q2{}); request(Q6{});
        round(Q3{}); request(Q7{});
        wave_lds_fence();
        put_half(Q1{});                                         // (behind the first half's reads: a wave's LDS operations execute in order)
        wave_lds_fence();
        if (Ty > 32) {
            round(Q4{});
            if (Ty > 40) {
                round(Q5{});
                if (Ty > 48) {
                    round(Q6{});
                    if (Ty > 56) round(Q7{});
                }
            }
        }
        PB_T(9);
        PB_TWAIT();
        PB_T(10);
        PB_TRT(13);
        return;
    }
    const int tyv = min(Ty, rg.y_hi - oy0);                     // rows of the tile inside the output region
    const int pxA = wxA + lane, pxB = wxB + lane;
    const bool okA = colin && pxA < rg.x_hi, okB = colin && hasB && pxB < rg.x_hi;
    const unsigned xoffA = okA && usex ? (unsigned)min(max(pxA - xsh, 0), xmax) * (unsigned)sizeof(TX) : kNoAccess;
    const unsigned xoffB = okB && usex ? (unsigned)min(max(pxB - xsh, 0), xmax) * (unsigned)sizeof(TX) : kNoAccess;
    const unsigned ooffA = okA ? (unsigned)(pxA - oo) * (unsigned)sizeof(TOut) : kNoAccess;
    const unsigned ooffB = okB ? (unsigned)(pxB - oo) * (unsigned)sizeof(TOut) : kNoAccess;
    float txa = 0.f, txb = 0.f, tyl = 0.f;
    if (taper) {
        txa = taper_weight(info->acorr_x, min(max(pxA, 0), Wp - 1), Wp);
        txb = taper_weight(info->acorr_x, min(max(pxB, 0), Wp - 1), Wp);
        // the rows' weights: lane y holds tile row y's, read back with a constant lane number below (one weight per row through
        // the scalar unit was two dependent scalar loads in front of each of the tile's 64 rows: the blends' border pairs --
        // all the pairs a blend has left -- took ~60 us at 4K)
        tyl = taper_weight(info->acorr_y, min(oy0 + lane, Hp - 1), Hp);
    }
    // Border pairs, the taper blend, narrower types: the last transform's second stage finishes the registers 8 n1 + n2
    // group by group (n2 = 0 .. 7); each group's rows go through the epilogue and to memory at once, and the x operand
    // travels in a ring four groups deep -- the loads of group n2 + 4 are issued when group n2 has been stored.  (Rows
    // outside the tile load and store at an out-of-range offset: nothing happens.)
    float xa[8][8], xb[8][8];
    auto request = [&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int y = 8 * n1 + n2;
            const bool rok = y < tyv;
            const int xr = min(max(oy0 + y - xsh, 0), ymax);
            const int so = xr * xpitchb;
            xa[n2][n1] = BufIO<TX>::ld(rx, rok ? xoffA : kNoAccess, so); xb[n2][n1] = BufIO<TX>::ld(rx, rok ? xoffB : kNoAccess, so);
        }
    };
    auto finish = [&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
        inv_stage1<n2>(v);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int y = 8 * n1 + n2;
            const int py = oy0 + y;
            const bool rok = y < tyv;
            float ra, rb;
            if (taper) {
                const float tyw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tyl), y));
                const float ala = tyw * txa, alb = tyw * txb;
                ra = ala * xa[n2][n1] + (1.f - ala) * v[y].x; rb = alb * xb[n2][n1] + (1.f - alb) * v[y].y;
            } else {
                ra = fmaf(sc, v[y].x, cfx * xa[n2][n1]); rb = fmaf(sc, v[y].y, cfx * xb[n2][n1]);
            }
            if (cl) { ra = fminf(fmaxf(ra, 0.f), 1.f); rb = fminf(fmaxf(rb, 0.f), 1.f); }
            const int so = (py - oo) * opitchb;
            BufIO<TOut>::st(ro, rok ? ooffA : kNoAccess, so, ra); BufIO<TOut>::st(ro, rok ? ooffB : kNoAccess, so, rb);
        }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
    typedef std::integral_constant<int, 6> I6; typedef std::integral_constant<int, 7> I7;
    request(I0{}); request(I1{}); request(I2{}); request(I3{});
    fft64_inv_stage2(v);                                        // columns
    PB_T(8);
    finish(I0{}); request(I4{});
    finish(I1{}); request(I5{});
    finish(I2{}); request(I6{});
    finish(I3{}); request(I7{});
    finish(I4{}); finish(I5{}); finish(I6{}); finish(I7{});
    PB_T(9);
    PB_TWAIT();
    PB_T(10);
    PB_TRT(13);
}

// Whether a pair takes the all-16-byte path: an fp32 or fp16 window, plain Horner epilogue, both windows inside the source
// without boundary mapping, both tiles complete inside the output region, the x operand addressed without clamping, and
// rows / origins on 16-byte boundaries.
template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ bool pair_is_fast(const ConvPass &a, int ty, int pxi, int hx, int hy) {
    // (the x operand and the output may be fp16: four samples are then an 8-byte piece)
    if (sizeof(TIn) < 2 || a.epilogue != EPI_HORNER) return false;      // (an 8-bit window -- the first step of an 8-bit image -- is fetched sample by sample)
    if (sizeof(TIn) == 2 && (a.in_pitch & 7) != 0) return false;      // (fp16 window: rows on 16-byte boundaries)
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    const OutRegion rg = out_region(a);
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const int oy0 = rg.y_lo + ty * Ty, wy0 = oy0 - hy, wxA = rg.x_lo + 2 * pxi * Tx - hx, wxB = wxA + Tx;
    const int lo = a.in_kind == SRC_VIRTUAL ? a.pad : 0;
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0, xsh = virt ? a.pad : 0;
    const int xw = virt ? a.W : Wp, xh = virt ? a.H : Hp;
    const int oxA = wxA + hx;
    return wy0 >= lo && wy0 + FT_N <= Hp - lo && wxA >= lo && wxB + FT_N <= Wp - lo && oy0 + Ty <= rg.y_hi && oxA + 2 * Tx <= rg.x_hi &&
           oy0 - xsh >= 0 && oxA - xsh >= 0 && oy0 + Ty - xsh <= xh && oxA + 2 * Tx - xsh <= xw &&
           ((a.in_pitch | a.x_pitch | a.out_pitch | (wxA - lo) | (oxA - xsh) | (oxA - oo)) & 3) == 0;
}

// Whether a border pair takes the same structure (MODE 2): fp32 windows of the circular domain (gathered through the boundary
// model where they are not inside the source), a plain Horner epilogue, rows and the region's and the x plane's ends on
// 16-byte boundaries.
template <typename TIn, typename TX, typename TOut>
__device__ __forceinline__ bool pair_is_gen(const ConvPass &a, int pxi, int hx) {
    if (sizeof(TIn) != 4 || a.epilogue != EPI_HORNER) return false;      // (either boundary model: the gather maps it per lane and row)
    const int Tx = FT_N - 2 * hx;
    const OutRegion rg = out_region(a);
    const int Wp = a.W + 2 * a.pad;
    const bool virt = a.x_kind == SRC_VIRTUAL;
    const int oo = a.out_kind == OUT_INTERIOR ? a.pad : 0, xsh = virt ? a.pad : 0, xw = virt ? a.W : Wp;
    const int oxA = rg.x_lo + 2 * pxi * Tx;
    return xw >= 4 && ((a.x_pitch | a.out_pitch | (oxA - xsh) | (oxA - oo) | (rg.x_hi - oxA) | xw) & 3) == 0;
}

// A taper blend (edgetaper.py:26-33) whose weight alpha = v1[py] v2[px] is exactly 1 on the whole tile pair -- both tiles at
// least 25 samples from every border of the padded domain, where the kernel's autocorrelation has no lag left
// (taper_weight: 1 - 0 / z[0]) -- is out = 1 x + 0 (K * in) = x: the pair is COPIED, no window fetched, no transform run.
// 91 % of the pairs of a 4K taper pass (every pair took the sample-by-sample form before: 3 blends were 60 % of a call with
// edgetaping).  Finite operands assumed, as everywhere (0 * inf would be NaN in the blend).
__device__ __forceinline__ bool taper_is_copy(const ConvPass &a, int ty, int pxi, int hx, int hy) {
    if (a.epilogue != EPI_TAPER) return false;
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    const OutRegion rg = out_region(a);
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const int y0 = rg.y_lo + ty * Ty, y1 = min(y0 + Ty, rg.y_hi), x0 = rg.x_lo + 2 * pxi * Tx, x1 = min(x0 + 2 * Tx, rg.x_hi);
    return y0 >= PB_KSIZE && y1 <= Hp - PB_KSIZE && x0 >= PB_KSIZE && x1 <= Wp - PB_KSIZE;
}
template <typename TX, typename TOut>
__device__ __forceinline__ void copy_pair(const ConvPass &a, int plane, int ty, int pxi, int hx, int hy) {
    const int lane = threadIdx.x & 63;
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    const OutRegion rg = out_region(a);
    const int y0 = rg.y_lo + ty * Ty, y1 = min(y0 + Ty, rg.y_hi), x0 = rg.x_lo + 2 * pxi * Tx, x1 = min(x0 + 2 * Tx, rg.x_hi);
    const int xsh = a.x_kind == SRC_VIRTUAL ? a.pad : 0, oo = a.out_kind == OUT_INTERIOR ? a.pad : 0;
    const TX *xpl = static_cast<const TX *>(a.x) + (long)plane * a.x_plane;
    TOut *opl = static_cast<TOut *>(a.out) + (long)plane * a.out_plane;
    // (the pair lies at least 25 samples inside the padded domain: inside the image too, no clamp)
    if (sizeof(TX) == 4 && sizeof(TOut) == 4 && ((a.x_pitch | a.out_pitch | (x0 - xsh) | (x0 - oo) | (x1 - x0)) & 3) == 0) {
        const int n4 = (x1 - x0) >> 2;                          // 16-byte pieces per row (<= 28)
        const int per = 64 / n4;                                // rows per wave instruction
        const int rl = lane / n4, pc = lane - rl * n4;
        if (rl < per) {
            for (int r = y0 + rl; r < y1; r += per) {
                const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(xpl) + (long)(r - xsh) * a.x_pitch + (x0 - xsh) + 4 * pc);
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(opl) + (long)(r - oo) * a.out_pitch + (x0 - oo) + 4 * pc) = v;
            }
        }
        return;
    }
    for (int r = y0; r < y1; ++r)
        for (int c = x0 + lane; c < x1; c += 64)
            pb_st(opl + (long)(r - oo) * a.out_pitch + (c - oo), pb_ld(xpl + (long)(r - xsh) * a.x_pitch + (c - xsh)));
}

// The border ring of a zero-boundary polynomial (ConvPass.ring = Horner step 1 / 2 / 3).  Under the zero boundary the three
// steps differ from the one window pass with the polynomial's spectrum only where a step's truncation to the padded domain
// (filters.py:40-49: F.conv2d pads every step's operand with zeros) is within reach: outputs within 24 samples of the padded
// border.  Step 3 therefore recomputes the pairs whose output rectangle comes within 24 samples of the border; they lie
// within 24 + Ty (24 + 2 Tx along x: pairs) of it and read t2 12 further; step 2 the pairs that come within 36 + Ty
// (36 + 2 Tx), which lie within 36 + 2 Ty (36 + 4 Tx) and read t1 12 further; step 1 the pairs that come within 48 + 2 Ty
// (48 + 4 Tx).  Every sample a live pair reads was written by a live pair of the step before.
__device__ __forceinline__ bool ring_live(const ConvPass &a, int ty, int pxi, int hx, int hy) {
    const int Tx = FT_N - 2 * hx, Ty = FT_N - 2 * hy;
    const OutRegion rg = out_region(a);
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const int y0 = rg.y_lo + ty * Ty, y1 = min(y0 + Ty, rg.y_hi), x0 = rg.x_lo + 2 * pxi * Tx, x1 = min(x0 + 2 * Tx, rg.x_hi);
    const int dy = min(y0, Hp - y1), dx = min(x0, Wp - x1);
    if (a.ring >= 4) {
        // the second (ring 5) and third (ring 4) blend of an edgetaper (edgetaper.py:26-33): alpha < 1 only within 24 samples of
        // the padded border, so the third blend computes the pairs that come within 25 of it -- everything else of its output
        // plane still holds the first blend's copy of the image -- and the second the pairs those read, 12 further
        const int mt = a.ring - 4;
        return dy < 25 + 12 * mt + mt * Ty || dx < 25 + 12 * mt + 2 * mt * Tx;
    }
    const int m = 3 - a.ring;                                    // 0 for step 3, 1 for step 2, 2 for step 1
    return dy < 24 + 12 * m + m * Ty || dx < 24 + 12 * m + 2 * m * Tx;
}

// One wave (= one workgroup) per window pair; the GRID is the job list.  The jobs are the window pairs of the images whose
// record selects this body, every image with its own halos and therefore its own tile size (prefix sum over the batch's
// pb_fft_sel records: no host read-back; the grid is sized for the smallest tile the records may select and the surplus
// workgroups leave at once).  Workgroup b belongs to list b % 8 (the XCD it is observed to run on -- used for speed only)
// at position b / 8; every list owns the same eighth of EVERY plane's pairs -- a contiguous run, so neighbouring windows
// share their halos in that XCD's L2 -- in the order image, plane, pair.
//
// (Measured and dropped, round 3: the three Horner steps of a polynomial in ONE launch -- step-major lists, per-plane
// completion counters, write-through stores, one agent-scope acquire per pair: with every synchronisation compiled out
// the single launch takes exactly what the three launches take; three waves per SIMD instead of two -- 168 registers, the
// spectrum streamed, a 16-row transpose tile: 119 us per 4K pass against 85, nothing can be requested far enough ahead.
// NOTEBOOK.md has the records.)
template <typename TIn, typename TX, typename TOut, bool ZERO>
__global__ __launch_bounds__(64, 2) void conv_wfft_kernel(const ConvPass a, const WGeom g) {
    extern __shared__ __attribute__((aligned(16))) char zb[];
    const int lane = threadIdx.x & 63;
    unsigned long long *tr = nullptr;
#ifdef PB_WF_TRACE
    if (blockIdx.x < kTraceWaves) tr = g_wf_trace + (long)blockIdx.x * kTraceStamps;
    PB_T(0);
    PB_TRT(12);
#endif
    const int C = a.C, B = a.P / C;
    const int q = (int)(blockIdx.x & 7u);
    int rem = (int)(blockIdx.x >> 3);                  // position in the list
    int img = 0, hx = 0, hy = 0;
    bool fold = false;
    if (B == 1) {
        // (one image: its record is read on the scalar side -- no trip through the vector memory queue)
        const PB_CONSTANT pb_fft_sel *s0 = as_constant(a.fsel);
        if (!s0->use_fft || s0->poly == 2 || !poly_match(a.poly, s0->poly)) return;     // (poly == 2: conv_w128.hip's image)
        hx = s0->hx; hy = s0->hy;
        fold = a.poly == 2 && s0->poly != 0;
    } else {
        // list entries of image i: its share of every plane
        auto share_of = [&](int i) -> int {
            if (i >= B) return 0;
            const pb_fft_sel s = a.fsel[i];
            if (!s.use_fft || s.poly == 2 || !poly_match(a.poly, s.poly)) return 0;
            return jobs_of(g, s.hx, s.hy).per * C;
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
        fold = a.poly == 2 && as_constant(a.fsel + img)->poly != 0;
    }
    const WJobs j = jobs_of(g, hx, hy);
    const int pl = __builtin_amdgcn_readfirstlane(div_rcp(rem, __builtin_amdgcn_rcpf((float)j.per)));
    if (pl >= C) return;                               // (one image: positions beyond its planes)
    const int pair = q * j.per + (rem - pl * j.per);
    if (pair >= j.njobs) return;                       // (the ragged end of the last list's run)
    const int ty = __builtin_amdgcn_readfirstlane(div_rcp(pair, j.inv_pairs_x)), pxi = pair - ty * j.pairs_x;
    const int plane = img * C + pl;
    const float *kp = a.khat + (long)img * PB_KHAT_STRIDE;
    const pb_blur_info *info = a.info + img;
    if (a.ring && !ring_live(a, ty, pxi, hx, hy)) return;
    const ConvPass af = fold_pass(a, fold);
    if (taper_is_copy(af, ty, pxi, hx, hy)) { copy_pair<TX, TOut>(af, plane, ty, pxi, hx, hy); return; }
    if (pair_is_fast<TIn, TX, TOut>(af, ty, pxi, hx, hy)) wave_pair<1, TIn, TX, TOut, ZERO>(af, info, plane, ty, pxi, hx, hy, zb, kp, tr);
    else if (pair_is_gen<TIn, TX, TOut>(af, pxi, hx)) wave_pair<2, TIn, TX, TOut, ZERO>(af, info, plane, ty, pxi, hx, hy, zb, kp, tr);
    else wave_pair<0, TIn, TX, TOut, ZERO>(af, info, plane, ty, pxi, hx, hy, zb, kp, tr);
}

// The output extent of a pass and the largest job list its records may ask for: `poly2` = the records may carry one-pass
// images with the composite filter's halos (PolySpec.on == 2: tiles down to PB_POLY_MIN_TX x PB_POLY_MIN_TY, but never
// smaller in area than the cost model of khat.h admits); otherwise halos are at most 12.  pairs12 = window pairs per plane at
// the 12-sample halo.  (Counts stay below 2^20: the kernel divides with reciprocals.)
bool wfft_geometry(const ConvPass &p, bool poly2, float min_area, WGeom &g, long &per_max, long &pairs12) {
    g.oh = (p.out_kind == OUT_INTERIOR) ? p.H : p.H + 2 * p.pad;
    g.ow = (p.out_kind == OUT_INTERIOR) ? p.W : p.W + 2 * p.pad;
    per_max = 0; pairs12 = 0;
    for (int hx = 4; hx <= 28; hx += 4) {
        for (int hy = 2; hy <= 30; hy += 2) {
            const int tx = FT_N - 2 * hx, ty = FT_N - 2 * hy;
            const bool three = hx <= 12 && hy <= 12;
            const bool one = poly2 && tx >= PB_POLY_MIN_TX && ty >= PB_POLY_MIN_TY && (float)(tx * ty) >= min_area;
            if (!three && !one) continue;
            const long nj = (long)(((g.ow + tx - 1) / tx + 1) / 2) * ((g.oh + ty - 1) / ty);
            if (nj > (1L << 20)) return false;
            per_max = std::max(per_max, (nj + 7) / 8);
            if (hx == 12 && hy == 12) pairs12 = nj;
        }
    }
    const long total = 8 * per_max * p.P;
    return total > 0 && total <= (1L << 23);
}

template <typename TIn, typename TX, typename TOut>
int launch_wfft_typed(pb_ctx *ctx, const ConvPass &p, const WGeom &g, long groups) {
    if (p.boundary == PB_ZERO)
        hipLaunchKernelGGL((conv_wfft_kernel<TIn, TX, TOut, true>), dim3((unsigned)groups), dim3(64), kWfLdsWave, ctx->stream, p, g);
    else
        hipLaunchKernelGGL((conv_wfft_kernel<TIn, TX, TOut, false>), dim3((unsigned)groups), dim3(64), kWfLdsWave, ctx->stream, p, g);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace

#ifdef PB_WF_TRACE
extern "C" int pb_debug_wf_trace_clear(void) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wf_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * kTraceStamps * kTraceWaves);
}
extern "C" int pb_debug_wf_trace(unsigned long long *host, int n_waves) {
    if (n_waves > kTraceWaves) n_waves = kTraceWaves;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wf_trace), sizeof(unsigned long long) * kTraceStamps * n_waves);
}
#endif

// (as pb_conv_w128_feasible: the largest job list a pass with one-pass images of `min_area`-sample tiles may need)
bool pb_conv_wfft_feasible(const ConvPass &p, bool poly2, int min_area) {
    WGeom g; long per_max = 0, pairs12 = 0;
    return wfft_geometry(p, poly2, (float)min_area, g, per_max, pairs12);
}

bool pb_conv_wfft_types(const ConvPass &p) {
    switch (p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype) {
        case 0: case 1: case 3: case 4: case 12: case 13: case 24: case 26: case 6: case 8: case 2: return true;
        default: return false;
    }
}

// PB_ERR_UNSUPPORTED: dtype combination not built (the caller falls back to the workgroup form).  Every pass whose types
// are built comes here whatever its size, so that what an image gets does not depend on the batch it travels in (the two
// forms round differently: this one rotates the window rows and has per-axis halos).  A wave takes ~19 us for its pair
// whatever the size of the launch, a 512-thread workgroup ~8 us, so a three-step pass of a few hundred pairs is a race of
// single pairs that the workgroup form used to win (700 x 500: 0.31 against 0.38 ms per call); with small images' polynomials
// mostly one window pass now, the call is faster through this form alone (0.29 ms; 1080p 0.45 against 0.53).
// PB_WAVE_MIN_JOBS=n in the environment sends passes of fewer than n pairs to the workgroup form again.
int pb_launch_conv_wfft(pb_ctx *ctx, const ConvPass &p) {
    const long min_jobs = ctx->wave_min_jobs;
    if (!pb_conv_wfft_types(p)) return PB_ERR_UNSUPPORTED;
    const bool poly2 = p.poly != 0 && pb_spec_of_spectra(ctx, p.khat).on >= 2;
    const float min_area = (float)ctx->poly_min_area;      // (the smallest one-pass tile the cost model of khat.h admits)
    WGeom g;
    long per_max = 0, pairs12 = 0;
    if (!wfft_geometry(p, poly2, min_area, g, per_max, pairs12)) {
        if (poly2) return pb_fail(ctx, PB_ERR_UNSUPPORTED, "conv pass: too many windows for the tile-spectrum body");
        return PB_ERR_UNSUPPORTED;
    }
    long jobs = pairs12 * p.P;
    long groups = 8L * per_max * p.P;                       // list entries if every image had the smallest tiles its records may select
    if (ctx->known_sel) {
        // the host has the records' choices (it built them): the grid is exactly the list of this launch's jobs
        const int B = p.P / p.C;
        long per_sum = 0;
        jobs = 0;
        for (int b = 0; b < B && b < (int)ctx->known_sel->size(); ++b) {
            const pb_fft_sel &e = (*ctx->known_sel)[(size_t)b];
            const bool takes = e.use_fft && e.poly != 2 && (p.poly == 2 || (e.poly != 0) == (p.poly != 0));      // (poly_match, conv_fft_common.h)
            if (!takes) continue;
            const int tx = FT_N - 2 * e.hx, ty = FT_N - 2 * e.hy;
            const long nj = (long)(((g.ow + tx - 1) / tx + 1) / 2) * ((g.oh + ty - 1) / ty);
            per_sum += (nj + 7) / 8; jobs += nj * p.C;
        }
        if (!jobs) return PB_OK;                            // nothing in this launch for any image
        groups = 8L * per_sum * p.C;                        // (the kernel's list: the images' shares of every plane, back to back)
    }
    if (!poly2 && jobs < min_jobs) return PB_ERR_UNSUPPORTED;
    ProfScope prof(ctx, PB_PROF_CONV_FFT);
    // fp32 planes; the second and third Horner step of fp16 images (fp32 temporaries in, fp16 x operand, fp32 or fp16 out);
    // the first step and the one-pass polynomial of fp16 images (fp16 window)
    switch (p.in_dtype * 9 + p.x_dtype * 3 + p.out_dtype) {
        case 0: return launch_wfft_typed<float, float, float>(ctx, p, g, groups);
        case 1: return launch_wfft_typed<float, float, __half>(ctx, p, g, groups);          // (the last store of an fp16 image after fp32 iterations)
        case 3: return launch_wfft_typed<float, __half, float>(ctx, p, g, groups);
        case 4: return launch_wfft_typed<float, __half, __half>(ctx, p, g, groups);
        case 12: return launch_wfft_typed<__half, __half, float>(ctx, p, g, groups);
        case 13: return launch_wfft_typed<__half, __half, __half>(ctx, p, g, groups);
        // 8-bit images: the first step (fp32 out) or the one-pass polynomial (fp32 or 8-bit out) from the 8-bit window, the
        // later steps with the 8-bit x operand, the last store
        case 24: return launch_wfft_typed<unsigned char, unsigned char, float>(ctx, p, g, groups);
        case 26: return launch_wfft_typed<unsigned char, unsigned char, unsigned char>(ctx, p, g, groups);
        case 6: return launch_wfft_typed<float, unsigned char, float>(ctx, p, g, groups);
        case 8: return launch_wfft_typed<float, unsigned char, unsigned char>(ctx, p, g, groups);
        case 2: return launch_wfft_typed<float, float, unsigned char>(ctx, p, g, groups);
        default: return PB_ERR_UNSUPPORTED;
    }
}
