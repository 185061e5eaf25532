// Internal declarations shared by the HIP translation units of libpolyblur_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/polyblur_hip.h"


// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
struct FftPlan {
    int n = 0;
    int nstage = 0;
    int radix[24];
    int bluestein_m = 0;          // 0: direct mixed-radix; else the power-of-two length used
    float2 *tw = nullptr;         // W_n^m (or W_m^m for bluestein), m = 0..n-1
    float *drev = nullptr;        // derivative multiplier in digit-reversed order, / n
    // bluestein extras
    float2 *chirp = nullptr;      // w[n] = exp(+i pi n^2 / N), n = 0..N-1
    float2 *bfilt_rev = nullptr;  // FFT_M(b) in digit-reversed order, / M
    float *dnat = nullptr;        // derivative multiplier in natural order, / N
};

struct ScratchBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct ProfRec {
    hipEvent_t a, b;
    int tag;
};

// One-pass polynomial: which filter the spectra in the context's scratch are those of (see pb_fft_sel.poly).
// on: 0 = the kernel's own spectrum (three Horner steps); 1 = the polynomial's for kernels within the 4-sample halo class
// (composite halo class 12: either tile-spectrum body can run it); 2 = the polynomial's wherever one window pass with
// the composite filter's own per-axis halos is cheaper than the three steps (wave body only: conv_wfft.hip); 3 = as 2, and
// on 128 x 128 windows (conv_w128.hip, pb_fft_sel.poly == 2) where THAT is cheapest -- cost128 = what a 128 x 128 window pair
// costs in units of a 64 x 64 one (four waves, longer transforms).
// gain, min_area: the cost model of mode 2 (khat.h).
// always == 2 (with on == 0): the kernel's own spectrum, and EVERY kernel takes the three-step window form whatever its phase
// count or rank -- the ring steps of a zero-boundary polynomial (pb_build_khat_ring), and the whole polynomial of a
// zero-boundary image too small for the ring form (api.hip: poly_spec): three launches of the wave body and nothing else.
// always: EVERY image of the record set takes a one-pass form (64 x 64 or 128 x 128 windows, whichever the cost model prices
// lower) -- the three-step and stencil forms are out of the model.  Asked for where the host can tell from (boundary,
// options, image size, ker_size) alone that every composite fits a 128 x 128 window and every kernel is a point-symmetric
// Gaussian the estimation itself builds: the polynomial then issues the two window launches and nothing else (no launch
// that finds no work, no side stream), without the host ever reading a record back.
struct PolySpec { int on; float a3, a2, a1, b; float gain; int min_area; float cost128; int always; };
inline PolySpec no_poly() { return PolySpec{0, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 0.f, 0}; }
inline bool same_spec(const PolySpec &x, const PolySpec &y) {
    return x.on == y.on && x.always == y.always &&
           (!x.on || (x.a3 == y.a3 && x.a2 == y.a2 && x.a1 == y.a1 && x.b == y.b && (x.cost128 > 0.f) == (y.cost128 > 0.f)));
}

// rf = window halo class of the workgroup form (conv_fft.hip): 4, 8 or 12 -- 0 when only the wave form can run the image
// (a composite halo beyond 12); hx, hy = the wave form's window halo per axis (conv_wfft.hip): hx a multiple of 4 (windows stay
// on 16-byte boundaries), hy even; strip: rank-1 kernel of full support (conv_strip.hip may take it).
// poly: the image's spectrum is that of the WHOLE polynomial a3 K^3 + a2 K^2 + a1 K + b (the reference's own 'fft' form,
// deblurring.py:139-169) and the halos those of that composite filter: one window pass (ConvPass.poly = 1 or 2) replaces
// the image's three Horner steps, whose launches skip it.
struct pb_fft_sel { int use_fft; int rf; int strip; int poly; int hx; int hy; int pad_[2]; };

struct pb_ctx {
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> evpool;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, ScratchBuf> scratch;
    std::map<int, FftPlan> plans;
    float *interp_w = nullptr;     // (n_interp x (n_angles+1)) Keys weights
    int interp_na = 0, interp_ni = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_switch = nullptr;
    // A second stream for the launches of a polynomial that device-built records may leave without work (pb_launch_conv_poly):
    // forked from and joined back into `stream` inside the call, idle between calls.  nullptr: its creation failed (the engine works without it).
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    size_t est_done_bytes = 0;     // size of the zero-initialised arrival counters
    // dense (non rank-1) kernels with at least this many live stencil phases are evaluated per tile in the frequency
    // domain (conv_fft.hip) instead of by the stencil body; < 0: never (pb_set_dense_eval, env PB_DENSE_EVAL)
    int fft_min_phases = 16;
    // which tile-spectrum body evaluates them: 1 = one wave per window pair (conv_wfft.hip), 0 = one workgroup per
    // window pair (conv_fft.hip); env PB_FFT_BODY=wg|wave
    int fft_wave = 1;
    // What the host knows about record sets it built itself and read back (pb_make_kernels / pb_set_kernels synchronise
    // anyway): whether any image takes the tile-spectrum body, whether any takes a stencil body -- a reblurring pass then
    // skips the launch nobody needs -- and whose spectra the context's scratch currently holds.  Records estimated on the
    // device (the pipeline) are never in here: their passes issue both launches.
    // any_other = any_strip || any_tile (by body of the fp32 pass); any_fft3 = some image takes the tile-spectrum body step by step.
    // The first set is the choice of body without the one-pass polynomial (every single pass); `poly` = the choice under the
    // PolySpec `spec` (read back the first time a polynomial with that spec meets these records; sel = the device's records)
    struct BodyFlags { bool any_fft = false, any_other = false, any_strip = false, any_tile = false, any_fft3 = false; };
    struct RecFlags { int B; BodyFlags plain; bool poly_valid; PolySpec spec; BodyFlags poly; std::vector<pb_fft_sel> sel; };
    std::map<const void *, RecFlags> rec_cache;
    // Caller-supplied taps that are not point-symmetric (pb_set_kernels): the records hold them as CORRELATION taps -- what
    // F.conv2d applies, filters.py:40-49, method='direct' --, while the reference's method='fft' is a true circular convolution
    // (filters.py:33-36: K = p2o(kernel)), i.e. the point-reflected taps.  The wrap-boundary stage entry points run such records
    // through a reflected copy (api.hip: wrap_records); the host keeps the reflected taps for that.  (The pipeline's own kernels
    // are point-symmetric, or -- even ker_size -- placed per method by the parameter kernel: nothing to reflect there.)
    struct FlipSet { int B; int support; std::vector<float> taps; };
    std::map<const void *, FlipSet> flip_sets;
    int sel_slot = 0, sel_last = 0, sel_B = 0;           // "conv.fftsel" holds PB_SEL_SLOTS runs of sel_B records: the slot passes write to / read from
    const std::vector<pb_fft_sel> *known_sel = nullptr;   // the records of the pass being launched, where the host has them (sizes its job grid)
    const void *khat_owner = nullptr;    // record set whose spectra "conv.khat" holds (nullptr: unknown)
    int khat_slot = 0;                   // ... and the slot of "conv.fftsel" their selections were written to
    int khat_B = 0;                      // ... and how many of its records they cover (a longer run at the same address has stale tails)
    const void *khat_buf = nullptr;
    bool khat_by_estimate = false;
    // one-pass polynomial experiment (env PB_POLY1=1): what the spectra in "conv.khat" were built for, and what the call
    // in progress wants (set around the estimation / the polynomial; off for every other pass)
    // 0 never; 1 every eligible polynomial, host-built records included (those are then not cached); 2 (default) the
    // pipeline's: mildly blurred images -- the method's own use case -- estimate kernels like sigma 0.6 / rho 0.3 at an
    // oblique angle or the clamped isotropic 0.3 / 0.3, within the 4-sample halo (under the adaptive policy; the latter
    // under full support too, its other taps underflow), and their polynomial is then 1.7 x faster.  Where the first step's
    // launch can take such images along (same output type as the last step's) that costs nothing when no image
    // qualifies; otherwise a composite launch is needed, which finds no work then and costs 1 % of a 4K call even on the
    // side stream (1.182 -> 1.195 ms): issued under the adaptive policy only.
    int poly_mode = 3;
    PolySpec poly_built = no_poly(), poly_want = no_poly();
    // A SECOND set of spectra + selections ("conv.khat2" / "conv.fftsel2"), built by the estimation beside the first where the
    // call will need both -- the zero boundary's ring steps want the kernels' own spectra next to the polynomial's, the
    // polynomial behind an edgetaper wants its own next to the kernels' -- instead of by a khat_kernel launch of its own
    // between the estimation and the pass (measured: 23 us per iteration of a 4K method='direct' call, 9 us with edgetaping).
    // poly_want2: what the estimation in progress is asked to build there (on == 0 && always == 0: nothing).
    PolySpec poly_want2 = no_poly(), khat2_spec = no_poly();
    const void *khat2_owner = nullptr;   // record set whose spectra the second set holds (nullptr: none)
    int khat2_B = 0;
    const void *khat2_buf = nullptr;
    unsigned sel2_mask = 0;              // iterations (slots of "conv.fftsel2", as of "conv.fftsel") whose polynomial read the SECOND set's selections: pb_body_selection reports those
    // cost model of the general one-pass form (PolySpec.on == 2; env PB_POLY_GAIN, PB_POLY_MIN_AREA): an image takes it when
    // its composite tile has at least poly_min_area samples and -- an image that would otherwise take three tile-spectrum
    // passes -- 3 x that area is at least poly_gain x the tile area of its three-step windows (cost per output sample, in
    // window pairs: 3 / (poly_gain x three-step tile area) against 1 / composite tile area).  A Horner step costs more than a
    // one-pass window of the same tile -- the x operand, 8 words per sample instead of 2, two launches more: measured 0.254 ms
    // for three steps at 40 x 40 tiles against 0.153 ms for one pass on 128 x 128 windows at 64 x 68 tiles (4K), 0.092 against
    // 0.050 at 1080p -- hence 0.7 (swept on 32 x 1080p: 1.0 -> 6.61, 0.85 -> 6.26, 0.7 -> 6.14, 0.6 -> 6.15 ms per step); one
    // pass over 768-sample tiles takes what three rank-1 stencil passes take
    float poly_gain = 0.7f;
    int poly_min_area = 768;
    long poly_min_pairs128 = 1;          // env PB_POLY_MIN_PAIRS128: 128 x 128 windows only for images of at least this many window pairs (at 90 x 90 tiles, all channels).  150 in round 4 (a 700 x 500 image was slower with them: 0.256 -> 0.279 ms); 1 since the one-pass class removed the other launches (round 5, same box: 700 x 500 0.242 -> 0.211 ms per call, 256 x 256 0.222 -> 0.192, 1080p gray 0.431 -> 0.319, 16 x 700 x 500 1.14 -> 0.72): every size is in the class
    float poly_cost128 = 8.0f;           // env PB_POLY_COST128; <= 0: never 128 x 128 windows.  Measured at 4K: a 128 x 128 pair costs 6 - 6.5 pairs of
                                         // 64 x 64 with host-built records, and a launch of its own (~10 us) in the pipeline
    int zero_ring_aside = 1;             // env PB_ZERO_RING_ASIDE: 0 = the ring steps of a zero-boundary polynomial all behind its window pass on the caller's stream
    long zero_ring_min_pairs = 4096;     // env PB_ZERO_RING_MIN_PAIRS: ... only for images of at least this many three-step window pairs (40 x 40 tiles, all channels: two rounds of the chip)
    int zero_ring = 1;                   // env PB_ZERO_RING: 0 = a polynomial under the zero boundary (method='direct') keeps three Horner steps over the whole image
    int taper_ring = 1;                  // env PB_TAPER_RING: 0 = every blend of an edgetaper covers the whole padded plane
    int poly_padded = 1;                 // env PB_POLY_PADDED: 0 = a polynomial whose operand is a padded plane (after an edgetaper) keeps three Horner steps
    int est_lean = 1;                    // env PB_EST_LEAN: 0 = the parameter kernel always forms the whole record before the spectra
    int dt_cols_strip = 1;               // env PB_DT_COLS_STRIP: 0 = the domain-transform column pass as two sweeps through global memory (dt_cols_fused_kernel), 2 = strips whose up sweep always forms the weights from J again (dt_cols_up_kernel)
    int dt_cols_coop = 1;                // env PB_DT_COLS_COOP: 0 = never the few-columns form of the column pass (dt_cols_coop_kernel), 2 = always where it applies
    int dt_rows_reg = 1;                 // env PB_DT_ROWS_REG: 0 = the domain-transform row pass always through global memory (dt_rows_fused_kernel), 2 = registers, one wave per row always, 3 = a row over four waves always (dt_rows_regw_kernel)
    int poly_always = 1;                 // env PB_POLY_ALWAYS: 0 = never PolySpec.always (every polynomial issues all the launches its records might need)
    long side_min_tiles = 12288;         // (PB_SIDE_MIN_TILES until round 6) stencil tiles per launch from which the launches that may find no work go to the side stream
    int main_stream_body = -1;           // (PB_MAIN_STREAM_BODY until round 6) which launch stays on the caller's stream when the others go to the side stream (0 = wave body, 1 = 128 x 128; -1 = by spec)
    int est_gray_rows = 1;               // env PB_EST_GRAY_ROWS: 1 = gray + range + row transform in one launch where measured faster (fp32 planes, lines of up to 4096 samples), 2 = for any line held in LDS, 0 = never
    int fft_ext_radix = 1;               // env PB_FFT_EXT_RADIX: 0 = greedy plans only (radices up to 16)
    int fft_first_rows = -1;             // env PB_FFT_FIRST_ROWS: the same for the row transforms (rows_plan)
    int fft_first = -1;                  // env PB_FFT_FIRST: the radix of the column transform's first / last stage where the plan holds it; 0 = the plan's own order; -1 = chosen by trips (launch_cols)
    // tuning / comparison knobs of single kernels, read once in pb_create (the table of every knob: api.hip, pb_read_knobs)
    long wave_min_jobs = 0;              // env PB_WAVE_MIN_JOBS: three-step passes of fewer window pairs than this go to the workgroup form of the tile-spectrum body
    int fft_lognb = -1;                  // env PB_FFT_LOGNB: log2 of the complex lines per column workgroup (-1: by LDS size)
    int rows_fixed = 1;                  // env PB_ROWS_FIXED: the same for the row transforms (gray_rows_kernel / grad_rows_kernel)
    int cols_fixed = 1;                  // env PB_COLS_FIXED: 0 = the column transform always by the run-time-plan kernel (grad_cols_kernel), also where lines_fixed.hip holds the plan
    int cols_wide = 1;                   // (PB_COLS_WIDE until round 6) 0 = never the double-width column tile
    int rows_nt = 0;                     // (PB_ROWS_NT until round 6) threads per row workgroup (128 / 256 / 512; 0 = by line length and grid size)
    int xt_two_launch = 0;               // env PB_XT=2: the x-t approximation as two launches of the general body
    int est_overlap = -1;                // env PB_EST_OVERLAP (--experimental builds): rows and columns side by side on two streams
    int strip_seg = 0;                   // env PB_STRIP_SEG (--experimental builds): segment height of the strip body
    int strip_mode = 0;                  // env PB_STRIP: 1 = rank-1 kernels of full support take the streaming strip body (fp32 planes; --experimental builds only)
};

int pb_fail(pb_ctx *ctx, int code, const char *fmt, ...);

// RAII bracket: records an event pair around the launches issued in its scope when profiling is on
struct ProfScope {
    pb_ctx *ctx;
    int idx = -1;
    ProfScope(pb_ctx *c, int tag) : ctx(c) {
        if (!c->prof_on) return;
        ProfRec r;
        r.tag = tag;
        hipEvent_t *ev[2] = {&r.a, &r.b};
        for (auto e : ev) {
            if (!c->evpool.empty()) { *e = c->evpool.back(); c->evpool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) return;
        }
        (void)hipEventRecord(r.a, c->stream);
        c->prof.push_back(r);
        idx = (int)c->prof.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(ctx->prof[idx].b, ctx->stream);
    }
};
void *pb_scratch(pb_ctx *ctx, const char *name, size_t bytes);   // nullptr on failure (error set)
const FftPlan *pb_get_plan(pb_ctx *ctx, int n, bool ext_radices = false, int first = 0);   // ext_radices: the caller's kernel holds radices 18 / 20 / 24; first: the radix (one of the plan's) of the stages that talk to global memory, 0 = the plan's own order
const float *pb_get_interp_weights(pb_ctx *ctx, int n_angles, int n_interp);

#define PB_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return pb_fail(ctx, PB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                               \
    } while (0)

#define PB_LAUNCH_CHECK()                                                                     \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess)                                                                 \
            return pb_fail(ctx, PB_ERR_HIP, "kernel launch failed: %s (%s:%d)",               \
                           hipGetErrorString(e_), __FILE__, __LINE__);                        \
    } while (0)

// ------------------------------------------------------------------------------------
// stencil pass (conv.hip)
// ------------------------------------------------------------------------------------
enum { SRC_VIRTUAL = 0,   // an un-padded H x W plane addressed in padded coordinates (replicate pad by index clamp)
       SRC_PADDED = 1 };  // a materialised Hp x Wp plane
enum { OUT_INTERIOR = 0,  // write the H x W crop into an un-padded plane
       OUT_PADDED = 1 };  // write the whole Hp x Wp domain
enum { EPI_HORNER = 0,    // out = scale * (K*in) + coef * x   [+ clamp]
       EPI_TAPER = 1 };   // out = a * x + (1-a) * (K*in),  a = v1[py] * v2[px]

// per image: which body evaluates a dense kernel (written on the device by khat_kernel, conv_fft.hip)
constexpr int PB_SEL_SLOTS = 16;
constexpr int PB_KHAT_STRIDE = 128 * 128;                  // floats of spectrum per image in "conv.khat": 64 x 64 of them, or 128 x 128 (one pass on 128 x 128 windows)
constexpr int PB_POLY128_MIN_T = 56;                       // smallest tile side of a one-pass 128 x 128 window (bounds the job grid): composite halos up to 36 = 3 x 12, every composite there is
constexpr int PB_POLY_MIN_TX = 24, PB_POLY_MIN_TY = 16;     // smallest tile of a one-pass window (bounds the job grid)


struct ConvPass {
    const void *in;  int in_kind;  int in_dtype;  int in_pitch;  long in_plane;
    const void *x;   int x_kind;   int x_dtype;   int x_pitch;   long x_plane;
    void *out;       int out_kind; int out_dtype; int out_pitch; long out_plane;
    int H, W;            // un-padded size; Hp = H + 2*pad, Wp = W + 2*pad
    int pad;             // replicate pad = ker_size / 2 (utils.py:48-53): 12 for the reference's default 25x25 kernel
    int C;               // planes per image
    int P;               // number of planes (B*C)
    const pb_blur_info *info;
    float scale, coef;
    int boundary;
    int epilogue;
    int clamp01;
    int skip_sep;        // rank-1 images are handled by another launch of this step: their tiles exit at once
    int skip_general;    // non-rank-1 images are handled by another launch of this step (conv_xt.hip)
    // tile-spectrum body (conv_fft.hip): per-image selection and kernel spectra (context scratch), filled in by
    // pb_launch_conv; the caller sets khat_ready when an earlier pass on the same stream built them from the same records
    const pb_fft_sel *fsel;
    const float *khat;
    int khat_ready;
    int strip;           // rank-1 images of full support are done by conv_strip.hip's launch of this step: their tiles exit at once
    int poly;            // one-pass polynomial (pb_fft_sel.poly): 0 = those images are skipped (another launch does them); 1 = the
                         // composite pass: only those images; 2 = the first step's launch takes them along -- for them it IS the
                         // composite pass, written to out2 with scale 1, coef 0 and clamp2 (same output type as `out`)
    void *out2;  int out2_kind;  int out2_pitch;  long out2_plane;  int clamp2;
    int no_fft;          // this pass keeps the stencil bodies (pb_launch_conv_poly: some step of the polynomial does not suit the other)
    int ring;            // 0: every tile.  4 / 5: the third / second blend of an edgetaper over the ring its weights differ from 1 on (api.hip:
                         // run_edgetaper).  1 / 2 / 3: Horner step 1 / 2 / 3 of the BORDER RING of a zero-boundary polynomial whose interior
                         // one window pass has done (pb_launch_conv_poly): only the window pairs the frame of outputs within 24 samples of
                         // the padded border depends on (conv_wfft.hip: ring_live)
};

int pb_launch_conv(pb_ctx *ctx, const ConvPass &p);
// the three Horner steps of one polynomial (same planes, records and x operand; step s reads what step s - 1 wrote)
int pb_launch_conv_poly(pb_ctx *ctx, const ConvPass *steps);
int pb_launch_conv_xt(pb_ctx *ctx, const ConvPass &p);                       // conv_xt.hip
int pb_build_khat(pb_ctx *ctx, const pb_blur_info *info, int B, float **khat, pb_fft_sel **sel, bool launch);
int pb_build_khat_ring(pb_ctx *ctx, const pb_blur_info *info, int B, float **khat, pb_fft_sel **sel);   // conv_fft.hip: the kernels' OWN spectra, every symmetric kernel on three-step windows, in a second scratch set
int pb_khat_buffers(pb_ctx *ctx, int B, float **khat, pb_fft_sel **sel);   // the scratch alone (the estimation fills it itself)
int pb_khat2_buffers(pb_ctx *ctx, int B, float **khat, pb_fft_sel **sel);  // ... of the second set
// the spec under which the spectra a pass reads were built: the second set's where the pass reads that set (the polynomial behind
// an edgetaper, whose FIRST set holds the blends' kernels -- job lists sized from the first set's spec were too short there)
inline const PolySpec &pb_spec_of_spectra(const pb_ctx *ctx, const void *khat) {
    return khat && khat == ctx->khat2_buf ? ctx->khat2_spec : ctx->poly_built;
}
int pb_cache_records(pb_ctx *ctx, const pb_blur_info *info, int B);        // conv.hip: after the host (re)built these records
void pb_forget_records(pb_ctx *ctx, const void *info, int B);                 // B records at info are about to be rewritten; nullptr: all
void pb_forget_range(pb_ctx *ctx, const void *dst, size_t bytes);            // a host write into device memory
int pb_launch_conv_fft(pb_ctx *ctx, const ConvPass &p);
int pb_launch_conv_strip(pb_ctx *ctx, const ConvPass &p);                    // conv_strip.hip; PB_ERR_UNSUPPORTED: not an all-fp32 plain Horner pass
int pb_launch_conv_wfft(pb_ctx *ctx, const ConvPass &p);                     // conv_wfft.hip; PB_ERR_UNSUPPORTED: dtype combination not built
bool pb_conv_fft_types(const ConvPass &p);                                   // conv_fft.hip: whether the workgroup form is built for the pass's types
bool pb_conv_wfft_types(const ConvPass &p);
int pb_launch_conv_w128(pb_ctx *ctx, const ConvPass &p);                     // conv_w128.hip: the one-pass polynomial on 128 x 128 windows (pb_fft_sel.poly == 2)
bool pb_conv_w128_feasible(const ConvPass &p);                               // ... and whether its worst-case job list fits the grid
bool pb_conv_wfft_feasible(const ConvPass &p, bool poly2, int min_area);     // conv_wfft.hip: likewise for the wave form
bool pb_conv_w128_types(int in_dtype, int out_dtype);                                  // ... whether it is built for the pass's types
bool pb_poly_three_steps_ok(pb_ctx *ctx, const ConvPass *steps);             // conv.hip: the three steps can all take the wave form (PolySpec.always == 2)
int pb_poly_spec_mode(pb_ctx *ctx, const ConvPass *steps);                   // conv.hip: the PolySpec.on a polynomial with these steps may ask for
// kernels larger than the 25 x 25 record (conv_big.hip): their taps on the ker_size grid, and one Horner step with them
int pb_build_big_taps(pb_ctx *ctx, const pb_blur_info *dev_info, int B, int ksize, int shift, const float **taps);
int pb_launch_conv_big(pb_ctx *ctx, const ConvPass &p, const float *taps, int ksize);
constexpr int PB_KSIZE_MAX = 49;
bool pb_conv_fft_feasible(const ConvPass &p);                                // window counts within the kernel's index arithmetic

// ------------------------------------------------------------------------------------
// estimation (estimate.hip)
// ------------------------------------------------------------------------------------
int pb_estimate_impl(pb_ctx *ctx, const void *in, int dtype, int B, int C, int H, int W,
                     const pb_options *opt, pb_blur_info *dev_info);
int pb_make_sep_records(pb_ctx *ctx, int B, const pb_blur_info *dev_info, pb_blur_info *sep, int support, int ksize);
int pb_kernel_size(const pb_options *opt);      // validated ker_size (2 .. PB_KSIZE_MAX; above PB_KSIZE: conv_big.hip's path); 0 if unsupported
int pb_fourier_gradients_impl(pb_ctx *ctx, const float *planes, int P, int H, int W, float *gx, float *gy);
int pb_fourier_gradients_typed(pb_ctx *ctx, const float *planes, int P, int H, int W, void *gx, void *gy, int out_dtype);   // PB_F16: __half planes out
bool pb_gradient_planes_half(pb_ctx *ctx, int H, int W);       // whether this context's line transforms can write __half planes for H x W
int pb_make_kernels_dev(pb_ctx *ctx, int B, pb_blur_info *dev_info, int support, int from_taps, int ksize = PB_KSIZE);

// ------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float pb_ld(const T *p);
template <> __device__ __forceinline__ float pb_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float pb_ld<__half>(const __half *p) { return __half2float(*p); }
// (the conversions are never contracted into a neighbouring FMA, so the fused edge computes exactly what the
// float pipeline computes between host-side conversions)
// 8-bit pixels: img_as_float32 on load, img_as_ubyte on store (skimage 0.19.2 util/dtype.py _convert)
__device__ __forceinline__ float pb_from_ubyte(unsigned u) {
    float r = (float)u * (1.0f / 255.0f);
    asm("" : "+v"(r));                   // keeps the product from being contracted into a neighbouring FMA (not volatile:
                                         // volatile statements keep their order, and with it the loads that feed them one
                                         // behind the other -- 128 byte loads of a window column, each waited for)
    return r;
}
template <> __device__ __forceinline__ float pb_ld<unsigned char>(const unsigned char *p) { return pb_from_ubyte(*p); }
template <typename T> __device__ __forceinline__ void pb_st(T *p, float v);
__device__ __forceinline__ unsigned pb_to_ubyte(float v) { return (unsigned)__float2int_rn(fminf(fmaxf(__fmul_rn(v, 255.f), 0.f), 255.f)); }
template <> __device__ __forceinline__ void pb_st<unsigned char>(unsigned char *p, float v) { *p = (unsigned char)pb_to_ubyte(v); }
template <> __device__ __forceinline__ void pb_st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void pb_st<__half>(__half *p, float v) { *p = __float2half_rn(v); }

// order-preserving float <-> uint encoding for atomicMin/atomicMax on floats
__device__ __forceinline__ unsigned pb_f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pb_ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
