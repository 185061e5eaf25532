#!/usr/bin/env python3
"""bench.py -- megapixels/s of the Polyblur hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--mode resident|from_root]

One "step" = one full polyblur_deblurring() call (n_iter=3, alpha=6, beta=1, c=0.362, b=0.468, q=0,
method='fft', full 25-tap support) on one resident batch.  Default workload: BASELINE config 2, a single
4K (3840x2160x3) fp32 image per GPU.  The other configs of BASELINE.json are selectable (per-GPU shares):
cfg3 = 64 x 1080p fp16 + halo removal + domain-transform prefilter, cfg4 = 32 x 1080p fp32 per GPU,
cfg5 = 1 x 8K fp16 per GPU with n_iter=5.

N > 1: one process per GPU over RCCL.  The driver launches the ranks with torch.distributed.run; when
bench.py is started directly with --gpus N > 1 (no RANK in the environment) it re-launches itself the same
way.  --mode resident (default, "scaling": "weak"): every rank owns its shard already (images shard with no
data-path collective); --mode from_root: the whole batch lives on rank 0 and every step is scatter + compute
+ gather (polyblur_amd.distributed.deblur_from_root, grouped RCCL point-to-point over xGMI).  At N > 1 the
resident run also carries a short from_root measurement in `context` (report both).  Either way
the timed region is bracketed by a barrier + device synchronise on both sides and the maximum over ranks is
reported.

Between the W warm-up steps and the K timed steps the device gets ~60 ms of the same (untimed) steps to settle its clocks
(`settle_steps` in the line): the timed region of a 4K run is 10 ms, and without them its first steps run 5 % slow.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      -- the dominant kernel class (the reblurring pass), from hipEvents on the engine's stream around every
                   launch of a second, identical run of the K steps (the headline `value` is timed without them).
                   `achieved` / `frac` are SURVEY 8d's ALGORITHMIC bytes (8 words per sample and polynomial, whatever
                   the form) per launch / average launch duration -- the contract's figure, an equivalent rate;
                   `hbm_min` = the bytes the forms that ran HAVE to move (2 words for a one-pass polynomial, 3 per
                   iteration end to end) over the same times: the physical HBM fraction; `issue` = vector
                   instructions issued per SIMD cycle (rocprofv3 PMC passes, profiles/); `bound` says which of them
                   the kernel is near
  parity        -- the step's output against the oracle on the same image (N == 1): max-abs difference and
                   whether the per-iteration theta sequences are identical; the run FAILS above tolerance
  cpu_baseline  -- the NumPy oracle (a port of the reference's CPU path) timed on this box's host cores on
                   bounded samples (rank 0, N == 1 only): 1 thread and all cores, 700x500 / 1080p / 4K
  stages_ms_per_step, context -- labelled side numbers, never the headline value; at N = 1 on the default config `context`
                   also carries the other BASELINE configs (cfg3, cfg4_share, cfg5_share: 1 warm-up + 3 timed steps each,
                   image 0 against the oracle for cfg3 / cfg4; a parity failure there fails the run too).
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
VALU_PEAK_TFLOPS = 157.3      # fp32 packed FMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    "cfg2": dict(batch=1, height=2160, width=3840, dtype="f32", n_iter=3, opts={}),
    "cfg3": dict(batch=64, height=1080, width=1920, dtype="f16", n_iter=3,
                 opts=dict(remove_halo=True, prefiltering=True, prefilter="domain_transform")),
    "cfg4": dict(batch=32, height=1080, width=1920, dtype="f32", n_iter=3, opts={}),
    "cfg5": dict(batch=1, height=4320, width=7680, dtype="f16", n_iter=5, opts={}),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="resident", choices=["resident", "from_root"])
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--dtype", default=None, choices=["f32", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the labelled side measurements, so that a profiler sees only the headline workload's kernels")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed steps for this long between the warm-up and the timed region (clocks settle); 0 under a profiler, whose passes must see the same launches")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU check of the N-rank launch path: rendezvous over gloo, all-reduce, print the line, no compute")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` started by hand: become N ranks, one per GPU, the way the driver starts them."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def make_batch(b, h, w, seed0):
    """Synthetic (b,3,h,w) batch.  Distinct images up to 4, then tiled: generation is
    CPU-side NumPy and would otherwise dominate start-up for the large batches."""
    from polyblur_amd.synthetic import synthetic_blurry_batch
    nd = min(b, 4)
    x, params = synthetic_blurry_batch(nd, 3, h, w, seed0=seed0)
    if b > nd:
        x = np.concatenate([x] * ((b + nd - 1) // nd))[:b]
    return x, params


def cpu_baseline_samples(x_np, kw, want_4k_result):
    """The oracle on this box's host cores: 700x500 (config 1's size), 1080p and the 4K headline image, 1 thread
    and all cores (scipy.fft workers; NumPy's elementwise passes stay single-threaded).  Returns (samples, the
    4K single-thread result and records or None)."""
    from oracle import polyblur_ref as ref          # checker / baseline only
    from polyblur_amd.synthetic import synthetic_blurry_batch
    cores = os.cpu_count() or 1
    H, W = x_np.shape[-2:]
    cases = [("700x500", synthetic_blurry_batch(1, 3, 500, 700, seed0=31)[0]),
             ("1920x1080", synthetic_blurry_batch(1, 3, 1080, 1920, seed0=32)[0]),
             ("%dx%d" % (W, H), np.ascontiguousarray(x_np[:1]).astype(np.float32))]
    samples, res4k = [], None
    for name, img in cases:
        for threads in (1, cores):
            ref.set_fft_workers(None if threads == 1 else threads)
            t0 = time.perf_counter()
            r = ref.polyblur_deblurring(img, method="fft", return_info=True, **kw)
            dt = time.perf_counter() - t0
            ref.set_fft_workers(None)
            samples.append(dict(image=name, threads=threads, seconds=round(dt, 3),
                                mp_per_s=round(img.shape[-1] * img.shape[-2] / 1e6 / dt, 3)))
            if img is cases[-1][1] and threads == 1 and want_4k_result:
                res4k = r
    return samples, res4k


def forms_per_iteration(eng, B, n_iter):
    """How each iteration's polynomial was evaluated (pb_body_selection: the device's choice, iteration by iteration)."""
    per_it = []
    for k in range(n_iter):
        sel = eng.body_selection(B, k)
        one = sel[(sel[:, 0] == 1) & (sel[:, 3] != 0)]
        three = sel[(sel[:, 0] == 1) & (sel[:, 3] == 0)]
        halos = lambda t: sorted({(int(r[4]), int(r[5])) for r in t})[:4]
        per_it.append(dict(one_pass_images=int(len(one)), one_pass_on_128x128_windows=int((one[:, 3] == 2).sum()),
                           one_pass_halos_xy=halos(one), three_step_images=int(len(three)),
                           three_step_halos_xy=halos(three), stencil_images=int((sel[:, 0] == 0).sum())))
    return per_it


def moved_bytes(per_it, opts, s, H, W):
    """Bytes the forms that ran HAVE to move through HBM: (polynomial launches, whole call).  A one-pass polynomial reads x
    and writes y: 2 words per sample; a three-step one SURVEY 8d's 8.  End to end one more read of the image per iteration
    (the estimation).  Word sizes are the stored types: the caller's at either end of the call, fp32 between iterations (and
    around the options); the options' own stages are not counted."""
    n_it, opts_on = len(per_it), bool(opts)
    moved_poly = moved_e2e = 0.0
    for k, pi in enumerate(per_it):
        rd = s if (k == 0 and not opts.get("prefiltering")) else 4
        wr = s if (k == n_it - 1 and not opts_on) else 4
        one_n, other_n = pi.get("one_pass_images", 0), pi.get("three_step_images", 0) + pi.get("stencil_images", 0)
        per_img = 3 * H * W
        moved_poly += per_img * (one_n * (rd + wr) + other_n * (3 * rd + 16 + wr))      # three steps: x read 3 times, t1 / t2 (fp32) written and read, y
        moved_e2e += per_img * (one_n + other_n) * (s if k == 0 else 4)                 # the estimation's read of the image
    return moved_poly, moved_e2e + moved_poly


def survey_words(opts):
    """9 words per sample and iteration, + 3 halo masking, + 5 domain-transform prefilter (+ 2 bilateral)."""
    words = 9.0
    if opts.get("remove_halo"):
        words += 3.0
    if opts.get("prefiltering"):
        words += 5.0 if opts.get("prefilter") == "domain_transform" else 2.0
    return words


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    cfg = dict(CONFIGS[args.config])
    for k in ("batch", "height", "width", "dtype"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.selftest_launcher:
        if use_dist:
            dist.init_process_group("gloo")
        ones = torch.ones(1)
        if use_dist:
            dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"selftest_launcher": True, "n_gpus": world, "gpus_requested": args.gpus,
                              "rccl_ranks": int(ones.item()), "backend": "gloo"}), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    from polyblur_amd import polyblur_deblurring
    from polyblur_amd import _capi as capi
    from polyblur_amd.distributed import deblur_from_root
    from polyblur_amd.engine import get_engine
    from polyblur_amd.synthetic import DEFAULT_SEED

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.gpus != world:
        raise SystemExit("--gpus %d but %d rank(s) were launched" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                            # RCCL is really up, on every rank
        rccl_ranks = int(ones.item())

    B, H, W = cfg["batch"], cfg["height"], cfg["width"]
    kw = dict(KW, n_iter=cfg["n_iter"], **cfg["opts"])
    tdt = torch.float32 if cfg["dtype"] == "f32" else torch.float16
    s = 4 if cfg["dtype"] == "f32" else 2
    eng = get_engine(local_rank)
    from_root = args.mode == "from_root"
    if from_root:
        # the whole job's batch on rank 0; every step scatters, deblurs and gathers it
        x_np = make_batch(B * world, H, W, DEFAULT_SEED)[0] if rank == 0 else None
        x = torch.from_numpy(x_np).to(dev).to(tdt).contiguous() if rank == 0 else None
        full_shape = (B * world, 3, H, W)

        n_calls = [0]

        def counted(t, **k2):                                  # (this rank's pb_polyblur_batch calls: the roofline's launches are its own)
            n_calls[0] += 1
            return polyblur_deblurring(t, **k2)

        def step(support="full"):
            return deblur_from_root(x, full_shape, tdt, compute=counted, device=dev, support=support, **kw)
    else:
        x_np = make_batch(B, H, W, DEFAULT_SEED + 1000 * rank)[0]
        x = torch.from_numpy(x_np).to(dev).to(tdt).contiguous()

        def step(support="full"):
            return polyblur_deblurring(x, support=support, **kw)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(nsteps, fn):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o = fn()
        sync_all()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # the slowest rank sets the time
            dt = float(t.item())
        return dt, o

    for _ in range(args.warmup):
        out = step()
    # The device's clocks need tens of milliseconds of work to settle (a 4K step is 0.5 ms: behind W = 5 warm-up steps the
    # first timed steps of round 5's line still ran 5 % slow -- VERDICT r5 weak #11, mean against median of the same run).
    # Untimed steps until ~60 ms of work have been issued and finished; how many is in the line (`settle_steps`).
    settle_steps = 0
    if not from_root:
        sync_all()
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < args.settle_ms * 1e-3 and settle_steps < 400:
            for _ in range(4):
                out = step()
            torch.cuda.synchronize(dev)
            settle_steps += 4
        if use_dist:                                                       # every rank settles the same number of steps' worth of time; then all start together
            dist.barrier()
    dt, out = timed(args.steps, step)                                      # the headline: no per-launch events
    ms_per_step = 1e3 * dt / args.steps
    mp_per_step = B * H * W * world / 1e6
    value = mp_per_step / (ms_per_step / 1e3)
    # the same K steps once more with an event pair around every launch: per-kernel-class device times
    if from_root:
        n_calls[0] = 0
    eng.profile_begin()
    dt_prof, _ = timed(args.steps, step)
    prof = eng.profile_end()
    calls_profiled = n_calls[0] if from_root else args.steps
    ms_per_step_prof = 1e3 * dt_prof / args.steps
    # ... and once more with one event between the steps ("median of >= 10 timed runs"): device time per step
    step_ms = []
    if not from_root:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(max(args.steps, 10) + 1)]
        sync_all()
        evs[0].record()
        for e in evs[1:]:
            step()
            e.record()
        sync_all()
        step_ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))

    side = {}
    side_failed = []                         # context configs whose parity check failed (the run then fails too)
    if world > 1 and not from_root and not args.no_context:
        # the same total batch starting and ending on rank 0 (scatter + compute + gather)
        xr = torch.cat([x] * world) if rank == 0 else None
        fr_shape = (B * world, 3, H, W)
        fr = lambda: deblur_from_root(xr, fr_shape, tdt, device=dev, **kw)
        try:                                     # a side measurement: its failure must not take the headline with it
            fr()
            n_fr = max(2, min(args.steps, 5))
            dt_fr, _ = timed(n_fr, fr)
            ms_fr = 1e3 * dt_fr / n_fr
            side["from_root_scatter_compute_gather"] = dict(ms_per_step=round(ms_fr, 4), mp_per_s=round(mp_per_step / (ms_fr * 1e-3), 1),
                                                            note="whole batch on rank 0 before and after; grouped RCCL send/recv per image")
        except Exception as e:                   # noqa: BLE001 -- reported in the line, not raised
            side["from_root_scatter_compute_gather"] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))
        del xr

    if world > 1 and not from_root and not args.no_context and args.config == "cfg2":
        # BASELINE's multi-GPU configs as resident per-GPU shares, so that a scaling run reports them without extra flags:
        # cfg4 = 256 x 1080p fp32 over 8 GPUs (32 per GPU), cfg5 = 8 x 8K fp16, n_iter=5 (one per GPU)
        for name in ("cfg4", "cfg5"):
            c2 = CONFIGS[name]
            try:
                xs = torch.from_numpy(make_batch(c2["batch"], c2["height"], c2["width"], DEFAULT_SEED + 1000 * rank)[0]).to(dev)
                xs = xs.to(torch.float32 if c2["dtype"] == "f32" else torch.float16).contiguous()
                kw2 = dict(KW, n_iter=c2["n_iter"])
                f2 = lambda: polyblur_deblurring(xs, **kw2)
                f2()
                dt2, _ = timed(3, f2)
                ms2 = 1e3 * dt2 / 3
                side[name + "_share_resident"] = dict(ms_per_step=round(ms2, 3), images_per_gpu=c2["batch"], n_iter=c2["n_iter"], dtype=c2["dtype"],
                                                      mp_per_s=round(c2["batch"] * c2["height"] * c2["width"] * world / 1e6 / (ms2 * 1e-3), 1),
                                                      note="whole-job MP/s over %d GPUs, shards resident" % world)
                del xs
            except Exception as e:                   # noqa: BLE001 -- a side measurement
                side[name + "_share_resident"] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))

    if rank != 0:
        if use_dist:
            dist.barrier()                       # leave together with rank 0 (it prints the line first)
            dist.destroy_process_group()
        return 0
    if world > 1:
        args.no_context = True                   # the labelled side measurements below belong to the 1-GPU run

    samples = B * 3 * H * W
    # ---- roofline of the dominant kernel (second, event-bracketed run of the same steps) ---------------
    # the reblurring pass is two launches per Horner step -- the stencil bodies' kernel and the tile-spectrum body's -- in
    # each of which an image's workgroups exit at once unless its record selects that body: the dominant one is judged
    conv_ms, conv_n = prof["conv"]
    dom_kernel = "conv_tile_kernel"
    if prof.get("conv_fft", (0.0, 0))[0] > conv_ms:
        conv_ms, conv_n = prof["conv_fft"]
        # one wave per window pair (csrc/conv_wfft.hip) for every plane type it is built for -- fp32, fp16, 8-bit --; the
        # workgroup form (conv_fft.hip) only on request (PB_FFT_BODY=wg) or for fp16 temporaries
        # (the one-pass polynomial on 128 x 128 windows, conv_w128_kernel, carries the same event tag: one class)
        dom_kernel = "conv_wfft_kernel + conv_w128_kernel" if os.environ.get("PB_FFT_BODY", "wave") != "wg" and not cfg["opts"].get("half_temporaries") else "conv_fft_kernel"
    # one polynomial application = (2s + 3s + 3s) bytes per sample, spread over its launches
    calls_per_step = max(calls_profiled / args.steps, 1e-9)       # from_root deblurs chunk by chunk as they arrive: counted, not assumed
    launches_per_poly = max(conv_n / (args.steps * cfg["n_iter"] * calls_per_step), 1e-9)
    alg_bytes_per_launch = 8.0 * s * (samples / calls_per_step) / launches_per_poly
    conv_avg_ms = conv_ms / max(conv_n, 1)
    achieved = alg_bytes_per_launch / (conv_avg_ms * 1e-3) / 1e9 if conv_n else 0.0
    # HBM-side bytes per launch of the same kernel from the rocprofv3 PMC passes of this same command
    # (tools/profile_bench.sh -> profiles/*_traffic.json; separate runs, FETCH_SIZE x2 on gfx950)
    traffic, traffic_src, cands = None, None, []
    try:
        import glob
        import hashlib
        tname = "bench" if args.config == "cfg2" else "bench_" + args.config
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_traffic.json" % tname)))
        c0 = CONFIGS[args.config]
        if cands and (B, H, W, cfg["dtype"]) == (c0["batch"], c0["height"], c0["width"], c0["dtype"]) and not from_root:
            tj = json.load(open(cands[-1]))
            h = hashlib.sha256()
            for f in sorted(glob.glob(os.path.join(REPO, "polyblur_amd", "csrc", "conv*")) + glob.glob(os.path.join(REPO, "polyblur_amd", "csrc", "khat.h"))):
                h.update(open(f, "rb").read())
            if tj.get("conv_sources_sha256_16") != h.hexdigest()[:16]:
                # counters taken from other code say nothing about this build: no number rather than a stale one
                traffic_src = "%s was taken from other sources of the reblurring pass (git %s): not used" % (os.path.basename(cands[-1]), tj.get("git", "?"))
            else:
                # every instantiation of the dominant kernel that ran, weighted by its launches
                num = den = 0
                doms = ("conv_wfft_kernel<", "conv_w128_kernel<") if dom_kernel.startswith("conv_wfft") else (dom_kernel + "<",)
                for k, v in tj.get("traffic", {}).items():
                    if k.startswith(doms):
                        num += v["hbm_bytes_per_launch"] * v["launches"]; den += v["launches"]
                if den:
                    traffic = int(num / den)
                    traffic_src = "%s (rocprofv3 --pmc passes of this command at git %s, same sources of the pass)" % (os.path.basename(cands[-1]), tj.get("git", "?"))
    except Exception:
        pass
    roofline = dict(bound="hbm", kernel=dom_kernel + " (stencil pass; taps as estimated, full 25x25 support)",
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    basis="SURVEY 8d algorithmic bytes: 8 words per sample and polynomial whatever the form (an EQUIVALENT three-pass "
                          "rate, not bytes moved: see hbm_min for the physical HBM fraction)",
                    traffic=traffic, traffic_source=traffic_src, launches=conv_n, avg_launch_ms=round(conv_avg_ms, 5),
                    launches_per_polynomial=round(launches_per_poly, 3),
                    algorithmic_bytes_per_launch=int(alg_bytes_per_launch),
                    ms_per_step_with_launch_events=round(ms_per_step_prof, 4))
    # A measured ceiling beside the nameplate: plain device copies in this same run -- 1 read + 1 write and
    # 2 reads + 1 write (the shape of a Horner step) of fp32 planes, on the headline's planes (which the 256 MB
    # last-level cache partly holds, as it does for the pass) and on a working set far beyond it
    if not from_root and world == 1:
        def copy_rate(n_elems, reads):
            a = torch.rand(n_elems, device=dev); b = torch.rand(n_elems, device=dev); c = torch.empty(n_elems, device=dev)
            f = (lambda: torch.add(a, b, out=c)) if reads == 2 else (lambda: c.copy_(a))
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize(dev)
            return (reads + 1) * 4.0 * n_elems * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        planes = 3 * 2160 * 3840
        c11, c21 = copy_rate(planes, 1), copy_rate(planes, 2)
        b11, b21 = copy_rate(16 * planes, 1), copy_rate(16 * planes, 2)
        roofline["copy_ceiling_GBps"] = dict(read1_write1=round(b11, 1), read2_write1=round(b21, 1),
                                             read1_write1_headline_planes=round(c11, 1), read2_write1_headline_planes=round(c21, 1),
                                             note="torch device copies in this run; the first two on 16 x the headline's planes (1.6 GB per "
                                                  "operand: HBM), the last two on the planes themselves (99.5 MB per operand)")
        roofline["frac_of_copy"] = round(achieved / b21, 4) if b21 else None
    # whole-step figure of SURVEY 8d: 9 words per sample per iteration (8 for the polynomial + 1 read for the estimate),
    # plus the options' adders: halo masking 3, domain-transform prefilter 4 + 1 for the residual add-back, bilateral 2
    words = survey_words(cfg["opts"])
    e2e_gbs = words * s * samples * cfg["n_iter"] * world / (ms_per_step * 1e-3) / 1e9
    roofline["end_to_end"] = dict(algorithmic_bytes_per_step=int(words * s * samples * cfg["n_iter"]), achieved=round(e2e_gbs, 1),
                                  unit="GB/s", frac=round(e2e_gbs / (HBM_PEAK_GBS * world), 4),
                                  words_per_sample_per_iteration=words,
                                  note="SURVEY 8d: 9 words per sample and iteration, + 3 halo masking, + 5 domain-transform prefilter")
    stages_ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}

    est, infos = None, None
    if not from_root:
        # ---- context: what the estimator found ---------------------------------------------------------
        out_info, infos = polyblur_deblurring(x, return_info=True, **kw)
        est = [dict(theta_deg=round(float(np.rad2deg(i["theta"][0])), 1), sigma=round(float(i["sigma"][0]), 3),
                    rho=round(float(i["rho"][0]), 3), separable=int(i["separable"][0]), radius=int(i["radius"][0]))
               for i in infos]
        # The synthetic blur is oblique (2 of the 30 candidate angles give a rank-1 kernel), so the estimated 25x25
        # kernels are dense and the stencil pass is fp32-VALU-bound, not HBM-bound: say how close to THAT ceiling it
        # runs (multiply-adds actually issued per launch / launch time; peak = 256 CU x 128 lanes x 2 x 2.4 GHz).
        # multiply-adds issued per output sample and pass: rank-1 body 2 (2R+1) with R rounded up to 4/8/12; general body
        # 4 per inner (kernel row, 4-tap segment) phase, 3 per first / last segment of a row (polyblur_hip.h: nphase)
        def issued(r, sp, nph):
            if sp:
                return 2 * (2 * (4 if r <= 4 else (8 if r <= 8 else 12)) + 1)
            return 4 * int(nph[0]) + 3 * (int(nph[1]) + int(nph[2]))
        macs = [sum(issued(int(r), int(sp), nph) for r, sp, nph in zip(i["radius"], i["separable"], i["nphase"])) / B
                for i in infos]                                   # per sample, per pass
        # which body evaluated the dense kernels (pb_set_dense_eval: the default threshold on live stencil phases)
        spectrum = [bool(not sp and int(sum(nph)) >= capi.PB_DENSE_MIN_PHASES)
                    for i in infos for sp, nph in zip(i["separable"], i["nphase"])]
        # how each iteration's polynomial was evaluated (pb_body_selection: the device's choice, iteration by iteration)
        try:
            per_it = forms_per_iteration(eng, B, cfg["n_iter"])
        except Exception as e:                                   # (a label, not a measurement)
            per_it = [dict(error="%s: %s" % (type(e).__name__, str(e)[:120]))]
        if any(spectrum) or any(p.get("one_pass_images") for p in per_it):
            roofline["kernel"] = (dom_kernel + " (the polynomial of an iteration as three launches, one Horner step each, or -- where the "
                                  "whole polynomial's filter fits a 64x64 or a 128x128 window (conv_w128_kernel, same event tag) -- as ONE window pass; "
                                  "taps as estimated, full 25x25 support; evaluated per window in the frequency domain inside registers / LDS)")
            work = [1 if p.get("one_pass_images") and not p.get("three_step_images") and not p.get("stencil_images") else 3 for p in per_it]
            roofline["body"] = dict(tile_spectrum_images=sum(spectrum), of=len(spectrum), per_iteration=per_it,
                                    working_launches_per_polynomial=round(sum(work) / max(len(work), 1), 2),
                                    stencil_multiply_adds_per_sample_it_replaces=round(sum(macs) / len(macs), 1),
                                    note="")
            roofline["body"]["note"] = ("launches_per_polynomial counts every launch issued; under PolySpec.always (wrap boundary, no edgetaper, "
                                        "the estimation's own kernels) a polynomial issues the 128x128 and the 64x64 window launch and nothing "
                                        "else -- the one whose images the other one has finds no work; context.end_to_end_three_step_form is the same "
                                        "call with PB_POLY1=0, context.end_to_end_dense_stencil_body through the 2-D stencil body")
            # ---- what the forms that ran HAVE to move through HBM (the physical fraction) -------------------------------------
            # a one-pass polynomial reads x and writes y: 2 words per sample; a three-step one SURVEY 8d's 8.  End to end one
            # more read of the image per iteration (the estimation): 3 words where every polynomial is one pass.  Word sizes are
            # the stored types: the caller's type at either end of the call, fp32 between iterations (and around the options).
            n_it = cfg["n_iter"]
            moved_poly, moved_e2e = moved_bytes(per_it, cfg["opts"], s, H, W)
            conv_ms_step = conv_ms / args.steps
            roofline["hbm_min"] = dict(
                polynomial=dict(bytes_per_step=int(moved_poly), achieved=round(moved_poly / (conv_ms_step * 1e-3) / 1e9, 1), unit="GB/s",
                                frac=round(moved_poly / (conv_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                note="bytes the reblurring launches have to move (read x, write y per one-pass polynomial) / their summed duration per step"),
                end_to_end=dict(bytes_per_step=int(moved_e2e), achieved=round(moved_e2e * world / (ms_per_step * 1e-3) / 1e9, 1), unit="GB/s",
                                frac=round(moved_e2e * world / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                                note="whole job against n_gpus x 8 TB/s; + one read of the image per iteration for the estimation (gray and its x derivative, one fp32 plane each, "
                                     "are written and read once more: on-chip candidates, not counted); options' stages not counted"))
            ceiling = (roofline.get("copy_ceiling_GBps") or {}).get("read2_write1")
            if ceiling and achieved > ceiling:
                roofline["bound"] = "latency/issue"
                roofline["bound_note"] = ("the SURVEY-8d equivalent rate (%.0f GB/s) is above what a 2-read-1-write copy reaches on this box (%.0f GB/s): the "
                                          "byte model no longer bounds a kernel that moves %.2f of its 8 words -- it is %.0f %% of HBM peak by bytes moved "
                                          "(hbm_min), and what bounds it is the window transforms' dependent chains at 2 waves per SIMD (issue)"
                                          % (achieved, ceiling, 8.0 * moved_poly / max(8.0 * 4 * samples * n_it, 1), 100 * roofline["hbm_min"]["polynomial"]["frac"]))
            # ---- issue side: vector instructions per SIMD cycle, from the SQ pass of profiles/ (same sources) ------------------
            try:
                tj = json.load(open(cands[-1])) if traffic is not None else None
                if tj and tj.get("sq") and tj.get("time"):
                    doms = ("conv_wfft_kernel<", "conv_w128_kernel<")
                    insts = sum(v.get("SQ_INSTS_VALU", 0) * tj["traffic"].get(k, {}).get("launches", 0) for k, v in tj["sq"].items() if k.startswith(doms))
                    ns = sum(v.get("total_ns", 0) for k, v in tj["time"].items() if k.startswith(doms))
                    launches_t = sum(v.get("calls", 0) for k, v in tj["time"].items() if k.startswith(doms))
                    launches_c = sum(tj["traffic"].get(k, {}).get("launches", 0) for k in tj["sq"] if k.startswith(doms))
                    clock = float(tj.get("clock_ghz") or 2.4)
                    if insts and ns and launches_t == launches_c:
                        per_cycle = insts / (ns * 1e-9 * 1024 * clock * 1e9)
                        roofline["issue"] = dict(valu_instructions_per_simd_cycle=round(per_cycle, 4), peak=0.5, frac=round(per_cycle / 0.5, 4),
                                                 valu_instructions_per_launch=int(insts / max(launches_c, 1)), launches_profiled=int(launches_c),
                                                 clock_ghz=clock, source=os.path.basename(cands[-1]),
                                                 note="SQ_INSTS_VALU of the class's launches / (their duration x 1024 SIMDs x clock); a wave64 vector "
                                                      "instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md): peak 0.5 -- packed fp32 "
                                                      "and LDS / memory instructions occupy issue slots beyond this count")
            except Exception:
                pass
        else:
            tflops = 2.0 * samples * (sum(macs) / len(macs)) / (conv_avg_ms * 1e-3) / 1e12 if conv_n else 0.0
            roofline["valu"] = dict(achieved=round(tflops, 1), peak=VALU_PEAK_TFLOPS, unit="TFLOP/s",
                                    frac=round(tflops / VALU_PEAK_TFLOPS, 4),
                                    note="stencil bodies: multiply-adds issued / launch time; "
                                         "context.inner_loop_rank1_* is the HBM-bound separable case")

    def inner_loop(theta_deg, sigma, rho, support, reps=20, e_=None):
        en = e_ or eng
        buf = en.make_kernels([sigma] * B, [rho] * B, [np.deg2rad(np.float32(theta_deg))] * B, support=support,
                              name="bench.info")
        o = torch.empty_like(x)
        en.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ms = en.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32 if s == 4 else capi.PB_F16, x.shape, buf.ptr,
                                 KW["alpha"], KW["beta"], capi.PB_WRAP, reps)
        gbs = 8.0 * s * samples / (ms * 1e-3) / 1e9
        # what the records' form moves: 2 words for a one-pass polynomial (pb_body_selection of the pass just timed), else 8
        try:
            sel = (e_ or eng).body_selection(B, -1)
            words = float(np.mean([2.0 if (r[0] == 1 and r[3] != 0) else 8.0 for r in sel]))
            form = "one window pass" if words == 2.0 else ("three Horner steps" if words == 8.0 else "mixed")
        except Exception:
            words, form = 8.0, "?"
        moved = words * s * samples / (ms * 1e-3) / 1e9
        return dict(ms=round(ms, 4), form=form, equivalent_three_pass_GBps=round(gbs, 1), moved_GBps=round(moved, 1),
                    frac_of_8TBps_by_bytes_moved=round(moved / HBM_PEAK_GBS, 4), mp_per_s=round(B * H * W / 1e6 / (ms * 1e-3), 1))

    if not args.no_context and not from_root:
        side.update({
            # the north-star figure: n_iter's separable-conv inner loop, rank-1 kernels (theta forced to 0)
            "inner_loop_rank1_full_support": inner_loop(0.0, 2.0, 1.0, capi.PB_SUPPORT_FULL),
            "inner_loop_rank1_adaptive_sigma1": inner_loop(0.0, 1.0, 0.6, capi.PB_SUPPORT_ADAPTIVE),
            "inner_loop_general_full_support": inner_loop(30.0, 2.0, 1.0, capi.PB_SUPPORT_FULL),
            "inner_loop_general_adaptive_sigma1": inner_loop(30.0, 1.0, 0.6, capi.PB_SUPPORT_ADAPTIVE),
        })
        # adaptive-support end-to-end (same results to fp32 rounding), labelled
        for _ in range(2):
            step("adaptive")
        dt_ad, _ = timed(args.steps, lambda: step("adaptive"))
        ms_ad = 1e3 * dt_ad / args.steps
        side["end_to_end_adaptive_support"] = dict(ms_per_step=round(ms_ad, 4), mp_per_s=round(B * H * W / 1e6 / (ms_ad * 1e-3), 1))
        # the same call with every polynomial as three Horner launches (a context created with PB_POLY1=0)
        if not cfg["opts"]:
            from polyblur_amd.engine import Engine
            old_env = os.environ.get("PB_POLY1")
            os.environ["PB_POLY1"] = "0"
            try:
                eng3 = Engine(local_rank)
            finally:
                if old_env is None:
                    del os.environ["PB_POLY1"]
                else:
                    os.environ["PB_POLY1"] = old_env
            try:
                eng3.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                o3 = eng3.make_options(n_iter=cfg["n_iter"], **KW)
                out3 = torch.empty_like(x)
                dt3_code = capi.PB_F32 if s == 4 else capi.PB_F16
                f3 = lambda: eng3.polyblur_ptr(x.data_ptr(), out3.data_ptr(), dt3_code, x.shape, o3)
                for _ in range(2):
                    f3()
                dt_3, _ = timed(args.steps, f3)
                ms_3 = 1e3 * dt_3 / args.steps
                side["end_to_end_three_step_form"] = dict(ms_per_step=round(ms_3, 4), mp_per_s=round(B * H * W / 1e6 / (ms_3 * 1e-3), 1),
                                                          max_abs_vs_default=float((out3.float() - out.float()).abs().max()))
                # the north star's LITERAL kernel: separable taps through the rank-1 stencil body with LDS line staging (three
                # launches, one Horner step each -- csrc/conv.hip), which the engine no longer picks for these records
                side["inner_loop_rank1_stencil_body"] = dict(inner_loop(0.0, 2.0, 1.0, capi.PB_SUPPORT_FULL, e_=eng3),
                                                             note="PB_POLY1=0: rank-1 taps through conv_tile_kernel, 8 words per sample moved")
            finally:
                eng3.close()
        # the same call with every dense kernel through the 2-D stencil body (pb_set_dense_eval: PB_DENSE_STENCIL)
        eng.set_dense_eval("stencil")
        try:
            for _ in range(2):
                step()
            dt_st, _ = timed(args.steps, step)
        finally:
            eng.set_dense_eval("auto", capi.PB_DENSE_MIN_PHASES)
        ms_st = 1e3 * dt_st / args.steps
        side["end_to_end_dense_stencil_body"] = dict(ms_per_step=round(ms_st, 4), mp_per_s=round(B * H * W / 1e6 / (ms_st * 1e-3), 1))
        # the reference's own choice of method on a GPU (main.py:109-112): the zero boundary; and the edgetaper option
        # ... and the other optional stages on the headline image, one at a time (deblurring.py:80-88,172-208; prefiltering=True
        # is the bilateral filter in the reference, filters.py:107-148, the domain transform its commented-out alternative)
        for name, kw_x in (("end_to_end_method_direct", dict(kw, method="direct")), ("end_to_end_edgetaping", dict(kw, edgetaping=True)),
                           ("end_to_end_remove_halo", dict(kw, remove_halo=True)),
                           ("end_to_end_prefiltering_bilateral", dict(kw, prefiltering=True)),
                           ("end_to_end_prefiltering_domain_transform", dict(kw, prefiltering=True, prefilter="domain_transform"))):
            try:
                for _ in range(2):
                    polyblur_deblurring(x, **kw_x)
                dt_x, _ = timed(args.steps, lambda: polyblur_deblurring(x, **kw_x))
                ms_x = 1e3 * dt_x / args.steps
                side[name] = dict(ms_per_step=round(ms_x, 4), mp_per_s=round(B * H * W / 1e6 / (ms_x * 1e-3), 1))
            except Exception as e:                               # a labelled extra must not cost the run its line
                side[name] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))
        # the opt-in x-t separable APPROXIMATION of the oblique kernels (method='direct_separable', zero boundary):
        # its speed, and its distance to the exact zero-boundary result on this image
        sep_kw = dict(kw, method="direct_separable")
        for _ in range(2):
            o_sep = polyblur_deblurring(x, **sep_kw)
        dt_sep, o_sep = timed(args.steps, lambda: polyblur_deblurring(x, **sep_kw))
        ms_sep = 1e3 * dt_sep / args.steps
        dsep = (o_sep.float() - polyblur_deblurring(x, **dict(kw, method="direct")).float()).abs()
        side["end_to_end_direct_separable_approximation"] = dict(
            ms_per_step=round(ms_sep, 4), mp_per_s=round(B * H * W / 1e6 / (ms_sep * 1e-3), 1),
            max_abs_vs_exact_direct=float(dsep.max()), mean_abs_vs_exact_direct=float(dsep.mean()),
            note="approximate by design; not the headline")
        # the method's own use case, labelled: the same scene under a MILD blur (sigma 0.7 / rho 0.45 at 30 degrees) -- the
        # later iterations estimate kernels within a 4-sample halo, whose whole polynomial is one window pass
        try:
            from polyblur_amd.synthetic import synthetic_blurry_image
            xm = torch.from_numpy(np.stack([synthetic_blurry_image(3, H, W, DEFAULT_SEED + i, blur=(0.7, 0.45, 30.0))[0]
                                            for i in range(B)])).to(x.device).to(tdt).contiguous()
            for _ in range(2):
                _, minfos = polyblur_deblurring(xm, return_info=True, **kw)
            dt_m, _ = timed(args.steps, lambda: polyblur_deblurring(xm, **kw))
            ms_m = 1e3 * dt_m / args.steps
            side["end_to_end_mild_blur"] = dict(
                ms_per_step=round(ms_m, 4), mp_per_s=round(B * H * W / 1e6 / (ms_m * 1e-3), 1),
                estimated_blur=[dict(sigma=round(float(i["sigma"][0]), 3), rho=round(float(i["rho"][0]), 3),
                                     theta_deg=round(float(np.rad2deg(i["theta"][0])), 1)) for i in minfos],
                note="same generator and call as the headline, blur (0.7, 0.45, 30 deg) instead of the drawn one; not the headline")
            del xm
        except Exception as e:                                   # a labelled extra must not cost the run its line
            side["end_to_end_mild_blur"] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))
        # ---- the other BASELINE configs, driver-observed: per-GPU shares on this GPU, 1 warm-up + 3 timed steps each,
        # image 0 of the timed batch against the oracle (cfg5: no oracle image -- the 8K NumPy call takes minutes; its parity
        # at full size is tests/test_gpu_fullsize.py's) ---------------------------------------------------------------------
        if args.config == "cfg2" and (B, H, W) == (1, 2160, 3840):
            x1080 = None
            for name in ("cfg3", "cfg4", "cfg5"):
                c2 = CONFIGS[name]
                key = name if name == "cfg3" else name + "_share"
                try:
                    b2, h2, w2 = c2["batch"], c2["height"], c2["width"]
                    t_dt = torch.float32 if c2["dtype"] == "f32" else torch.float16
                    s2 = 4 if c2["dtype"] == "f32" else 2
                    if name == "cfg5":
                        xs_np = np.tile(x_np[:1], (1, 1, 2, 2))            # the headline's 4K image 2 x 2: an 8K image without 40 s of generator
                        src = "the headline's 4K image tiled 2 x 2"
                    else:
                        if x1080 is None:
                            x1080 = make_batch(4, h2, w2, DEFAULT_SEED + 77)[0]
                        xs_np = np.concatenate([x1080] * (b2 // 4))
                        src = "4 distinct synthetic 1080p images, repeated"
                    xs = torch.from_numpy(xs_np).to(dev).to(t_dt).contiguous()
                    del xs_np
                    kw2 = dict(KW, n_iter=c2["n_iter"], **c2["opts"])
                    o2, inf2 = polyblur_deblurring(xs, return_info=True, **kw2)            # the warm-up step; also the parity output
                    dt2, _ = timed(3, lambda: polyblur_deblurring(xs, **kw2))
                    ms2 = 1e3 * dt2 / 3
                    n2 = b2 * 3 * h2 * w2
                    words2 = survey_words(c2["opts"])
                    e2e2 = words2 * s2 * n2 * c2["n_iter"] / (ms2 * 1e-3) / 1e9
                    ent = dict(ms_per_step=round(ms2, 4), steps=3, warmup=1, mp_per_s=round(b2 * h2 * w2 / 1e6 / (ms2 * 1e-3), 1),
                               workload="batch=%d %dx%dx3 %s, n_iter=%d%s" % (b2, w2, h2, "fp32" if s2 == 4 else "fp16", c2["n_iter"],
                                                                              "".join(", %s=%s" % kv for kv in sorted(c2["opts"].items()))),
                               data=src, end_to_end=dict(achieved=round(e2e2, 1), unit="GB/s", frac=round(e2e2 / HBM_PEAK_GBS, 4),
                                                         words_per_sample_per_iteration=words2, basis="SURVEY 8d algorithmic bytes"))
                    if name != "cfg3":
                        ent["note"] = "the per-GPU share of the 8-GPU config, resident on one GPU"
                    try:
                        pit = forms_per_iteration(eng, b2, c2["n_iter"])
                        mp2, me2 = moved_bytes(pit, c2["opts"], s2, h2, w2)
                        ent["hbm_min"] = dict(bytes_per_step=int(me2), achieved=round(me2 / (ms2 * 1e-3) / 1e9, 1), unit="GB/s",
                                              frac=round(me2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              one_pass_images_per_iteration=[q["one_pass_images"] for q in pit],
                                              on_128x128_windows_per_iteration=[q["one_pass_on_128x128_windows"] for q in pit],
                                              note="bytes the forms that ran have to move end to end (options' own stages not counted)")
                    except Exception as e:                               # noqa: BLE001 -- a label
                        ent["hbm_min"] = dict(error="%s: %s" % (type(e).__name__, str(e)[:120]))
                    if name != "cfg5" and not args.no_parity:
                        from oracle import polyblur_ref as ref          # checker only
                        xin = xs[:1].float().cpu().numpy()
                        want2, winf2 = ref.polyblur_deblurring(xin, method="fft", return_info=True, **kw2)
                        err2 = float(np.abs(o2[:1].float().cpu().numpy() - want2).max())
                        th_a = [float(i["theta"][0]) for i in inf2]
                        th_b = [float(i["theta"][0]) for i in winf2]
                        tol2 = 2e-5 if s2 == 4 else 1e-3
                        ent["parity"] = dict(max_abs=err2, tolerance=tol2, theta_sequence_equal=th_a == th_b, image="image 0 of the batch vs the oracle")
                        if not (err2 <= tol2 and th_a == th_b):
                            side_failed.append(key)
                    elif name == "cfg5":
                        ent["parity"] = dict(skipped="no oracle image in the bench run (the 8K NumPy call takes minutes); tests/test_gpu_fullsize.py "
                                                     "compares this config with the oracle at full size", finite=bool(torch.isfinite(o2.float()).all()))
                    side[key] = ent
                    del xs, o2
                    torch.cuda.empty_cache()
                except Exception as e:                                   # noqa: BLE001 -- a labelled extra must not cost the run its line
                    side[key] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))
        # host buffers in and out (PCIe-inclusive; never the headline value)
        xn = x_np.astype(np.float32 if s == 4 else np.float16)
        polyblur_deblurring(torch.from_numpy(xn), **kw)
        t0 = time.perf_counter()
        polyblur_deblurring(torch.from_numpy(xn), **kw)
        ms_pcie = 1e3 * (time.perf_counter() - t0)
        side["end_to_end_host_buffers_pcie"] = dict(ms_per_step=round(ms_pcie, 3), mp_per_s=round(B * H * W / 1e6 / (ms_pcie * 1e-3), 1))

    # ---- CPU baseline + full-size parity: the oracle (port of the reference's CPU path) -----------------
    cpu, parity, failed = None, None, False
    want_parity = world == 1 and not args.no_parity and not from_root
    ref_res = None
    if world == 1 and not args.no_cpu_baseline and not from_root:
        cpu_samples, ref_res = cpu_baseline_samples(x_np if s == 4 else x_np.astype(np.float16).astype(np.float32),
                                                    dict(KW, n_iter=cfg["n_iter"]), want_parity and not cfg["opts"])
        head = [c for c in cpu_samples if c["threads"] == 1][-1]
        cpu = dict(value=head["mp_per_s"], unit="MP/s", cores=1, kind="port",
                   sample="one %s x3 fp32 image (the headline's), n_iter=%d, method='fft', NumPy oracle, single thread, %.1f s"
                          % (head["image"], cfg["n_iter"], head["seconds"]),
                   host_cores=os.cpu_count(), omp_num_threads=os.environ.get("OMP_NUM_THREADS"), samples=cpu_samples)
    if want_parity:
        from oracle import polyblur_ref as ref          # checker only
        xin = np.ascontiguousarray(x_np[:1]).astype(np.float32 if s == 4 else np.float16).astype(np.float32)
        if ref_res is None:
            ref_res = ref.polyblur_deblurring(xin, method="fft", return_info=True, **kw)
        want, winfos = ref_res
        got = out_info[:1].float().cpu().numpy()
        err = float(np.abs(got - want).max())
        th_hip = [float(i["theta"][0]) for i in infos]
        th_ref = [float(i["theta"][0]) for i in winfos]
        tol = 2e-5 if s == 4 else 1e-3
        parity = dict(max_abs=err, tolerance=tol, theta_sequence_equal=th_hip == th_ref, image="%dx%d, image 0 of the batch" % (W, H),
                      oracle="oracle/polyblur_ref.py (NumPy fp32, pinned to the reference by tests/golden)")
        failed = not (err <= tol and th_hip == th_ref)
    if side_failed:
        failed = True

    desc = "batch=%d %dx%dx3 %s per GPU, n_iter=%d, method=fft (circular), full 25-tap support" % (
        B, W, H, "fp32" if s == 4 else "fp16", cfg["n_iter"])
    if cfg["opts"]:
        desc += ", " + ", ".join("%s=%s" % kv for kv in sorted(cfg["opts"].items()))
    line = {
        "metric": "megapixels/sec (n_iter=3, alpha=6, beta=1)", "value": round(value, 1), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle_steps, "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_device": (dict(median=round(step_ms[len(step_ms) // 2], 4), min=round(step_ms[0], 4), max=round(step_ms[-1], 4),
                                    n=len(step_ms), note="one event between steps, device time; `value` is the contract's total / K")
                               if step_ms else None),
        "value_at_median_step": (round(mp_per_step / (step_ms[len(step_ms) // 2] * 1e-3), 1) if step_ms and world == 1 else None),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": desc, "name": args.config, "mode": args.mode, "images_per_gpu": B, "height": H, "width": W,
                   "parallelism": "images sharded, no data-path collective" if not from_root else
                                  "batch on rank 0: RCCL scatter + compute + gather per step"},
        "rccl_ranks": rccl_ranks, "roofline": roofline, "parity": parity, "cpu_baseline": cpu,
        "stages_ms_per_step": stages_ms, "estimated_blur": est, "context": side, "workspace_bytes": eng.workspace_bytes(),
    }
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("PARITY FAILURE: %r %r" % (parity, side_failed), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
