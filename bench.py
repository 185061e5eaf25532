#!/usr/bin/env python3
"""bench.py -- megapixels/s of the Polyblur hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full polyblur_deblurring() call (n_iter=3, alpha=6, beta=1, c=0.362, b=0.468,
q=0, method='fft', full 25-tap support) on one resident batch: BASELINE config 2, a single
4K (3840x2160x3) fp32 image per GPU.  With N > 1 every rank owns an independent image
(images shard with no data-path collective; "scaling": "weak"); the timed region is bracketed by
a barrier + device synchronise on both sides and the maximum over ranks is reported.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      -- the dominant kernel (the stencil pass) over the timed region: algorithmic
                   bytes per launch / average launch duration (hipEvents on the engine's stream)
  cpu_baseline  -- the NumPy oracle (a port of the reference's CPU path) timed on this box's host
                   cores on a bounded sample (rank 0, N == 1 only)
  stages_ms, inner_loop_rank1, adaptive -- context numbers (labelled), never the headline value.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
VALU_PEAK_TFLOPS = 157.3      # fp32 packed FMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the labelled side measurements, so that a profiler sees only the headline workload's kernels")
    ap.add_argument("--cpu-sample", default="2160x3840", help="HxW crop the CPU baseline is timed on")
    return ap.parse_args()


def make_batch(b, h, w, seed0):
    """Synthetic (b,3,h,w) batch.  Distinct images up to 4, then tiled: generation is
    CPU-side NumPy and would otherwise dominate start-up for the large batches."""
    from polyblur_amd.synthetic import synthetic_blurry_batch
    nd = min(b, 4)
    x, params = synthetic_blurry_batch(nd, 3, h, w, seed0=seed0)
    if b > nd:
        x = np.concatenate([x] * ((b + nd - 1) // nd))[:b]
    return x, params


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from polyblur_amd import polyblur_deblurring
    from polyblur_amd import _capi as capi
    from polyblur_amd.engine import get_engine
    from polyblur_amd.synthetic import DEFAULT_SEED

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B, H, W = args.batch, args.height, args.width
    tdt = torch.float32 if args.dtype == "f32" else torch.float16
    x_np, true_params = make_batch(B, H, W, DEFAULT_SEED + 1000 * rank)
    x = torch.from_numpy(x_np).to(dev).to(tdt).contiguous()
    eng = get_engine(local_rank)

    def step(support="full"):
        return polyblur_deblurring(x, support=support, **KW)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    sync_all()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # the slowest rank sets the time
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    mp_per_step = B * H * W * world / 1e6
    value = mp_per_step / (ms_per_step / 1e3)

    if rank != 0:
        if use_dist:
            dist.barrier()                       # leave together with rank 0 (it prints the line first)
            dist.destroy_process_group()
        return
    if world > 1:
        args.no_context = True                   # the labelled side measurements belong to the 1-GPU run

    s = 4 if args.dtype == "f32" else 2
    samples = B * 3 * H * W
    # ---- roofline of the dominant kernel over the timed region ---------------------------------
    conv_ms, conv_n = prof["conv"]
    # one polynomial application = 3 launches = (2s + 3s + 3s) bytes per sample
    alg_bytes_per_launch = 8.0 * s * samples / 3.0
    conv_avg_ms = conv_ms / max(conv_n, 1)
    achieved = alg_bytes_per_launch / (conv_avg_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of the same kernel from the rocprofv3 PMC passes of this same command
    # (tools/profile_bench.sh -> profiles/*_traffic.json; separate runs, FETCH_SIZE x2 on gfx950)
    traffic, traffic_src = None, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_traffic.json")))
        if cands and B == 1 and (H, W) == (2160, 3840) and s == 4:
            tj = json.load(open(cands[-1]))
            for k, v in tj.get("traffic", {}).items():
                if k.startswith("conv_tile_kernel<float, float, float>"):
                    traffic, traffic_src = v["hbm_bytes_per_launch"], os.path.basename(cands[-1])
    except Exception:
        pass
    roofline = dict(bound="hbm", kernel="conv_tile_kernel (stencil pass; taps as estimated, full 25x25 support)",
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    traffic=traffic, traffic_source=traffic_src, launches=conv_n, avg_launch_ms=round(conv_avg_ms, 5),
                    algorithmic_bytes_per_launch=int(alg_bytes_per_launch))
    # whole-step figure of SURVEY 8d: 9 words per sample per iteration (8 for the polynomial + 1 read for the estimate)
    e2e_gbs = 9.0 * s * samples * KW["n_iter"] * world / (ms_per_step * 1e-3) / 1e9
    roofline["end_to_end"] = dict(algorithmic_bytes_per_step=int(9.0 * s * samples * KW["n_iter"]), achieved=round(e2e_gbs, 1),
                                  unit="GB/s", frac=round(e2e_gbs / (HBM_PEAK_GBS * world), 4))
    stages_ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}

    # ---- context: what the estimator found, and the labelled side numbers ----------------------
    _, infos = polyblur_deblurring(x, return_info=True, **KW)
    est = [dict(theta_deg=round(float(np.rad2deg(i["theta"][0])), 1), sigma=round(float(i["sigma"][0]), 3),
                rho=round(float(i["rho"][0]), 3), separable=int(i["separable"][0]), radius=int(i["radius"][0]))
           for i in infos]
    # The synthetic blur is oblique (2 of the 30 candidate angles give a rank-1 kernel), so the estimated 25x25
    # kernels are dense and the stencil pass is fp32-VALU-bound, not HBM-bound: say how close to THAT ceiling it
    # runs (multiply-adds actually issued per launch / launch time; peak = 256 CU x 128 lanes x 2 x 2.4 GHz).
    macs = [sum((2 * int(r) + 1) * (2 if sp else (2 * int(r) + 1)) for r, sp in zip(i["radius"], i["separable"]))
            for i in infos]                                       # per sample position of the batch, per pass
    tflops = 2.0 * 3 * H * W * (sum(macs) / len(macs)) / (conv_avg_ms * 1e-3) / 1e12
    roofline["valu"] = dict(achieved=round(tflops, 1), peak=VALU_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(tflops / VALU_PEAK_TFLOPS, 4),
                            note="dense (non-rank-1) kernels estimated for this input: the pass is VALU-bound; "
                                 "context.inner_loop_rank1_* is the HBM-bound separable case")

    def inner_loop(theta_deg, sigma, rho, support, reps=20):
        buf = eng.make_kernels([sigma] * B, [rho] * B, [np.deg2rad(np.float32(theta_deg))] * B, support=support,
                               name="bench.info")
        o = torch.empty_like(x)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ms = eng.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32 if s == 4 else capi.PB_F16, x.shape, buf.ptr,
                                 KW["alpha"], KW["beta"], capi.PB_WRAP, reps)
        gbs = 8.0 * s * samples / (ms * 1e-3) / 1e9
        return dict(ms=round(ms, 4), achieved_GBps=round(gbs, 1), frac_of_8TBps=round(gbs / HBM_PEAK_GBS, 4),
                    mp_per_s=round(B * H * W / 1e6 / (ms * 1e-3), 1))

    side = {} if args.no_context else {
        # the north-star figure: n_iter's separable-conv inner loop, rank-1 kernels (theta forced to 0)
        "inner_loop_rank1_full_support": inner_loop(0.0, 2.0, 1.0, capi.PB_SUPPORT_FULL),
        "inner_loop_rank1_adaptive_sigma1": inner_loop(0.0, 1.0, 0.6, capi.PB_SUPPORT_ADAPTIVE),
        "inner_loop_general_full_support": inner_loop(30.0, 2.0, 1.0, capi.PB_SUPPORT_FULL),
        "inner_loop_general_adaptive_sigma1": inner_loop(30.0, 1.0, 0.6, capi.PB_SUPPORT_ADAPTIVE),
    }
    if not args.no_context:
        # adaptive-support end-to-end (same results to fp32 rounding), labelled
        for _ in range(2):
            step("adaptive")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step("adaptive")
        torch.cuda.synchronize(dev)
        ms_ad = 1e3 * (time.perf_counter() - t0) / args.steps
        side["end_to_end_adaptive_support"] = dict(ms_per_step=round(ms_ad, 4), mp_per_s=round(B * H * W / 1e6 / (ms_ad * 1e-3), 1))
        # host buffers in and out (PCIe-inclusive; never the headline value)
        xn = x_np.astype(np.float32 if s == 4 else np.float16)
        polyblur_deblurring(torch.from_numpy(xn), **KW)
        t0 = time.perf_counter()
        polyblur_deblurring(torch.from_numpy(xn), **KW)
        ms_pcie = 1e3 * (time.perf_counter() - t0)
        side["end_to_end_host_buffers_pcie"] = dict(ms_per_step=round(ms_pcie, 3), mp_per_s=round(B * H * W / 1e6 / (ms_pcie * 1e-3), 1))

    # ---- CPU baseline: the oracle (port of the reference's CPU path) on a bounded sample ---------
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import polyblur_ref as ref          # checker / baseline only
        ch, cw = (int(v) for v in args.cpu_sample.split("x"))
        ch, cw = min(ch, H), min(cw, W)
        crop = np.ascontiguousarray(x_np[:1, :, :ch, :cw]).astype(np.float32)
        t0 = time.perf_counter()
        ref.polyblur_deblurring(crop, method="fft", **KW)
        cdt = time.perf_counter() - t0
        cpu = dict(value=round(ch * cw / 1e6 / cdt, 4), unit="MP/s", cores=1, kind="port",
                   sample="one %dx%dx3 fp32 crop of the same synthetic image, n_iter=3, method='fft', NumPy oracle, "
                          "single thread (host has %d cores), %.1f s" % (cw, ch, os.cpu_count() or 0, cdt))

    line = {
        "metric": "megapixels/sec (n_iter=3, alpha=6, beta=1)", "value": round(value, 1), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "batch=%d %dx%dx3 %s per GPU, n_iter=3, method=fft (circular), full 25-tap support"
                               % (B, W, H, "fp32" if s == 4 else "fp16"),
                   "images_per_gpu": B, "height": H, "width": W, "parallelism": "images sharded, no collective"},
        "roofline": roofline, "cpu_baseline": cpu, "stages_ms_per_step": stages_ms, "estimated_blur": est,
        "context": side, "workspace_bytes": eng.workspace_bytes(),
    }
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
