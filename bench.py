#!/usr/bin/env python3
"""Headline benchmark: particles/s splatted to a 1024^2 float32 buffer (+ ms/frame).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of the hot path over the resident synthetic snapshot: splat kernels ->
(N > 1) RCCL sum-reduce of the image to rank 0 -> colormap kernel on rank 0; synchronous, inputs
already resident in HBM, upload/generation excluded.

Workload = BASELINE.json configs[3]: the 1e9-particle dm snapshot, density, 1024^2, camera A (identity
rotation, scale 200), reference TestDataLoader distribution and h-law, generated on device.  The N-series is STRONG
scaling on that ONE snapshot (--total-particles, default 1e9): rank g holds the index range [g N/G, (g+1) N/G)
(the arithmetic of the reference's split_buffers.py:26-38), so N = 1 renders the whole snapshot on one GPU (20 GB
resident) and N = 8 is the config verbatim: 8 shards of 1.25e8 + one RCCL image reduce.  The same h-law (h ~
N_total^-1/3) holds at every N, so value(N) / value(1) is a speed-up of the same frame and nothing else.  On N > 1
lines rank 0 also times the whole snapshot on its own GPU afterwards and reports `speedup_vs_1gpu_same_snapshot`.
`--particles-per-gpu X` selects the old weak series instead (n_total = X * N; its h-law changes with N).

Prints ONE JSON line on rank 0 with the contract's keys plus `roofline` (dominant kernel against HBM, timed with
hipEvents on the stream it runs on), `roofline_fragment` (the frame and every kernel against the f32 vector peak, priced by
the fragments they draw: the bound that actually decides this workload), `roofline_mega` (the matrix-core kernel: issued
and USEFUL flops) and `cpu_baseline` (the CPU oracle, kind "port", on a bounded uniform sample of the same snapshot on the
host cores; `pynbody.sph.image` beside it when pynbody can be imported).  At N = 1 the line also carries driver-timed
extras: BASELINE configs[1] (1e7 weighted), configs[2] (exactly 1e8), configs[4] (5e7 rgb, 2048^2), one 1.25e8-particle
shard of the snapshot (the headline of rounds 1-3), its h-capped bandwidth regime and the option integrated_px.

At N = 1 the process never imports torch (north_star: no PyTorch on this path); torch.distributed is used at
N > 1 only, as the launcher's rendezvous for the 128-byte RCCL id, the barrier and the max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBPS = 6290.0 # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 64 FLOP/clk/SIMD
VALU_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32 vector peak (256 CUs x 4 SIMDs x 64 lanes x 2 flop x 2.4 GHz)
# canonical arithmetic of one fragment: bilinear (P >= 64 px: kernels H / H2 / H3) acc += gy*top + fy*bot
# = 2 FMAs once the x-interpolated texel rows exist; nearest (kernels S / M) one multiply-add of the texel into the pixel
FMAS_PER_FRAGMENT = {"stream": 1, "mid": 1, "huge": 2, "mega": 2}
B_ALG = {"density": 20, "weighted": 24, "rgb": 28}     # algorithmic bytes/particle
KERNELS = ("stream", "mid", "huge", "mega")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-particles", type=float, default=1e9,
                    help="the snapshot every N renders (strong scaling): rank g holds the index range [g n/N, (g+1) n/N)")
    ap.add_argument("--particles-per-gpu", type=float, default=0.0,
                    help="weak series instead: n_total = this * N (0 = off, the default)")
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--scale", type=float, default=200.0)
    ap.add_argument("--h-cap-px", type=float, default=0.0,
                    help="cap smoothing lengths so footprints are <= this many pixels (bandwidth-bound variant)")
    ap.add_argument("--mode", choices=["density", "weighted", "rgb"], default="density")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--generic", action="store_true", help="use the generic (global-atomic) kernel")
    ap.add_argument("--no-reorder", action="store_true")
    ap.add_argument("--integrated-px", type=int, default=0,
                    help="option integrated_px of the library (kernel I for footprints at least this wide); 0 = the default path")
    ap.add_argument("--p-mega-px", type=int, default=0,
                    help="option p_mega_px / p_mega2_px of the library: footprints at least this wide go to the matrix-core kernel "
                         "H3 (0 = the default since the end of round 4: kernel H2 draws every footprint >= 64 px)")
    ap.add_argument("--shared-device-dry-run", action="store_true",
                    help="harness test on a 1-GPU box: every rank uses device 0 and the image reduce is skipped (RCCL refuses two "
                         "ranks on one device), so the launcher logic of an N > 1 run -- rendezvous, shards, barriers, the "
                         "max-over-ranks time, the JSON line -- can be exercised; the line is marked and its value means nothing")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline frames: no extra configurations, no CPU baseline (what the profiler runs, so "
                         "that every kernel instance in a profile belongs to the headline workload)")
    return ap.parse_args()


def camera(scale):
    """camera A (reference default view): identity rotation, zero offset"""
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] /= scale
    M[2, :] = [0.0, 0.0, 0.5 / scale, 0.5]
    return M, np.float32(1.0 / scale)


def num_strata(n):
    from topsy_amd.particle_buffers import ParticleBuffers
    return ParticleBuffers._num_strata(n)


def profile_entry(prof, kernel, mode, section="per_kernel"):
    """The PMC entry of `kernel` for the instantiation the headline frames run (MODE, channels = the first two
    template arguments): exactly one key may match, else None -- never 'the last one that contains the name'."""
    tmpl_mode = 2 if mode == "rgb" else 0
    first = {"density": "1", "weighted": "2", "rgb": None}[mode]
    if kernel == "splat_mega64_kernel":      # <MODE, waves per SIMD>: density only
        first = None
    hits = []
    for k, v in prof.get(section, {}).items():
        name, _, args = k.partition("<")
        if name.split("::")[-1] != kernel:
            continue
        a = [x.strip() for x in args.rstrip(">").split(",")]
        if not a or a[0] != str(tmpl_mode):
            continue
        if a[-1] == "true":          # the instantiation with fragment counting compiled in: the one counting frame, not the timed ones
            continue
        if first is not None and (len(a) < 2 or a[1] != first):
            continue
        hits.append((k, v))
    return hits[0] if len(hits) == 1 else (None, None)


def shard_range(n_total, world, rank):
    """index range of rank `rank` (the reference's SplitBuffers._calculate_splits arithmetic, split_buffers.py:26-38)"""
    first = (n_total * rank) // world
    return first, (n_total * (rank + 1)) // world - first


def make_context(_native, mips, R, channels, device, n_total, first, count, args, h_cap=0.0, mode_name="density"):
    ctx = _native.Context(R, channels, device_id=device)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n_total, first=first, count=count, seed=1337, h_cap=h_cap,
                           with_quantity=mode_name == "weighted", with_rgb=mode_name == "rgb")
    if not args.no_reorder:
        ctx.reorder_spatial(num_strata(count), 1337)       # load-time ordering, as the product path does
    if args.p_mega_px:
        ctx.set_option("p_mega_px", args.p_mega_px); ctx.set_option("p_mega2_px", args.p_mega_px)
    return ctx


def count_fragments(ctx, M, sf, mode, flags=0):
    """one extra frame with the fragment counters on: total and per kernel (S, M, H / H2, H3)"""
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, clear=True, mode=mode, flags=flags)
    st = ctx.stats()
    ctx.set_option("count_fragments", 0)
    return st["n_fragments"], {k: st["n_fragments_" + k] for k in KERNELS}


def fragment_roofline(frags_by_kernel, kernel_ms, frame_ms):
    """The frame and each kernel against the f32 vector peak, priced by the fragments they draw (SURVEY section 8d:
    t >= max(N B / BW, F c / Rate)): flop = fragments x canonical FMAs x 2."""
    per = {}
    flop_total = 0.0
    for k in KERNELS:
        flop = float(frags_by_kernel.get(k, 0)) * FMAS_PER_FRAGMENT[k] * 2.0
        flop_total += flop
        ms = kernel_ms.get(k, 0.0)
        tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        per[k] = {"fragments": int(frags_by_kernel.get(k, 0)), "fmas_per_fragment": FMAS_PER_FRAGMENT[k], "kernel_ms": ms,
                  "fragments_per_s": frags_by_kernel.get(k, 0) / (ms * 1e-3) if ms > 0 else 0.0,
                  "achieved": tf, "frac": tf / VALU_F32_PEAK_TFLOPS}
    tf = flop_total / (frame_ms * 1e-3) / 1e12 if frame_ms > 0 else 0.0
    return {"bound": "valu_f32", "achieved": tf, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_F32_PEAK_TFLOPS,
            "flop_per_frame": flop_total, "per_kernel": per,
            "note": "useful flop = fragments x canonical FMAs (1 nearest, 2 bilinear) x 2; set-up, addressing and atomics are overhead"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        args.gpus = world

    dist = torch = None
    if world > 1:
        # torch.distributed is the launcher's rendezvous only (the 128-byte RCCL id, barriers, the max-over-ranks time):
        # a gloo group of host tensors.  The data path's one collective -- the image reduce -- is this library's own RCCL
        # communicator (tsp_comm_*), so torch never touches the GPU here and there is a single RCCL communicator per rank.
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")

    from topsy_amd import _native, kernel_lut

    R = args.resolution
    weak = args.particles_per_gpu > 0
    if weak:
        n_per = int(args.particles_per_gpu)
        n_total = n_per * world
        first = rank * n_per
    else:
        n_total = int(args.total_particles)
        first, n_per = shard_range(n_total, world, rank)
    mode = {"density": _native.MODE_WEIGHTED, "weighted": _native.MODE_WEIGHTED, "rgb": _native.MODE_RGB}[args.mode]
    channels = 4 if args.mode == "rgb" else 2
    mips = kernel_lut.kernel_mips()
    h_cap = args.h_cap_px * args.scale / (2.0 * R) if args.h_cap_px > 0 else 0.0
    t_setup = time.time()
    if args.shared_device_dry_run:
        local_rank = 0
    ctx = make_context(_native, mips, R, channels, local_rank, n_total, first, n_per, args, h_cap, args.mode)
    t_setup = time.time() - t_setup
    if args.integrated_px:
        ctx.set_option("integrated_px", args.integrated_px)

    use_comm = world > 1 and not args.shared_device_dry_run
    if use_comm:
        ids = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(world, rank, ids[0])

    M, sf = camera(args.scale)
    import matplotlib
    lut = matplotlib.colormaps["twilight_shifted"](np.linspace(0.001, 0.999, 1000)).astype(np.float32)
    flags = _native.PIPE_GENERIC if args.generic else _native.PIPE_DEFAULT

    kernel_ms = {k: [] for k in KERNELS + ("total", "reduce")}
    vmin, vmax = -12.0, -4.0

    def frame(record):
        ctx.render(M, sf, clear=True, mode=mode, flags=flags)
        if record:
            st = ctx.stats()
            for k in KERNELS + ("total",):
                kernel_ms[k].append(st["ms_" + k])
        if use_comm:
            ms = ctx.comm_reduce_image(root=0)
            if record:
                kernel_ms["reduce"].append(ms)
        if rank == 0:
            if args.mode == "rgb":
                return ctx.colormap_rgb(vmin, vmax, 1.0)
            return ctx.colormap_scalar(lut, vmin, vmax, True, args.mode == "weighted")
        return None

    def barrier():
        # every tsp_* call is synchronous (it returns after its GPU work has completed: hipStreamSynchronize on the stream
        # the kernels and the RCCL reduce run on), so nothing is in flight here; at N > 1 the ranks meet
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        frame(False)
    barrier()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        frame(True)
        step_s.append(time.perf_counter() - ts)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frags, frags_by_kernel = count_fragments(ctx, M, sf, mode, flags)
    # the slowest shard decides the frame: per-kernel times of every rank (strong scaling: how even the shards are)
    means = {k: float(np.mean(v)) if v else 0.0 for k, v in kernel_ms.items()}
    per_rank_ms = None
    if dist is not None:
        f = torch.tensor([float(frags)] + [float(frags_by_kernel[k]) for k in KERNELS], dtype=torch.float64)
        dist.all_reduce(f)
        frags = f[0].item()
        frags_by_kernel = {k: f[1 + i].item() for i, k in enumerate(KERNELS)}
        mine = torch.tensor([means[k] for k in KERNELS + ("total", "reduce")], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [{k: float(e[i]) for i, k in enumerate(KERNELS + ("total", "reduce"))} for e in every]

    if rank != 0:
        if dist is not None:
            dist.barrier()           # rank 0 times the whole snapshot on one GPU meanwhile
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total / (elapsed / args.steps)
    ms_median = float(np.median(step_s)) * 1e3
    # dominant kernel of the frame and its roofline (HBM: B_alg bytes/particle streamed once)
    parts = {k: means[k] for k in KERNELS}
    if sum(parts.values()) <= 0.0:
        dom, dom_ms = "splat_generic_kernel", means["total"]
    else:
        dom = max(parts, key=parts.get)
        dom_ms = parts[dom]
        dom = {"stream": "splat_stream_kernel", "mid": "splat_mid_kernel",
               "mega": "splat_mega64_kernel" if args.mode == "density" else "splat_mega_kernel",
               "huge": "splat_huge2_kernel"}[dom]
    bytes_per_launch = B_ALG[args.mode] * n_per
    achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    stream_ms = means["stream"] if means["stream"] > 0 else means["total"]
    config_tag = " = BASELINE.json configs[3]" if (n_total == 10**9 and args.mode == "density" and R == 1024 and not weak
                                                    and args.h_cap_px <= 0) else ""
    workload_name = (f"{n_total:.4g} dm particles{config_tag}, {args.mode}, {R}^2 buffer, camera A "
                     f"(scale {args.scale:g}), reference TestDataLoader h-law, "
                     + (f"index-range sharded x{world} ({n_per:.4g}/GPU)" if world > 1 else "whole snapshot resident on one GPU")
                     + (f", h capped at {args.h_cap_px:g} px" if args.h_cap_px > 0 else "")
                     + (f", option integrated_px = {args.integrated_px}" if args.integrated_px else "")
                     + (f", option p_mega_px = {args.p_mega_px}" if args.p_mega_px else "")
                     + ", splat + " + ("RCCL image reduce + " if world > 1 else "") + "colormap")
    measured_peak = ctx.measure_read_bandwidth(4 << 30, 5)
    # HBM bytes of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); null when no profile matches this workload
    traffic = traffic_kernel = None
    mfma_per_launch = None      # v_mfma_f32_* instructions of kernel H3 per launch (PMC SQ_INSTS_MFMA)
    mega_kernel = "splat_mega_kernel"
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "latest_bench_counters.json")))
        option = f", option p_mega_px = {args.p_mega_px}" if args.p_mega_px else ""
        section = "mfma_option_per_kernel" if args.p_mega_px == 768 else "per_kernel"      # (the option's own PMC pass: tools/profile_bench.sh)
        if prof.get("bench_line", {}).get("config", {}).get("workload") == workload_name.replace(option, "") and (section != "per_kernel" or not option):
            traffic_kernel, v = profile_entry(prof, dom, args.mode, section)
            if v is not None and "hbm_read_bytes_corrected" in v:
                traffic = (v.get("hbm_read_bytes_corrected", 0.0) + v.get("hbm_write_bytes", 0.0)) / 1e9
            for mega_name in ("splat_mega64_kernel", "splat_mega_kernel"):      # 64 x 64 or 64 x 32 strips, whichever ran
                _, v = profile_entry(prof, mega_name, args.mode, section)
                if v is not None and v.get("SQ_INSTS_MFMA"):
                    mfma_per_launch, mega_kernel = v["SQ_INSTS_MFMA"], mega_name
                    break
    except Exception:
        pass
    result = {
        "metric": "particles/sec splatted to 1024^2 buffer",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_name, "total_particles": n_total,
                   "particles_per_gpu": n_per, "resolution": R, "sharding": f"index-range x{world}",
                   "pipeline": "generic" if args.generic else ("four-class (stream / mid scatter / row-uniform gather / MFMA)" if args.p_mega_px
                                                               else "three-class (stream / mid scatter / row-uniform gather)"),
                   "fragments_per_particle": frags / n_total, "frames_per_s": 1e3 / ms_per_step},
        "ms_per_step_median": ms_median, "value_at_median": n_total / (ms_median * 1e-3),
        "fragments_per_s": frags / (ms_per_step * 1e-3),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_unit": "GB per launch (PMC)",
                     "traffic_kernel": traffic_kernel, "kernel_ms": dom_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "measured_read_peak_GBps": measured_peak, "guide_measured_copy_GBps": HBM_MEASURED_COPY_GBPS,
                     "stream_kernel_ms": stream_ms,
                     "stream_kernel_GBps": bytes_per_launch / (stream_ms * 1e-3) / 1e9 if stream_ms > 0 else 0.0,
                     "stream_kernel_frac": bytes_per_launch / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if stream_ms > 0 else 0.0,
                     "frame_GBps": B_ALG[args.mode] * n_total / (ms_per_step * 1e-3) / 1e9,
                     "frame_frac": B_ALG[args.mode] * n_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS / world,
                     "note": "the dominant kernel is priced against HBM because the contract asks for it; it is bound by the "
                             "fragments it draws -- see roofline_fragment"},
        # per GPU: a rank draws 1/N of the frame's fragments (rank 0's kernel times beside the mean share)
        "roofline_fragment": fragment_roofline({k: v / world for k, v in frags_by_kernel.items()}, means, ms_per_step),
        "kernel_ms": means,
        "setup_s": t_setup,
    }
    if args.shared_device_dry_run:
        result["dry_run"] = "ranks shared device 0 and no image reduce ran: harness test only, `value` is not a measurement"
    if world > 1:
        result["roofline_fragment"]["note_n_gpus"] = (f"per GPU: 1/{world} of the frame's fragments against rank 0's kernel times "
                                                      "and the step time")
        result["per_rank_kernel_ms"] = per_rank_ms
        tot = [r["total"] for r in per_rank_ms]
        result["shard_balance_max_over_mean"] = max(tot) / (sum(tot) / len(tot)) if sum(tot) > 0 else None
    if mfma_per_launch and means["mega"] > 0 and not args.integrated_px:
        # the matrix-core kernel against ITS roofline: instruction count from the committed PMC pass, duration live.
        # Issued flop count every K slot of every MFMA; useful flop are the 2 FMAs per fragment the kernel exists for.
        issued = mfma_per_launch * 2 * 32 * 32 * 2
        useful = float(frags_by_kernel["mega"]) * FMAS_PER_FRAGMENT["mega"] * 2.0 / world
        tflops = issued / (means["mega"] * 1e-3) / 1e12
        useful_tflops = useful / (means["mega"] * 1e-3) / 1e12
        result["roofline_mega"] = {"bound": "mfma", "kernel": mega_kernel, "achieved": useful_tflops, "peak": MFMA_F32_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": useful_tflops / MFMA_F32_PEAK_TFLOPS, "kernel_ms": means["mega"],
                                   "issued_TFLOPs": tflops, "issued_frac": tflops / MFMA_F32_PEAK_TFLOPS,
                                   "useful_frac": useful / issued if issued > 0 else None,
                                   "mfma_instructions_per_launch": mfma_per_launch,
                                   "note": "achieved / frac count USEFUL flop (fragments x 2 FMAs x 2); issued_* count every K slot "
                                           "of every v_mfma_f32_32x32x2_f32 (2 x 32 x 32 x 2 flop each)"}
    if world > 1 and not weak and not args.headline_only:
        # the same snapshot whole on ONE GPU (rank 0's, the others wait at the barrier): what N = 1 prints as `value`
        try:       # (a second context next to this rank's shard: the communicator stays up until every rank is done)
            one = whole_snapshot_line(_native, mips, R, channels, local_rank, n_total, args, mode, lut, vmin, vmax)
            result["one_gpu_same_snapshot"] = one
            result["speedup_vs_1gpu_same_snapshot"] = one["ms_per_step"] / ms_per_step
        except _native.BackendError as e:
            result["one_gpu_same_snapshot"] = {"error": str(e)[:200]}
            result["speedup_vs_1gpu_same_snapshot"] = None
    extras = (world == 1 and not args.generic and not args.headline_only and args.h_cap_px <= 0 and not args.integrated_px
              and not args.p_mega_px)
    if extras and args.mode == "density":
        if ctx is not None:
            ctx.close()
        # one 1.25e8-particle shard of the 1e9 snapshot = the headline of rounds 1-3 (what each of 8 GPUs renders), and on it
        # the opt-in kernel I and the bandwidth regime (the same positions with footprints capped at
        # 8 px isolate the streaming kernels); reported beside the headline, never as `value`
        n_sh = 125_000_000
        ctx = make_context(_native, mips, R, 2, local_rank, n_sh, 0, n_sh, args)
        result["shard_1p25e8"] = config_line(ctx, n_sh, "density", R, args, "a 1.25e8-particle snapshot (the headline of "
                                             "rounds 1-3; h-law of N = 1.25e8)", regenerate=False)
        result["matrix_core_option"] = matrix_core_line(ctx, M, sf, mode, n_sh)
        result["integrated_option"] = integrated_line(ctx, M, sf, mode, n_sh)
        result["bandwidth_regime"] = hcapped_line(args, ctx, n_sh, n_sh, 0, M, sf, mode, measured_peak)
        # the other single-GPU configurations of BASELINE.json, driver-timed beside the headline (never `value`)
        result["baseline_config_1"] = config_line(ctx, 10_000_000, "weighted", R, args,
                                                  "BASELINE.json configs[1]: 1e7 particles, density-weighted quantity")
        result["baseline_config_2"] = config_line(ctx, 100_000_000, "density", R, args,
                                                  "BASELINE.json configs[2]: exactly 1e8 dm particles, density")
        ctx.close()
        ctx = None
        c5 = _native.Context(2048, 4, device_id=local_rank)
        c5.set_kernel_mips(mips)
        result["baseline_config_4"] = config_line(c5, 50_000_000, "rgb", 2048, args,
                                                  "BASELINE.json configs[4]: 5e7 star particles, rgb, 2048^2")
        c5.close()
    if not args.no_cpu_baseline and not args.headline_only and world == 1:      # reported baseline: rank 0 at N = 1 only
        result["cpu_baseline"] = cpu_baseline(args, n_total, M, sf, R)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def whole_snapshot_line(_native, mips, R, channels, device, n_total, args, mode, lut, vmin, vmax, frames=5):
    """The whole snapshot on one GPU, timed like the headline (wall clock around render + colormap)."""
    c = make_context(_native, mips, R, channels, device, n_total, 0, n_total, args, 0.0, args.mode)
    M_, sf_ = camera(args.scale)

    def one():
        c.render(M_, sf_, clear=True, mode=mode)
        if args.mode == "rgb":
            c.colormap_rgb(vmin, vmax, 1.0)
        else:
            c.colormap_scalar(lut, vmin, vmax, True, args.mode == "weighted")

    one()
    t = time.perf_counter()
    for _ in range(frames):
        one()
    ms = (time.perf_counter() - t) / frames * 1e3
    st = c.stats()
    c.close()
    return {"workload": f"the same {n_total:.4g}-particle snapshot whole on one GPU", "ms_per_step": ms,
            "value": n_total / (ms * 1e-3), "unit": "particles/s", "frames": frames,
            "kernel_ms": {k: st["ms_" + k] for k in KERNELS}}


def matrix_core_line(ctx, M, sf, mode, n_per, px=768, frames=10):
    """The frame with kernel H3 (v_mfma_f32_32x32x2_f32) drawing the footprints >= px -- the default of rounds 2-4 -- beside the
    default path, whose kernel H2 draws them: why the matrix cores are an option now."""
    def run():
        ms, h2, h3 = [], [], []
        for i in range(frames + 1):
            t = ctx.render(M, sf, clear=True, mode=mode)
            if i:
                st = ctx.stats(); ms.append(t); h2.append(st["ms_huge"]); h3.append(st["ms_mega"])
        return float(np.median(ms)), float(np.median(h2)), float(np.median(h3))
    ms0, h20, _ = run()
    _, by0 = count_fragments(ctx, M, sf, mode)
    ctx.set_option("p_mega_px", px)
    ms1, h21, h31 = run()
    _, by1 = count_fragments(ctx, M, sf, mode)
    n_mega = int(ctx.stats()["n_mega"])
    ctx.set_option("p_mega_px", 0)
    useful = float(by1["mega"]) * FMAS_PER_FRAGMENT["mega"] * 2.0
    return {"workload": f"the 1.25e8-particle snapshot with the option p_mega_px = {px} (footprints >= {px} px through kernel H3)",
            "ms_per_step": ms1, "ms_per_step_default_path": ms0, "value": n_per / (ms1 * 1e-3), "unit": "particles/s",
            "kernel_H2_ms": h21, "kernel_H3_ms": h31, "kernel_H2_ms_default_path": h20, "records_through_kernel_H3": n_mega,
            "fragments_through_kernel_H3": int(by1["mega"]),
            "kernel_H2_ms_for_the_same_fragments": h20 - h21,
            "roofline_mega": {"bound": "mfma", "achieved": useful / (h31 * 1e-3) / 1e12 if h31 > 0 else 0.0, "peak": MFMA_F32_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": useful / (h31 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if h31 > 0 else 0.0,
                              "note": "useful flop (fragments x 2 FMAs x 2); f32 MFMA and packed-f32 VALU share one peak on gfx950, and "
                                      "the GEMM form issues 1.9x the useful flop (profiles/round4d: useful_frac 0.52, issued_frac 0.50)"}}


def integrated_line(ctx, M, sf, mode, n_per, px=256, frames=10):
    ctx.render(M, sf, clear=True, mode=mode)
    exact = ctx.read_image()[..., 0].astype(np.float64)
    ctx.set_option("integrated_px", px)
    ms, mega = [], []
    for i in range(frames + 1):
        t = ctx.render(M, sf, clear=True, mode=mode)
        if i:
            ms.append(t); mega.append(ctx.stats()["ms_mega"])
    st = ctx.stats()
    fast = ctx.read_image()[..., 0].astype(np.float64)
    ctx.set_option("integrated_px", 0)
    lit = exact > 0
    rel = np.abs(fast - exact)[lit] / exact[lit]
    return {"workload": f"the headline snapshot with the option integrated_px = {px} (footprints >= {px} px through kernel I)",
            "ms_per_step": float(np.median(ms)), "value": n_per / (float(np.median(ms)) * 1e-3), "unit": "particles/s",
            "kernel_I_ms": float(np.median(mega)), "records_through_kernel_I": int(st["n_mega"]),
            "max_relative_difference_per_pixel_from_the_exact_kernels": float(rel.max()) if rel.size else 0.0,
            "pixels_compared": int(lit.sum())}


def hcapped_line(args, ctx, n_total, n_per, rank, M, sf, mode, measured_peak, cap_px=8.0, frames=10):
    R = args.resolution
    ctx.generate_synthetic(n_total, first=rank * n_per, count=n_per, seed=1337, h_cap=cap_px * args.scale / (2.0 * R),
                           with_quantity=args.mode == "weighted", with_rgb=args.mode == "rgb")
    if not args.no_reorder:
        ctx.reorder_spatial(num_strata(n_per), 1337)
    ms, st, mid = [], [], []
    for i in range(frames + 1):
        t = ctx.render(M, sf, clear=True, mode=mode)
        if i:
            s = ctx.stats()
            ms.append(t); st.append(s["ms_stream"]); mid.append(s["ms_mid"])
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, clear=True, mode=mode)
    frags = ctx.stats()["n_fragments"]
    ctx.set_option("count_fragments", 0)
    b = B_ALG[args.mode] * n_per
    gbps = b / (float(np.median(st)) * 1e-3) / 1e9
    frame_gbps = b / (float(np.median(ms)) * 1e-3) / 1e9
    return {"workload": f"same snapshot, h capped so footprints <= {cap_px:g} px", "ms_per_step": float(np.median(ms)),
            "value": n_per / (float(np.median(ms)) * 1e-3), "unit": "particles/s", "fragments_per_particle": frags / n_per,
            "stream_kernel_ms": float(np.median(st)), "mid_kernel_ms": float(np.median(mid)),
            "stream_kernel_GBps": gbps, "frame_GBps": frame_gbps,
            "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS, "frame_frac_of_hbm_peak": frame_gbps / HBM_PEAK_GBPS,
            "frac_of_guide_measured_copy": gbps / HBM_MEASURED_COPY_GBPS, "frame_frac_of_guide_measured_copy": frame_gbps / HBM_MEASURED_COPY_GBPS,
            "frac_of_measured_read_peak": gbps / measured_peak}


def config_line(ctx, n, mode_name, R, args, label, frames=10, regenerate=True):
    """One of BASELINE.json's other configurations on this GPU: camera A, reference h-law, generated on device, load-time
    ordering; median of `frames` frames (splat only: tsp_render's own hipEvent time) after one warm-up."""
    from topsy_amd import _native
    mode = _native.MODE_RGB if mode_name == "rgb" else _native.MODE_WEIGHTED
    if regenerate:
        ctx.generate_synthetic(n, first=0, count=n, seed=1337, h_cap=0.0, with_quantity=mode_name == "weighted", with_rgb=mode_name == "rgb")
        if not args.no_reorder:
            ctx.reorder_spatial(num_strata(n), 1337)
    M, sf = camera(args.scale)
    ms, per = [], {k: [] for k in KERNELS}
    for i in range(frames + 1):
        t = ctx.render(M, sf, clear=True, mode=mode)
        if i:
            ms.append(t)
            st = ctx.stats()
            for k in KERNELS:
                per[k].append(st["ms_" + k])
    frags, by_kernel = count_fragments(ctx, M, sf, mode)
    med = float(np.median(ms))
    kms = {k: float(np.median(v)) for k, v in per.items()}
    rf = fragment_roofline(by_kernel, kms, med)
    return {"workload": f"{label}, {R}^2 buffer, camera A, reference h-law", "particles": n, "ms_per_step": med,
            "value": n / (med * 1e-3), "unit": "particles/s", "frames_per_s": 1e3 / med, "fragments_per_particle": frags / n,
            "fragments_per_s": frags / (med * 1e-3), "frac_of_f32_peak": rf["frac"],
            "kernel_frac_of_f32_peak": {k: v["frac"] for k, v in rf["per_kernel"].items()},
            "frame_GBps": B_ALG[mode_name] * n / (med * 1e-3) / 1e9, "kernel_ms": kms}


def cpu_baseline(args, n_total, M, sf, R):
    """The CPU oracle (C restatement, OpenMP over all host cores) on a bounded uniform sample of the
    same snapshot: rows [0, n_s) of the generator's index bijection are a uniform sample."""
    from topsy_amd import _native, kernel_lut
    from oracle import oracle_c
    mips = kernel_lut.kernel_mips()
    cores = oracle_c.max_threads()
    h_cap = args.h_cap_px * args.scale / (2.0 * R) if args.h_cap_px > 0 else 0.0

    def sample(n_s):
        c = _native.Context(64, 2, device_id=int(os.environ.get("LOCAL_RANK", "0")))
        c.generate_synthetic(n_total, first=0, count=n_s, seed=1337, h_cap=h_cap)
        d = c.download_particles(("x", "y", "z", "h", "mass"))
        c.close()
        return d

    def run(d):
        t = time.perf_counter()
        oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], None, None, mode=0, M=M, sf=float(sf), R=R, mips=mips)
        return time.perf_counter() - t

    # grow the sample until the oracle needs >= ~cpu_seconds/2 of wall time (bounded: <= 1e8 particles)
    n_s = 200000
    d = sample(n_s)
    run(d)                                   # warm-up
    secs = run(d)
    while secs < 0.5 * args.cpu_seconds and n_s < min(1e8, n_total):
        n_s = int(min(n_s * max(2.0, 0.8 * args.cpu_seconds / max(secs, 1e-3)), 1e8, n_total))
        d = sample(n_s)
        secs = run(d)
    out = {"value": n_s / secs, "unit": "particles/s", "cores": cores, "kind": "port",
           "sample": f"{n_s} particles (uniform sample of the {n_total:.4g}-particle snapshot, same camera, {R}^2, "
                     f"density), oracle/oracle.c OpenMP over image tiles, {secs:.1f} s"}
    out["pynbody_sph_image"] = pynbody_baseline(d, args, R)
    return out


def pynbody_baseline(d, args, R, max_particles=2_000_000):
    """pynbody's own CPU renderer on the same sample when pynbody can be imported (it is not installed
    in this image and there is no network: then the leg says so).  Labelled separately: its kernel normalisation and
    sub-pixel handling differ from topsy's GPU path, parity against it is unpinned."""
    try:
        import pynbody
    except Exception as e:
        return {"available": False, "reason": f"import pynbody: {type(e).__name__}: {e}"[:160]}
    try:
        n = min(len(d["x"]), max_particles)
        snap = pynbody.new(dm=n)
        snap["pos"] = np.stack([d["x"][:n], d["y"][:n], d["z"][:n]], axis=1).astype(np.float64)
        snap["mass"] = d["mass"][:n].astype(np.float64)
        snap["smooth"] = d["h"][:n].astype(np.float64)
        snap["rho"] = np.ones(n)
        snap["pos"].units, snap["mass"].units, snap["smooth"].units, snap["rho"].units = "kpc", "Msol", "kpc", "Msol kpc^-3"
        t = time.perf_counter()
        pynbody.plot.sph.image(snap, qty="rho", width=2.0 * args.scale, resolution=R, units="Msol kpc^-2", noplot=True, threaded=True)
        secs = time.perf_counter() - t
        return {"available": True, "value": n / secs, "unit": "particles/s", "cores": os.cpu_count(),
                "sample": f"{n} particles, pynbody {pynbody.__version__} plot.sph.image, {secs:.1f} s"}
    except Exception as e:
        return {"available": True, "error": f"{type(e).__name__}: {e}"[:200]}


if __name__ == "__main__":
    main()
