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

Prints ONE JSON line on rank 0 with the contract's keys plus `roofline` (HBM: the frame and kernel S -- the one kernel that
streams the particles -- first, then every kernel with ITS OWN algorithmic bytes, PMC traffic and traffic / algorithmic;
durations from hipEvents on the stream the kernels run on), `roofline_fragment` (the frame and every kernel against the f32
vector peak, priced by the fragments they draw: the bound that decides kernels G and H2) and `cpu_baseline` (the CPU oracle,
kind "port", on a bounded uniform sample of the same snapshot on the host cores; `pynbody.sph.image` beside it when pynbody
can be imported).  At N = 1 the line also carries driver-timed extras: `shards_of_1e9_x8` (the 8 REAL index-range shards of
the snapshot, one after another on this GPU: per-shard frame, max / mean, and the projected 1 -> 8 speed-up), the product
path through the Visualizer (`visualizer_export_frame`, `interactive_frame`), BASELINE configs[1] (1e7 weighted), configs[2]
(exactly 1e8), configs[4] (5e7 rgb, 2048^2), a stand-alone 1.25e8-particle snapshot (the headline of rounds 1-3), its
h-capped bandwidth regime.  At N > 1 the line is self-validating: before timing a
rank-dependent constant image is reduced and checked (`reduce_selftest`), after timing the reduced frame is compared with
the same snapshot rendered whole on rank 0's GPU (`reduce_check`); a failure exits non-zero.

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
VALU_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32 vector peak (256 CUs x 4 SIMDs x 64 lanes x 2 flop x 2.4 GHz)
# canonical arithmetic of one fragment: bilinear (P >= 64 px: kernel H2) acc += gy*top + fy*bot
# = 2 FMAs once the x-interpolated texel rows exist; nearest (kernels S / M) one multiply-add of the texel into the pixel
FMAS_PER_FRAGMENT = {"stream": 1, "mid": 1, "huge": 2}
B_ALG = {"density": 20, "weighted": 24, "rgb": 28}     # algorithmic bytes/particle
KERNELS = ("stream", "mid", "huge")        # tsp_stats names of kernels S, N (the mid footprints; kernel G when option mid_narrow_px_milli = 0), H2
KERNEL_SYMBOL = {"stream": "splat_stream_kernel", "mid": "splat_narrow_gather_kernel", "huge": "splat_huge2_kernel"}
# one ncclReduce of the R^2 x C float32 image onto the root over xGMI (ring: 7 steps of 1/8 of the image per link, ~153 GB/s per
# link and ~20 us per step): an ESTIMATE -- no multi-GPU box was available to this build -- used only by `projected_speedup_1to8`
REDUCE_ESTIMATE_MS = 0.3


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
    ap.add_argument("--shared-device-dry-run", action="store_true",
                    help="harness test on a 1-GPU box: every rank uses device 0 and the image reduce is skipped (RCCL refuses two "
                         "ranks on one device) -- the shard images are summed on the host instead, through the same self-test and "
                         "reduce_check -- so the launcher logic of an N > 1 run (rendezvous, shards, barriers, the max-over-ranks "
                         "time, the checks, the JSON line) can be exercised; the line is marked and its value means nothing")
    ap.add_argument("--as-shard", default="",
                    help="G:g -- profiling aid at N = 1: this GPU holds and renders only the index-range shard g of G of the snapshot "
                         "(what rank g of a G-GPU run does, without the reduce); the line is marked and is not the headline")
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


def camera_b(scale, x_angle=0.0, y_angle=0.4):
    """camera B (reference tests/test_render_output.py:161-198: `vis.scale = 20; vis.rotate(0, 0.4)`): the
    Visualizer's own rotation (visualizer.py:347-357) and transform (sph.py:268-289) at the given scale"""
    cx, sx, cy, sy = np.cos(x_angle), np.sin(x_angle), np.cos(y_angle), np.sin(y_angle)
    rot = np.array([[cx, 0, sx], [0, 1, 0], [-sx, 0, cx]]) @ np.array([[1, 0, 0], [0, cy, -sy], [0, sy, cy]])
    to_clip = np.diag([1.0, 1.0, 0.5, 1.0]); to_clip[2, 3] = 0.5
    rs = np.zeros((4, 4)); rs[:3, :3] = rot / scale; rs[3, 3] = 1.0
    return (to_clip @ rs).astype(np.float32), np.float32(1.0 / scale)


def num_strata(n):
    from topsy_amd.particle_buffers import ParticleBuffers
    return ParticleBuffers._num_strata(n)


def profile_entry(prof, kernel, mode, section="per_kernel"):
    """The PMC entry of `kernel` for the instantiation the headline frames run (MODE, channels = the first two
    template arguments): exactly one key may match, else None -- never 'the last one that contains the name'."""
    tmpl_mode = 2 if mode == "rgb" else 0
    first = {"density": "1", "weighted": "2", "rgb": None}[mode]
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
    return ctx


def count_fragments(ctx, M, sf, mode, flags=0):
    """one extra frame with the fragment counters on: total and per kernel (S, G, H2)"""
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
        if args.as_shard and world == 1:
            G, g = (int(v) for v in args.as_shard.split(":"))
            first, n_per = shard_range(n_total, G, g)
    mode = {"density": _native.MODE_WEIGHTED, "weighted": _native.MODE_WEIGHTED, "rgb": _native.MODE_RGB}[args.mode]
    channels = 4 if args.mode == "rgb" else 2
    mips = kernel_lut.kernel_mips()
    h_cap = args.h_cap_px * args.scale / (2.0 * R) if args.h_cap_px > 0 else 0.0
    t_setup = time.time()
    if args.shared_device_dry_run:
        local_rank = 0
    ctx = make_context(_native, mips, R, channels, local_rank, n_total, first, n_per, args, h_cap, args.mode)
    t_setup = time.time() - t_setup

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

    rgba = np.empty((R, R, 4), dtype=np.uint8)      # the presentation buffer of the loop (reused, as a display loop does)

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
            return ctx.colormap_scalar(lut, vmin, vmax, True, args.mode == "weighted", out=rgba)
        return None

    def barrier():
        # every tsp_* call is synchronous (it returns after its GPU work has completed: hipStreamSynchronize on the stream
        # the kernels and the RCCL reduce run on), so nothing is in flight here; at N > 1 the ranks meet
        if dist is not None:
            dist.barrier()

    selftest = None
    if world > 1:
        selftest = reduce_selftest(ctx, dist, torch, rank, world, use_comm, R, channels)

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

    # ---- N > 1: the reduced frame, for the check against the whole snapshot below (one more frame, untimed) ----------
    reduced_image = None
    if world > 1:
        frame(False)
        reduced_image = collect_reduced_image(ctx, dist, torch, rank, world, use_comm)

    frags, frags_by_kernel = count_fragments(ctx, M, sf, mode, flags)
    records = ctx.stats()          # class counts of this rank's shard (identical every frame)
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
            ok = torch.zeros(1, dtype=torch.int32)
            dist.broadcast(ok, src=0)
            dist.destroy_process_group()
            if int(ok.item()) != 1:
                sys.exit(3)
        return

    ms_per_step = elapsed / args.steps * 1e3
    shard_only = bool(args.as_shard) and world == 1        # only n_per particles are resident and drawn: every rate counts those
    n_drawn = n_per if shard_only else n_total
    value = n_drawn / (elapsed / args.steps)
    ms_median = float(np.median(step_s)) * 1e3
    config_tag = " = BASELINE.json configs[3]" if (n_total == 10**9 and args.mode == "density" and R == 1024 and not weak
                                                    and args.h_cap_px <= 0) else ""
    workload_name = (f"{n_total:.4g} dm particles{config_tag}, {args.mode}, {R}^2 buffer, camera A "
                     f"(scale {args.scale:g}), reference TestDataLoader h-law, "
                     + (f"index-range sharded x{world} ({n_per:.4g}/GPU)" if world > 1 else "whole snapshot resident on one GPU")
                     + (f", h capped at {args.h_cap_px:g} px" if args.h_cap_px > 0 else "")
                     + ", splat + " + ("RCCL image reduce + " if world > 1 else "") + "colormap")
    measured_peak = ctx.measure_read_bandwidth(4 << 30, 5)
    prof = None
    try:       # PMC passes committed under profiles/ (tools/profile_bench.sh): only when they were taken on THIS workload
        prof = json.load(open(os.path.join(ROOT, "profiles", "latest_bench_counters.json")))
        if prof.get("bench_line", {}).get("config", {}).get("workload") != workload_name:
            prof = None
    except Exception:
        prof = None
    result = {
        "metric": "particles/sec splatted to 1024^2 buffer",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_name, "total_particles": n_total,
                   "particles_per_gpu": n_per, "resolution": R, "sharding": f"index-range x{world}",
                   "pipeline": "generic" if args.generic else "three-class (stream S / four-records-per-wave strip gather N / row-uniform gather H2)",
                   "fragments_per_particle": frags / n_drawn, "frames_per_s": 1e3 / ms_per_step},
        "ms_per_step_median": ms_median, "value_at_median": n_drawn / (ms_median * 1e-3),
        "fragments_per_s": frags / (ms_per_step * 1e-3),
        "roofline": hbm_roofline(args, means, records, n_per, n_drawn, world, ms_per_step, R, channels, prof, measured_peak),
        # per GPU: a rank draws 1/N of the frame's fragments (rank 0's kernel times beside the mean share)
        "roofline_fragment": fragment_roofline({k: v / world for k, v in frags_by_kernel.items()}, means, ms_per_step),
        "kernel_ms": means,
        "setup_s": t_setup,
    }
    if args.as_shard and world == 1:
        result["as_shard"] = (f"only the index range [{first}, {first + n_per}) of the {n_total:.4g}-particle snapshot is resident and rendered "
                              f"(shard {args.as_shard}): `value`, the roofline and the per-particle figures count these {n_per} particles")
    if args.shared_device_dry_run:
        result["dry_run"] = ("ranks shared device 0 and the shard images were summed on the host instead of by RCCL: harness test "
                             "only, `value` is not a measurement")
    ok = True
    if world > 1:
        result["roofline_fragment"]["note_n_gpus"] = (f"per GPU: 1/{world} of the frame's fragments against rank 0's kernel times "
                                                      "and the step time")
        result["per_rank_kernel_ms"] = per_rank_ms
        tot = [r["total"] for r in per_rank_ms]
        result["shard_balance_max_over_mean"] = max(tot) / (sum(tot) / len(tot)) if sum(tot) > 0 else None
        result["reduce_selftest"] = selftest
        ok = ok and bool(selftest.get("ok"))
    if world > 1 and not weak and not args.headline_only:
        # the same snapshot whole on ONE GPU (rank 0's, the others wait at the barrier): what N = 1 prints as `value`, and the
        # witness of the collective: the reduced frame must equal it within the 1e-5 of SURVEY section 8e
        try:       # (a second context next to this rank's shard: the communicator stays up until every rank is done)
            one, whole_image = whole_snapshot_line(_native, mips, R, channels, local_rank, n_total, args, mode, lut, vmin, vmax)
            result["one_gpu_same_snapshot"] = one
            result["speedup_vs_1gpu_same_snapshot"] = one["ms_per_step"] / ms_per_step
            result["reduce_check"] = compare_images(reduced_image, whole_image, world)
            ok = ok and bool(result["reduce_check"]["ok"])
        except _native.BackendError as e:
            result["one_gpu_same_snapshot"] = {"error": str(e)[:200]}
            result["speedup_vs_1gpu_same_snapshot"] = None
            result["reduce_check"] = {"ok": False, "error": "the whole snapshot could not be rendered on rank 0: " + str(e)[:160], "ranks": world}
            ok = False
    extras = world == 1 and not args.generic and not args.headline_only and args.h_cap_px <= 0 and not args.as_shard
    if extras and args.mode == "density":
        whole_ms = ms_per_step
        # fragment-heavy cameras (SURVEY section 8d camera B) on the headline snapshot while it is resident, and on 1e8 below
        result["zoomed_frame"] = {"workload": f"one whole-snapshot tsp_render, density, {R}^2, camera B of SURVEY section 8d "
                                              "(tests/test_render_output.py:161-198) and the same rotation at scale 50",
                                  f"{n_total:.4g}": zoomed_lines(ctx, n_total, R, mode)}
        if ctx is not None:
            ctx.close()
        if n_total == 10**9 and R == 1024 and not weak:
            # config 3 as the 8 GPUs will see it: every index-range shard of THIS snapshot, one after another on this GPU
            result["shards_of_1e9_x8"] = shards_line(_native, mips, R, local_rank, n_total, 8, args, mode, lut, vmin, vmax, whole_ms)
        # a stand-alone 1.25e8-particle snapshot = the headline of rounds 1-3 (NOT a shard of the 1e9 snapshot: its h-law is that
        # of N = 1.25e8), and on it the opt-in kernel I and the bandwidth regime (the same positions with
        # footprints capped at 8 px isolate the streaming kernels); reported beside the headline, never as `value`
        n_sh = 125_000_000
        ctx = make_context(_native, mips, R, 2, local_rank, n_sh, 0, n_sh, args)
        result["standalone_1p25e8"] = config_line(ctx, n_sh, "density", R, args, "a stand-alone 1.25e8-particle snapshot (the headline "
                                                  "of rounds 1-3; h-law of N = 1.25e8, wider footprints than a shard of the 1e9 snapshot)",
                                                  regenerate=False)
        result["bandwidth_regime"] = hcapped_line(args, ctx, n_sh, n_sh, 0, M, sf, mode, measured_peak)
        # the other single-GPU configurations of BASELINE.json, driver-timed beside the headline (never `value`)
        result["baseline_config_1"] = config_line(ctx, 10_000_000, "weighted", R, args,
                                                  "BASELINE.json configs[1]: 1e7 particles, density-weighted quantity")
        result["baseline_config_2"] = config_line(ctx, 100_000_000, "density", R, args,
                                                  "BASELINE.json configs[2]: exactly 1e8 dm particles, density")
        result["zoomed_frame"]["1e+08"] = zoomed_lines(ctx, 100_000_000, R, mode)
        ctx.close()
        ctx = None
        c5 = _native.Context(2048, 4, device_id=local_rank)
        c5.set_kernel_mips(mips)
        result["baseline_config_4"] = config_line(c5, 50_000_000, "rgb", 2048, args,
                                                  "BASELINE.json configs[4]: 5e7 star particles, rgb, 2048^2")
        c5.close()
        # the product path: the same frames through the Visualizer / SPH / ColormapHolder surface (reference visualizer.py, sph.py:306-332)
        result["visualizer_export_frame"] = visualizer_lines(local_rank, R, args, whole_ms, n_total if (n_total == 10**9 and not weak) else None)
        result["interactive_frame"] = interactive_line(local_rank, R, args)
        result["interactive_frame_zoomed"] = interactive_line(local_rank, R, args, zoom=20.0)
        # the smallest block the progressive renderer can draw is one stratum of the load-time order (<= 3.2e7 particles, 1/8 of a
        # snapshot up to 2.5e8): the largest such snapshot at the fragment-heaviest camera measured (scale 50: 2500 fragments / particle)
        result["interactive_frame_zoomed_2p5e8_scale50"] = interactive_line(local_rank, R, args, n=250_000_000, frames=16, zoom=50.0)
    if not args.no_cpu_baseline and not args.headline_only and world == 1 and not args.as_shard:      # reported baseline: rank 0 at N = 1 only
        result["cpu_baseline"] = cpu_baseline(args, n_total, M, sf, R)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.broadcast(flag, src=0)
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)       # (a fresh exit of this process; nothing is re-executed)


def hbm_roofline(args, means, records, n_per, n_total, world, ms_per_step, R, channels, prof, measured_peak):
    """The HBM view of the frame.  First the two figures that mean something against the 8 TB/s roof: the
    FRAME (B_alg bytes per particle over the whole step) and kernel S, the one kernel that streams the particles.  Then every
    kernel with its own algorithmic bytes per launch -- S: particles x B_alg; M, H2: their records x 20 B (24 B rgb) plus one
    float64 flush of the image -- its PMC traffic (profiles/latest_bench_counters.json: 2 x FETCH_SIZE + WRITE_SIZE, as
    MI355X_MICROARCH.md prescribes for gfx950) and traffic / algorithmic.  Kernels G and H2 are NOT HBM kernels: their bound is
    the fragment rate (`roofline_fragment`); their rows are here so that wasted re-reads show."""
    b_alg = B_ALG[args.mode]
    rec_bytes = 24 if args.mode == "rgb" else 20
    image_bytes = R * R * channels * 8
    alg = {"stream": b_alg * n_per,
           "mid": int(records.get("n_mid", 0)) * rec_bytes + image_bytes,
           "huge": int(records.get("n_huge", 0)) * rec_bytes + image_bytes}
    per = {}
    for k in ("stream", "mid", "huge"):
        ms = means.get(k, 0.0)
        traffic = name = None
        if prof is not None:
            name, v = profile_entry(prof, KERNEL_SYMBOL[k], args.mode)
            if v is not None and "hbm_read_bytes_corrected" in v:
                traffic = v.get("hbm_read_bytes_corrected", 0.0) + v.get("hbm_write_bytes", 0.0)
        gbps = alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        per[k] = {"kernel": KERNEL_SYMBOL[k], "kernel_ms": ms, "algorithmic_bytes_per_launch": alg[k], "achieved": gbps,
                  "frac": gbps / HBM_PEAK_GBPS, "traffic": None if traffic is None else traffic / 1e9,
                  "traffic_over_algorithmic": None if traffic is None else traffic / max(alg[k], 1), "traffic_kernel": name}
    s_ms = means["stream"] if means.get("stream", 0.0) > 0 else means.get("total", 0.0)      # (generic pipeline: one kernel)
    s_gbps = alg["stream"] / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
    frame_gbps = b_alg * n_total / (ms_per_step * 1e-3) / 1e9 / world
    by_time = max(("stream", "mid", "huge"), key=lambda k: means.get(k, 0.0))
    traffic_known = [per[k]["traffic"] for k in ("stream", "mid", "huge") if per[k]["traffic"] is not None]
    frame_traffic = float(sum(traffic_known)) if len(traffic_known) == 3 else None
    source = None
    if prof is not None:
        source = (f"profiles/{prof.get('tag', 'latest')}_bench_counters.json = profiles/latest_bench_counters.json (committed rocprofv3 --pmc passes of this "
                  "workload, tools/profile_bench.sh), NOT measured in this run")
    # Top level = the FRAME against the HBM roof, SURVEY section 8d's definition (N x B_alg / step time): the figure the north
    # star's 40 % is about.  Kernel S -- the one kernel that streams the particles -- and every kernel's own row sit beneath it.
    return {"bound": "hbm", "kernel": "frame", "achieved": frame_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": frame_gbps / HBM_PEAK_GBPS, "traffic": frame_traffic, "traffic_unit": "GB per frame: the three splat kernels' PMC traffic summed",
            "traffic_source": source, "ms": ms_per_step, "algorithmic_bytes_per_gpu": b_alg * n_per, "north_star_target_frac": 0.40,
            "stream_kernel": {"kernel": KERNEL_SYMBOL["stream"], "achieved": s_gbps, "frac": s_gbps / HBM_PEAK_GBPS, "kernel_ms": s_ms,
                              "algorithmic_bytes_per_launch": alg["stream"], "traffic": per["stream"]["traffic"],
                              "frac_of_measured_read_peak": s_gbps / measured_peak if measured_peak else None},
            "per_kernel": per, "longest_kernel": KERNEL_SYMBOL[by_time],
            "measured_read_peak_GBps": measured_peak, "guide_measured_copy_GBps": HBM_MEASURED_COPY_GBPS,
            "note": "the frame is bound by the fragments it draws (roofline_fragment), not by bytes: kernel S is the path's one HBM-shaped "
                    "kernel (it streams every particle once), the longest kernel draws records, not particles"}


# ---- N > 1: self-validation of the one collective of the path ---------------------------------------------------------
def selftest_pattern(R, channels, rank):
    """rank-dependent image of small integers: any sum over <= 64 ranks is exact in float32"""
    j, i = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    base = ((i + 2 * j) % 7 + 1).astype(np.float32)
    return np.stack([base * (rank + 1) + c for c in range(channels)], axis=-1).astype(np.float32)


def host_sum(img, dist, torch, rank, world):
    """the shard images summed on rank 0 through the rendezvous group (stand-in for the RCCL reduce in --shared-device-dry-run)"""
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    parts = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, parts, dst=0)
    if rank != 0:
        return None
    acc = np.zeros(img.shape, dtype=np.float32)
    for q in parts:                       # rank order, float32: what ncclReduce(sum, float32) computes up to association
        acc += q.numpy()
    return acc


def reduce_selftest(ctx, dist, torch, rank, world, use_comm, R, channels):
    """Every rank writes a rank-dependent constant image, ONE reduce, rank 0 checks the exact sum."""
    ctx.write_image(selftest_pattern(R, channels, rank))
    if use_comm:
        ctx.comm_reduce_image(root=0)
        got = ctx.read_image() if rank == 0 else None
    else:
        got = host_sum(ctx.read_image(), dist, torch, rank, world)
    if rank != 0:
        return None
    want = np.zeros((R, R, channels), dtype=np.float64)
    for r in range(world):
        want += selftest_pattern(R, channels, r)
    bad = int((got.astype(np.float64) != want).sum())
    return {"ok": bad == 0, "ranks": world, "pixels_wrong": bad, "collective": "ncclReduce(sum, float32) over RCCL" if use_comm
            else "host sum over the rendezvous group (dry run: ranks share one device)"}


def collect_reduced_image(ctx, dist, torch, rank, world, use_comm):
    """the frame rank 0 presents: after the RCCL reduce it is rank 0's render target; in the dry run the host sum of the shards"""
    if use_comm:
        return ctx.read_image() if rank == 0 else None
    return host_sum(ctx.read_image(), dist, torch, rank, world)


def compare_images(reduced, whole, world, rtol=1e-5):
    """reduced frame of N shards against the same snapshot whole on one GPU: 1e-5 relative per pixel"""
    a, b = reduced.astype(np.float64), whole.astype(np.float64)
    lit = b != 0
    rel = np.abs(a - b)[lit] / np.abs(b[lit])
    dark_ok = bool((a[~lit] == 0).all())
    max_rel = float(rel.max()) if rel.size else 0.0
    return {"ok": bool(max_rel <= rtol and dark_ok and lit.any()), "max_rel": max_rel, "rtol": rtol, "ranks": world,
            "pixels_compared": int(lit.sum()), "zero_pixels_stay_zero": dark_ok}


def whole_snapshot_line(_native, mips, R, channels, device, n_total, args, mode, lut, vmin, vmax, frames=5):
    """The whole snapshot on one GPU, timed like the headline (wall clock around render + colormap)."""
    c = make_context(_native, mips, R, channels, device, n_total, 0, n_total, args, 0.0, args.mode)
    M_, sf_ = camera(args.scale)

    rgba = np.empty((R, R, 4), dtype=np.uint8)

    def one():
        c.render(M_, sf_, clear=True, mode=mode)
        if args.mode == "rgb":
            c.colormap_rgb(vmin, vmax, 1.0)
        else:
            c.colormap_scalar(lut, vmin, vmax, True, args.mode == "weighted", out=rgba)

    one()
    t = time.perf_counter()
    for _ in range(frames):
        one()
    ms = (time.perf_counter() - t) / frames * 1e3
    st = c.stats()
    image = c.read_image()
    c.close()
    return {"workload": f"the same {n_total:.4g}-particle snapshot whole on one GPU", "ms_per_step": ms,
            "value": n_total / (ms * 1e-3), "unit": "particles/s", "frames": frames,
            "kernel_ms": {k: st["ms_" + k] for k in KERNELS}}, image


def shards_line(_native, mips, R, device, n_total, G, args, mode, lut, vmin, vmax, whole_ms, frames=6):
    """BASELINE config 3 as its 8 GPUs will see it, on the one GPU at hand: every index-range shard [g N / G, (g + 1) N / G)
    of THE snapshot (split_buffers.py:26-38) is generated, ordered and rendered in turn -- no fragment counting, wall clock around
    the synchronous tsp_render as in the headline -- and the 8-GPU frame is projected as the slowest shard + one image reduce
    (an estimate: REDUCE_ESTIMATE_MS) + the colormap on the root."""
    M, sf = camera(args.scale)
    ctx = _native.Context(R, 2, device_id=device)
    ctx.set_kernel_mips(mips)
    per, cmap_ms = [], None
    for g in range(G):
        first, cnt = shard_range(n_total, G, g)
        ctx.generate_synthetic(n_total, first=first, count=cnt, seed=1337, h_cap=0.0)
        if not args.no_reorder:
            ctx.reorder_spatial(num_strata(cnt), 1337)
        ms, kms = [], {k: [] for k in KERNELS}
        for i in range(frames + 2):
            t = time.perf_counter()
            ctx.render(M, sf, clear=True, mode=mode)
            dt = (time.perf_counter() - t) * 1e3
            if i >= 2:
                ms.append(dt)
                st = ctx.stats()
                for k in KERNELS:
                    kms[k].append(st["ms_" + k])
        if cmap_ms is None:
            ts, rgba = [], np.empty((R, R, 4), dtype=np.uint8)
            for _ in range(6):
                t = time.perf_counter()
                ctx.colormap_scalar(lut, vmin, vmax, True, False, out=rgba)
                ts.append((time.perf_counter() - t) * 1e3)
            cmap_ms = float(np.median(ts[1:]))
        st = ctx.stats()
        per.append({"shard": g, "first": first, "particles": cnt, "ms_per_step": float(np.median(ms)),
                    "kernel_ms": {k: float(np.median(v)) for k, v in kms.items()},
                    "records": {"small": int(st["n_small"]), "mid": int(st["n_mid"]), "huge": int(st["n_huge"]), "culled": int(st["n_culled"])}})
    ctx.close()
    t = [x["ms_per_step"] for x in per]
    slowest, mean = max(t), sum(t) / len(t)
    step8 = slowest + REDUCE_ESTIMATE_MS + cmap_ms
    return {"workload": f"the {G} index-range shards of the {n_total:.4g}-particle snapshot (what each of {G} GPUs holds), one after "
                        f"another on this GPU, {R}^2, camera A", "per_shard": per, "max_shard_ms": slowest, "mean_shard_ms": mean,
            "max_over_mean": slowest / mean, "sum_of_shards_ms": sum(t), "colormap_ms": cmap_ms,
            "reduce_estimate_ms": REDUCE_ESTIMATE_MS, "whole_snapshot_on_one_gpu_ms": whole_ms,
            "projected_ms_per_step_8gpu": step8, "projected_speedup_1to8": whole_ms / step8,
            "projected_frames_per_s_8gpu": 1e3 / step8, "north_star_target_speedup": 6.0,
            "note": "projection = whole-snapshot step on one GPU / (slowest shard + reduce estimate + colormap); the reduce term is an "
                    "estimate (no multi-GPU box was available to this build), everything else is measured here"}


def visualizer_lines(device, R, args, whole_ms, n_big=None, frames=5):
    """The drop-in path: Visualizer.get_sph_presentation_image() = SPH.render(EXPORT) + ColormapHolder pass (reference
    visualizer.py:456-474, sph.py:306-332) on a device-generated snapshot, beside the bare C-ABI frame (tsp_render of everything +
    tsp_colormap_scalar) on the SAME context.  EXPORT is one block on this backend (config.MAX_PARTICLES_PER_EXPORT_RENDERCALL)."""
    import topsy_amd
    from topsy_amd import DrawReason
    out = []
    for n in [100_000_000] + ([n_big] if n_big else []):
        try:
            t0 = time.perf_counter()
            vis = topsy_amd.synthetic_on_device(n, render_resolution=R, device_id=device)
            setup = time.perf_counter() - t0
            vis.get_sph_presentation_image()
            ts = []
            for _ in range(frames):
                t = time.perf_counter()
                img = vis.get_sph_presentation_image()
                ts.append((time.perf_counter() - t) * 1e3)
            blocks = getattr(vis._sph, "last_render_blocks", None)
            ctx = vis.particle_buffers.context
            M, sf = vis._sph._transform
            params = vis.colormap.get_parameters()
            import matplotlib
            lut = matplotlib.colormaps[params["colormap_name"]](np.linspace(0.001, 0.999, 1000)).astype(np.float32)
            tc, rgba_c = [], np.empty((R, R, 4), dtype=np.uint8)
            for _ in range(frames + 1):
                t = time.perf_counter()
                ctx.render(M, sf, clear=True, mode=vis._sph._mode)
                ctx.colormap_scalar(lut, float(params["vmin"]), float(params["vmax"]), bool(params["log"]), False, out=rgba_c)
                tc.append((time.perf_counter() - t) * 1e3)
            vis_ms, cabi_ms = float(np.median(ts)), float(np.median(tc[1:]))
            line = {"particles": n, "visualizer_export_frame_ms": vis_ms, "c_abi_frame_ms": cabi_ms, "ratio": vis_ms / cabi_ms,
                    "frames_per_s": 1e3 / vis_ms, "setup_s": setup, "image_shape": list(img.shape), "image_dtype": str(img.dtype)}
            if blocks is not None:
                line["render_blocks_per_frame"] = blocks
            if n == n_big:
                line["headline_ms_per_step"] = whole_ms
                line["ratio_to_headline"] = vis_ms / whole_ms
            vis.close()
            out.append(line)
        except Exception as e:       # (never lose the bench line to an extra)
            out.append({"particles": n, "error": f"{type(e).__name__}: {e}"[:240]})
    return {"workload": "Visualizer.get_sph_presentation_image(): SPH.render(EXPORT) + colormap through the reference's object "
                        f"protocol, density, {R}^2, camera A, beside tsp_render + tsp_colormap_scalar on the same context", "sizes": out}


def interactive_line(device, R, args, n=1_000_000_000, frames=24, zoom=None):
    """The regime the reference is built around (config.py:6-7: 30 frames/s, progressive blocks): Visualizer.draw(CHANGE) on the
    1e9-particle snapshot -- one time-budgeted block + colormap per frame -- until the adaptive block size has settled: how many
    particles a 1/30 s frame draws, and how long the frame really takes."""
    import topsy_amd
    from topsy_amd import DrawReason, config
    try:
        vis = topsy_amd.synthetic_on_device(n, render_resolution=R, device_id=device)
        if zoom is not None:          # camera B (reference tests/test_render_output.py:161-198)
            vis.scale = zoom
            vis.rotate(0.0, 0.4)
        drawn, ms = [], []
        for i in range(frames):
            t = time.perf_counter()
            vis.draw(DrawReason.CHANGE)
            ms.append((time.perf_counter() - t) * 1e3)
            drawn.append(n / vis._sph.last_render_mass_scale)
        # REFINE frames continue from the cursor until the snapshot is complete
        refine = 0
        t = time.perf_counter()
        while vis._sph.needs_refine() and refine < 1000:
            vis.draw(DrawReason.REFINE)
            refine += 1
        refine_ms = (time.perf_counter() - t) * 1e3
        vis.close()
        tail = slice(frames // 2, None)
        cam_name = "camera A" if zoom is None else f"camera B (scale {zoom:g}, rotate(0, 0.4))"
        return {"workload": f"Visualizer.draw(CHANGE) on the {n:.4g}-particle snapshot, {R}^2, {cam_name}: one progressive block of "
                            f"the 1/{config.TARGET_FPS} s budget + colormap per frame (progressive_render.py:48-86)",
                "particles_per_frame": float(np.median(drawn[tail])), "ms_per_frame": float(np.median(ms[tail])),
                "frames_per_s": 1e3 / float(np.median(ms[tail])), "target_frames_per_s": config.TARGET_FPS,
                "fraction_of_snapshot_per_frame": float(np.median(drawn[tail])) / n,
                "first_frame_particles": drawn[0], "first_frame_ms": ms[0],
                "slowest_frame_ms": float(np.max(ms[1:])), "frames_over_budget": int(np.sum(np.asarray(ms[1:]) > 1e3 / config.TARGET_FPS)),
                "frames_timed": frames - 1, "smallest_block_particles": float(np.min(drawn)),
                "refine_frames_to_complete": refine, "refine_ms_to_complete": refine_ms}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:240]}


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


def zoomed_lines(ctx, n, R, mode, frames=4):
    """Fragment-heavy cameras on a resident snapshot (SURVEY section 8d camera B: scale 20, rotate(0, 0.4); and the same rotation
    at scale 50): per-kernel times, record counts and fragments per particle of one whole-snapshot tsp_render."""
    out = []
    for scale in (20.0, 50.0):
        try:
            M, sf = camera_b(scale)
            ms, per = [], {k: [] for k in KERNELS}
            for i in range(frames + 1):
                t = ctx.render(M, sf, clear=True, mode=mode)
                if i:
                    ms.append(t)
                    st = ctx.stats()
                    for k in KERNELS:
                        per[k].append(st["ms_" + k])
            frags, by_kernel = count_fragments(ctx, M, sf, mode)
            st = ctx.stats()
            med = float(np.median(ms))
            out.append({"camera": f"B: scale {scale:g}, rotate(0, 0.4)", "particles": n, "ms_per_step": med, "frames_per_s": 1e3 / med,
                        "value": n / (med * 1e-3), "unit": "particles/s", "kernel_ms": {k: float(np.median(v)) for k, v in per.items()},
                        "records": {k: int(st["n_" + k]) for k in ("small", "mid", "huge", "culled")},
                        "fragments_per_particle": frags / n, "fragments_per_s": frags / (med * 1e-3),
                        "fragments_by_kernel": {k: int(v) for k, v in by_kernel.items()}})
        except Exception as e:       # (never lose the bench line to an extra)
            out.append({"camera": f"B: scale {scale:g}, rotate(0, 0.4)", "particles": n, "error": f"{type(e).__name__}: {e}"[:240]})
    return out


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
