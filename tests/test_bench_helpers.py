"""bench.py's host-side arithmetic (no GPU): the strong-scaling shard ranges, the fragment-term roofline and the lookup of a
kernel's PMC entry in profiles/*_bench_counters.json."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_shard_ranges_partition_the_snapshot():
    # the reference's SplitBuffers._calculate_splits arithmetic (split_buffers.py:26-38): contiguous, exhaustive, sizes within 1
    for n in (10**9, 1_000_003, 7, 0):
        for world in (1, 2, 3, 8):
            parts = [bench.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert bench.shard_range(10**9, 8, 3) == (375_000_000, 125_000_000)         # config 3: 8 shards of 1.25e8


def test_fragment_roofline_arithmetic():
    frags = {"stream": 1e10, "mid": 2e10, "huge": 1e11, "mega": 1.5e11}
    ms = {"stream": 10.0, "mid": 20.0, "huge": 25.0, "mega": 15.0}
    r = bench.fragment_roofline(frags, ms, 70.0)
    flop = (1e10 + 2e10) * 1 * 2 + (1e11 + 1.5e11) * 2 * 2
    assert np.isclose(r["flop_per_frame"], flop)
    assert np.isclose(r["achieved"], flop / 70e-3 / 1e12) and np.isclose(r["frac"], r["achieved"] / bench.VALU_F32_PEAK_TFLOPS)
    assert np.isclose(r["per_kernel"]["mega"]["achieved"], 1.5e11 * 4 / 15e-3 / 1e12)
    assert r["per_kernel"]["stream"]["fmas_per_fragment"] == 1 and r["per_kernel"]["huge"]["fmas_per_fragment"] == 2
    zero = bench.fragment_roofline({}, {}, 0.0)
    assert zero["achieved"] == 0.0 and zero["per_kernel"]["mid"]["fragments"] == 0


def test_profile_entry_picks_the_timed_instantiation():
    prof = {"per_kernel": {
        "tsp::splat_huge2_kernel<0, 1, 1, 32, 8, false>": {"x": 1},
        "tsp::splat_huge2_kernel<0, 1, 1, 32, 8, true>": {"x": 2},        # the one counting frame
        "tsp::splat_huge2_kernel<0, 2, 1, 16, 7, false>": {"x": 3},       # weighted
        "tsp::splat_huge2_kernel<2, 3, 1, 16, 5, false>": {"x": 4},       # rgb
        "tsp::splat_mega64_kernel<0, 4, false>": {"x": 5},
        "tsp::splat_mega64_kernel<0, 4, true>": {"x": 6},
        "tsp::splat_stream_kernel<0, 1>": {"x": 7}, "tsp::splat_stream_kernel<0, 2>": {"x": 8}}}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "density")[1] == {"x": 1}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "weighted")[1] == {"x": 3}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "rgb")[1] == {"x": 4}
    assert bench.profile_entry(prof, "splat_mega64_kernel", "density")[1] == {"x": 5}
    assert bench.profile_entry(prof, "splat_stream_kernel", "weighted")[1] == {"x": 8}
    assert bench.profile_entry(prof, "splat_mid_kernel", "density") == (None, None)
    # two candidates that cannot be told apart: no guess
    prof["per_kernel"]["tsp::splat_huge2_kernel<0, 1, 1, 16, 8, false>"] = {"x": 9}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "density") == (None, None)
