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
    frags = {"stream": 1e10, "mid": 2e10, "huge": 2.5e11}
    ms = {"stream": 10.0, "mid": 20.0, "huge": 40.0}
    r = bench.fragment_roofline(frags, ms, 70.0)
    flop = (1e10 + 2e10) * 1 * 2 + 2.5e11 * 2 * 2
    assert np.isclose(r["flop_per_frame"], flop)
    assert np.isclose(r["achieved"], flop / 70e-3 / 1e12) and np.isclose(r["frac"], r["achieved"] / bench.VALU_F32_PEAK_TFLOPS)
    assert np.isclose(r["per_kernel"]["huge"]["achieved"], 2.5e11 * 4 / 40e-3 / 1e12)
    assert r["per_kernel"]["stream"]["fmas_per_fragment"] == 1 and r["per_kernel"]["huge"]["fmas_per_fragment"] == 2
    zero = bench.fragment_roofline({}, {}, 0.0)
    assert zero["achieved"] == 0.0 and zero["per_kernel"]["mid"]["fragments"] == 0


def test_profile_entry_picks_the_timed_instantiation():
    prof = {"per_kernel": {
        "tsp::splat_huge2_kernel<0, 1, 1, 32, 8, false>": {"x": 1},
        "tsp::splat_huge2_kernel<0, 1, 1, 32, 8, true>": {"x": 2},        # the one counting frame
        "tsp::splat_huge2_kernel<0, 2, 1, 16, 7, false>": {"x": 3},       # weighted
        "tsp::splat_huge2_kernel<2, 3, 1, 16, 5, false>": {"x": 4},       # rgb
        "tsp::splat_stream_kernel<0, 1>": {"x": 7}, "tsp::splat_stream_kernel<0, 2>": {"x": 8}}}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "density")[1] == {"x": 1}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "weighted")[1] == {"x": 3}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "rgb")[1] == {"x": 4}
    assert bench.profile_entry(prof, "splat_stream_kernel", "weighted")[1] == {"x": 8}
    assert bench.profile_entry(prof, "splat_mid_gather_kernel", "density") == (None, None)
    # two candidates that cannot be told apart: no guess
    prof["per_kernel"]["tsp::splat_huge2_kernel<0, 1, 1, 16, 8, false>"] = {"x": 9}
    assert bench.profile_entry(prof, "splat_huge2_kernel", "density") == (None, None)


def test_hbm_roofline_prices_every_kernel_with_its_own_bytes():
    class A: mode = "density"
    means = {"stream": 17.0, "mid": 16.0, "huge": 33.0, "total": 66.5}
    records = {"n_mid": 24_905_644, "n_huge": 4_232_789}
    prof = {"per_kernel": {"tsp::splat_stream_kernel<0, 1>": {"hbm_read_bytes_corrected": 20.06e9, "hbm_write_bytes": 1.49e9},
                           "tsp::splat_huge2_kernel<0, 1, 1, 32, 8, false>": {"hbm_read_bytes_corrected": 12.73e9, "hbm_write_bytes": 11.08e9}}}
    r = bench.hbm_roofline(A, means, records, 10**9, 10**9, 1, 67.3, 1024, 2, prof, 6660.0)
    # top level = the frame against the HBM roof (N x B_alg / step time); kernel S beneath it
    assert r["kernel"] == "frame" and r["bound"] == "hbm" and r["longest_kernel"] == "splat_huge2_kernel"
    assert np.isclose(r["achieved"], 20e9 / 67.3e-3 / 1e9) and np.isclose(r["frac"], r["achieved"] / 8000.0)
    sk = r["stream_kernel"]
    assert sk["kernel"] == "splat_stream_kernel" and np.isclose(sk["achieved"], 20e9 / 17e-3 / 1e9) and np.isclose(sk["traffic"], 21.55)
    assert "NOT measured in this run" in r["traffic_source"]
    h = r["per_kernel"]["huge"]
    assert h["algorithmic_bytes_per_launch"] == 4_232_789 * 20 + 1024 * 1024 * 2 * 8
    assert np.isclose(h["traffic"], 23.81) and h["traffic_over_algorithmic"] > 200
    assert r["per_kernel"]["mid"]["traffic"] is None          # no PMC entry: null, never a guess
    assert r["traffic"] is None                               # the frame's traffic needs all three kernels' entries


def test_reduce_selftest_pattern_sums_exactly_and_images_compare():
    want = sum(bench.selftest_pattern(64, 2, r).astype(np.float64) for r in range(8))
    acc = np.zeros((64, 64, 2), dtype=np.float32)
    for r in range(8):
        acc += bench.selftest_pattern(64, 2, r)
    assert np.array_equal(acc.astype(np.float64), want)
    whole = np.random.RandomState(0).uniform(0.0, 1.0, (64, 64, 2)).astype(np.float32)
    whole[:8] = 0.0
    ok = bench.compare_images(whole * np.float32(1.000001), whole, 8)
    assert ok["ok"] and ok["max_rel"] < 1e-5 and ok["ranks"] == 8
    bad = whole.copy(); bad[20, 20, 0] *= 1.001
    assert not bench.compare_images(bad, whole, 8)["ok"]
    dust = whole.copy(); dust[0, 0, 0] = 1e-30
    assert not bench.compare_images(dust, whole, 8)["ok"]
