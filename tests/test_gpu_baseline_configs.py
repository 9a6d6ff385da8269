"""BASELINE.json's configurations at FULL size on the default pipeline, against the CPU oracle directly (not against the
library's own generic kernel) wherever the oracle finishes in seconds on the GPU box's host cores, and config 3 (the
1e9-particle snapshot in 8 index-range shards) through its sharding contract:

  configs[1]  1e7 particles, density-weighted quantity, 1024^2 ........ vs oracle: density 1e-5 relative per pixel, weighted
                                                                         channel within 1e-5 of its sum of |terms|, exact
                                                                         fragment count
  configs[2/3] a 4e7-particle index range of the 1e9 snapshot, 1024^2 .. vs oracle: 1e-5 relative, exact fragment count
  configs[3]  1e9 particles = 8 shards of 1.25e8, 1024^2 ............... sum of the eight float32 shard images == the whole
                                                                         snapshot resident on one GPU (1e-5), every particle
                                                                         accounted for once, fragment totals equal
  configs[4]  rgb, 2048^2, 5e6 star particles ........................... vs oracle: colour channels 1e-5, count channel and
                                                                         fragment count exact

What does NOT meet the oracle at full size: configs[2] at exactly 1e8 particles (mass / shard-additivity properties) and
configs[4] at 5e7 particles (against the library's own generic kernel) -- tests/test_gpu_scale.py; the oracle sees those
configurations at the sizes listed above.

Tolerances: the reference's own tests (tests/test_render_output.py:161-241) allow atol 1.5e-7 on a weighted image whose
values are ~1e-5 (1.5e-2 relative) and ~1 % statistical agreement on the density image; north_star asks 1e-5 relative on the
float buffer, which is what is asserted here.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from topsy_amd import _native
    _native.load_library()
    return _native


def camera(scale):
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] /= scale
    M[2, :] = [0.0, 0.0, 0.5 / scale, 0.5]
    return M, np.float32(1.0 / scale)


def max_rel(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    den = np.maximum(np.abs(a), np.abs(b))
    lit = den > 0
    return float((np.abs(a - b)[lit] / den[lit]).max()) if lit.any() else 0.0


def num_strata(n):
    from topsy_amd.particle_buffers import ParticleBuffers
    return ParticleBuffers._num_strata(n)


def test_config1_1e7_weighted_against_the_oracle(native, mips):
    """BASELINE configs[1] verbatim: 1e7 gas particles, density-weighted quantity, 1024^2, camera A, default pipeline
    with the load-time ordering, against oracle/oracle.c on the same downloaded arrays."""
    from oracle import oracle_c
    n, R = 10_000_000, 1024
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True)
    ctx.reorder_spatial(num_strata(n), 1337)
    d = ctx.download_particles(("x", "y", "z", "h", "mass", "q"))
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED)
    st = ctx.stats()
    got = ctx.read_image()
    ctx.close()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    assert min(st["n_small"], st["n_mid"], st["n_huge"]) > 0 and st["n_mega"] == 0, "kernels S, M and H2 take part"
    assert st["n_fragments_stream"] + st["n_fragments_mid"] + st["n_fragments_huge"] + st["n_fragments_mega"] == st["n_fragments"]
    t = time.time()
    want, nfrag = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], d["q"], mode=0, M=M, sf=float(sf), R=R, mips=mips)
    scale, _ = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], np.abs(d["q"]), mode=0, M=M, sf=float(sf), R=R, mips=mips)
    print(f"oracle: 2 x {nfrag:.3g} fragments in {time.time() - t:.1f} s on {oracle_c.max_threads()} threads")
    assert st["n_fragments"] == nfrag, "coverage decisions differ from the oracle"
    assert max_rel(got[..., 0], want[..., 0]) <= 1e-5
    # the weighted channel cancels (signed q): within 1e-5 of the sum of |terms|
    assert (np.abs(got[..., 1].astype(np.float64) - want[..., 1]) <= 1e-5 * scale[..., 1].astype(np.float64) + 1e-30).all()


def test_sample_of_the_1e9_snapshot_against_the_oracle(native, mips):
    """The headline workload's own particles: the index range [0, 4e7) of the 1e9-particle snapshot (a uniform sample: the
    generator's index bijection), density, 1024^2, camera A, through S + M + H2 against the oracle."""
    from oracle import oracle_c
    n_total, n, R = 10**9, 40_000_000, 1024
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n_total, 0, n, 1337, 0.0)
    ctx.reorder_spatial(num_strata(n), 1337)
    d = ctx.download_particles(("x", "y", "z", "h", "mass"))
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    st = ctx.stats()
    got = ctx.read_image()
    ctx.close()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    assert min(st["n_small"], st["n_mid"], st["n_huge"]) > 0 and st["n_mega"] == 0
    t = time.time()
    want, nfrag = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], None, mode=0, M=M, sf=float(sf), R=R, mips=mips)
    print(f"oracle: {nfrag:.3g} fragments in {time.time() - t:.1f} s on {oracle_c.max_threads()} threads")
    assert st["n_fragments"] == nfrag, "coverage decisions differ from the oracle"
    assert max_rel(got[..., 0], want[..., 0]) <= 1e-5
    assert (got[..., 1] == 0).all()      # density render: q = 0 (particle_buffers.py:96-99)


def test_config4_mode_rgb_2048_against_the_oracle(native, mips):
    """BASELINE configs[4]'s mode and buffer (rgb, 2048^2, camera A) with 5e6 star particles -- a tenth of the config's
    5e7, which would cost the oracle minutes -- against the oracle: colour channels 1e-5, count channel exact.
    (The full 5e7 run is compared with the generic kernel in test_gpu_scale.py::test_config5_rgb_2048_full_size.)"""
    from oracle import oracle_c
    n, R = 5_000_000, 2048
    M, sf = camera(200.0)
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=False, with_rgb=True)
    ctx.reorder_spatial(num_strata(n), 1337)
    d = ctx.download_particles(("x", "y", "z", "h", "r", "g", "b"))
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=native.MODE_RGB)
    st = ctx.stats()
    got = ctx.read_image()
    ctx.close()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    t = time.time()
    want, nfrag = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["r"], d["g"], d["b"], mode=2, M=M, sf=float(sf), R=R, mips=mips)
    print(f"oracle: {nfrag:.3g} fragments in {time.time() - t:.1f} s on {oracle_c.max_threads()} threads")
    assert st["n_fragments"] == nfrag, "coverage decisions differ from the oracle"
    assert np.array_equal(got[..., 3], want[..., 3]), "fragment-count channel must be exact"
    for c in range(3):
        assert max_rel(got[..., c], want[..., c]) <= 1e-5, f"channel {c}"


def test_config3_1e9_in_eight_index_range_shards(native, mips):
    """BASELINE configs[3] minus RCCL: the 1e9-particle snapshot cut into the 8 index ranges the 8 GPUs hold
    (split_buffers.py:26-38 arithmetic), each rendered on its own (own load-time ordering, as each rank does), the eight
    float32 images summed in rank order (what ncclReduce(sum, float32) computes) == the whole snapshot resident on one
    GPU, within 1e-5 relative per pixel; every particle is accounted for exactly once on both sides and the fragment totals
    are equal as integers."""
    n, G, R = 10**9, 8, 1024
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0)
    ctx.reorder_spatial(num_strata(n), 1337)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    st = ctx.stats()
    whole = ctx.read_image()[..., 0].copy()
    ctx.close()
    assert st["n_particles"] == n
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    mass = whole.astype(np.float64).sum() * (2 * 200.0 / R) ** 2
    assert 0.97 * n * 1e-8 < mass < 1.01 * n * 1e-8
    total = np.zeros((R, R), dtype=np.float32)
    seen = frags = 0
    slowest = 0.0
    c = native.Context(R, 2)
    c.set_kernel_mips(mips)
    c.set_option("count_fragments", 1)
    for g in range(G):
        first = (n * g) // G
        count = (n * (g + 1)) // G - first
        c.generate_synthetic(n, first, count, 1337, 0.0)
        c.reorder_spatial(num_strata(count), 1337)
        ms = c.render(M, sf)
        s = c.stats()
        assert s["n_particles"] == count == 125_000_000
        seen += s["n_small"] + s["n_mid"] + s["n_huge"] + s["n_culled"]
        frags += s["n_fragments"]
        slowest = max(slowest, ms)
        total += c.read_image()[..., 0]                 # float32 adds in rank order
    c.close()
    assert seen == n
    assert frags == st["n_fragments"], "the shards draw exactly the fragments of the whole snapshot"
    assert max_rel(total, whole) <= 1e-5
    print(f"whole snapshot {st['ms_total']:.1f} ms on one GPU; slowest shard {slowest:.1f} ms (with fragment counting on)")
