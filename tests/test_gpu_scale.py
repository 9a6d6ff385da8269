"""Full-size checks through size-independent properties (the oracle cannot run 1e8 particles in a
test): pipeline == generic kernel, mass conservation, shard additivity (the multi-GPU contract),
order invariance under the load-time reordering, block additivity, and the RCCL entry points."""
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


def rel_close(a, b, rtol):
    return (np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b)) + 1e-30).all()


def test_pipeline_equals_generic_and_oracle_1e6(native, mips):
    """1e6-particle reference snapshot (BASELINE config 1 size), 512^2: the three-class pipeline, the
    generic atomic kernel and the CPU oracle agree; every particle is accounted for exactly once."""
    from oracle import oracle_c
    n, R = 1000000, 512
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True)
    d = ctx.download_particles(("x", "y", "z", "h", "mass", "q"))
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    a = ctx.read_image()
    st = ctx.stats()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    want, nfrag = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], d["q"], mode=0, M=M, sf=float(sf), R=R, mips=mips)
    assert st["n_fragments"] == nfrag, "coverage decisions differ from the oracle"
    assert rel_close(a[..., 0], want[..., 0], 1e-5)
    scale, _ = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], np.abs(d["q"]), mode=0, M=M, sf=float(sf), R=R, mips=mips)
    assert (np.abs(a[..., 1] - want[..., 1]) <= 1e-5 * scale[..., 1] + 1e-30).all()
    ctx.render(M, sf, flags=native.PIPE_GENERIC)
    g = ctx.read_image()
    assert ctx.stats()["n_fragments"] == nfrag
    # (both paths accumulate the render target in float64)
    assert rel_close(g[..., 0], want[..., 0], 1e-5)
    # load-time reordering changes nothing but the summation order
    ctx.reorder_spatial(32, 7)
    ctx.render(M, sf)
    b = ctx.read_image()
    assert ctx.stats()["n_fragments"] == nfrag
    assert rel_close(b[..., 0], want[..., 0], 1e-5)
    ctx.close()


@pytest.mark.parametrize("n,hcap_px", [(20000000, 0.0), (100000000, 8.0), (100000000, 0.0)])
def test_mass_and_shard_additivity_at_scale(native, mips, n, hcap_px):
    """sum over pixels * pixel area == visible mass; image(shard A) + image(shard B) == image(all)
    (index-range shards, BASELINE config 4's contract), up to float32 summation noise.
    (100000000, 0.0) is BASELINE config 3 at full size with the reference h-law (uncapped)."""
    R, scale = 1024, 200.0
    M, sf = camera(scale)
    hcap = hcap_px * scale / (2.0 * R)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, hcap)
    ctx.reorder_spatial(32, 1337)
    ctx.render(M, sf)
    full = ctx.read_image()[..., 0].astype(np.float64)
    st = ctx.stats()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    mass = full.sum() * (2 * scale / R) ** 2
    # particles outside the z-slab/viewport and sub-pixel footprints that miss every pixel centre are not
    # drawn (reference semantics), the rest conserves mass: kernel mips are normalised to unit integral
    assert 0.97 * n * 1e-8 < mass < 1.01 * n * 1e-8
    # two index-range blocks of the SAME resident set, accumulated without clearing == one block
    half = n // 2
    ctx.render(M, sf, np.array([0]), np.array([half]), clear=True)
    ctx.render(M, sf, np.array([half]), np.array([n - half]), clear=False)
    two = ctx.read_image()[..., 0].astype(np.float64)
    assert rel_close(two, full, 1e-5)
    ctx.close()
    # two separately generated shards (what two ranks hold) add up to the full image
    parts = np.zeros((R, R), dtype=np.float64)
    for first, count in ((0, half), (half, n - half)):
        c = native.Context(R, 2)
        c.set_kernel_mips(mips)
        c.generate_synthetic(n, first, count, 1337, hcap)
        c.reorder_spatial(32, 1337)
        c.render(M, sf)
        parts += c.read_image()[..., 0]
        c.close()
    assert rel_close(parts, full, 1e-5)


def test_linearity_in_mass_and_rotation_symmetry(native, mips):
    n, R = 2000000, 256
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 99, 0.0)
    d = ctx.download_particles(("x", "y", "z", "h", "mass"))
    M, sf = camera(60.0)
    ctx.render(M, sf)
    a = ctx.read_image()[..., 0].copy()
    ctx.upload_particles(d["x"], d["y"], d["z"], d["h"], d["mass"] * np.float32(4.0))   # exact scaling
    ctx.render(M, sf)
    b = ctx.read_image()[..., 0]
    assert rel_close(b, 4.0 * a, 1e-5)        # summation order differs between runs (atomics)
    # 90 degree rotation about z == transpose + flip of the image (reference test_rotated_sph_output)
    rot = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    M2 = M.copy()
    M2[:3, :3] = (np.diag([1, 1, 0.5]) @ rot / 60.0).astype(np.float32)
    ctx.render(M2, sf)
    c = ctx.read_image()[..., 0]
    assert rel_close(b.T[:, ::-1], c, 1e-3)
    ctx.close()


def test_rgb_and_weighted_at_scale(native, mips):
    """5e6-particle rgb (BASELINE config 5 mode) and weighted (config 2 mode): pipeline vs generic."""
    n, R = 5000000, 1024
    M, sf = camera(200.0)
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 5, 0.0, with_quantity=True, with_rgb=True)
    ctx.reorder_spatial(32, 5)
    ctx.set_option("count_fragments", 1)
    for mode in (native.MODE_RGB, native.MODE_WEIGHTED, native.MODE_DEPTH):
        ctx.render(M, sf, mode=mode)
        a = ctx.read_image().astype(np.float64)
        fa = ctx.stats()["n_fragments"]
        ctx.render(M, sf, mode=mode, flags=native.PIPE_GENERIC)
        g = ctx.read_image().astype(np.float64)
        assert ctx.stats()["n_fragments"] == fa
        assert rel_close(a[..., 0], g[..., 0], 1e-5)
        if mode == native.MODE_RGB:
            assert np.array_equal(a[..., 3], g[..., 3])          # fragment counter channel is exact
            assert rel_close(a[..., 1:3], g[..., 1:3], 1e-5)
        else:
            tol = 1e-5 * np.abs(g[..., 1]).max()
            assert np.abs(a[..., 1] - g[..., 1]).max() <= tol
    ctx.close()


def test_rccl_single_rank_reduce(native, mips):
    """RCCL entry points on one GPU: communicator of size 1, in-place reduce leaves the image intact."""
    ctx = native.Context(128, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(100000, 0, 100000, 3, 0.0)
    M, sf = camera(100.0)
    ctx.render(M, sf)
    before = ctx.read_image()
    assert ctx.comm_reduce_image(0) == 0.0            # no communicator: single GPU, nothing to do
    uid = native.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(1, 0, uid)
    ms = ctx.comm_reduce_image(0)
    assert ms >= 0.0
    assert np.array_equal(ctx.read_image(), before)
    with pytest.raises(native.BackendError, match="already reduced"):
        ctx.comm_reduce_image(0)                       # one reduce per frame: a repeat would double-count on the root
    ctx.render(M, sf)                                  # a render rebuilds the local partial image ...
    ctx.comm_reduce_image(-1)                          # ... which may be reduced again (all-reduce form)
    assert rel_close(ctx.read_image(), before, 1e-6)   # atomics: the summation order differs between renders
    ctx.render(M, sf, [0], [50000])
    ctx.render(M, sf, [50000], [50000], clear=False)   # a REFINE-style block also re-arms the reduce
    ctx.comm_reduce_image(0)
    assert np.allclose(ctx.read_image(), before, rtol=1e-6, atol=0)
    with pytest.raises(native.BackendError):
        ctx.comm_init(1, 0, uid)                       # already initialised
    ctx.close()


def test_render_ranges_are_clipped_without_overflow(native, mips):
    """tsp_render clips (start, len) to [0, n) with saturating arithmetic: a huge len means "to the end", a range
    that starts before 0 keeps its in-range part, negative lengths are rejected."""
    ctx = native.Context(96, 2)
    ctx.set_kernel_mips(mips)
    n = 50000
    ctx.generate_synthetic(n, 0, n, 5, 0.0)
    M, sf = camera(120.0)
    ctx.render(M, sf)
    full = ctx.read_image()
    i64max = np.iinfo(np.int64).max
    ctx.render(M, sf, [0], [i64max])
    assert rel_close(ctx.read_image(), full, 1e-6)
    ctx.render(M, sf, [20000], [i64max])
    tail = ctx.read_image()
    ctx.render(M, sf, [20000], [n - 20000])
    assert rel_close(ctx.read_image(), tail, 1e-6)
    ctx.render(M, sf, [-5, i64max - 3, n + 7], [20005, i64max, 10])     # [-5, 20000) -> [0, 20000); the others are empty
    head = ctx.read_image()
    ctx.render(M, sf, [0], [20000])
    assert rel_close(ctx.read_image(), head, 1e-6)
    assert ctx.stats()["n_particles"] == 20000
    with pytest.raises(native.BackendError, match="negative length"):
        ctx.render(M, sf, [10], [-1])
    ctx.close()


def test_synthetic_generator_statistics(native, mips):
    """The device generator restates TestDataLoader's distribution (loader.py:241-296)."""
    n = 4000000
    ctx = native.Context(64, 2)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True, with_rgb=True)
    d = ctx.download_particles(("x", "y", "z", "h", "mass", "q", "r", "g", "b"))
    pos = np.stack([d["x"], d["y"], d["z"]], axis=1).astype(np.float64)
    # mixture moments: weights (.5,.4,.1), means (0,0,0),(0,0,0),(6,10,0), sigmas (20,20,20),(4,.2,4),(2,2,3)
    np.testing.assert_allclose(pos.mean(axis=0), [0.6, 1.0, 0.0], atol=0.05)
    var = 0.5 * np.array([400.0, 400, 400]) + 0.4 * np.array([16.0, 0.04, 16]) + 0.1 * (np.array([4.0, 4, 9]) + np.array([36.0, 100, 0])) \
        - np.array([0.6, 1.0, 0.0]) ** 2
    np.testing.assert_allclose(pos.var(axis=0), var, rtol=0.01)
    # h-law: h = 2 / rho^0.333333 with the reference's (exponent without 1/2) density
    W, MU, SD = [0.5, 0.4, 0.1], np.array([[0, 0, 0], [0, 0, 0], [6, 10, 0.0]]), np.array([[20, 20, 20], [4, 0.2, 4], [2, 2, 3.0]])
    sel = slice(0, 50000)
    den = sum(w * np.exp(-np.sum((pos[sel] - mu) ** 2 / sd ** 2, axis=1)) / ((2 * np.pi) ** 1.5 * np.prod(sd)) for w, mu, sd in zip(W, MU, SD)) * n
    np.testing.assert_allclose(d["h"][sel], 2.0 / den ** 0.333333, rtol=2e-6)
    assert (d["mass"] == np.float32(1e-8)).all()
    np.testing.assert_allclose(d["q"][sel], np.sin(d["x"][sel]) * np.cos(d["y"][sel]) * np.cos(d["z"][sel]) * 1e-4, atol=2e-10)
    np.testing.assert_allclose(d["r"][sel], np.abs(np.sin(d["x"][sel] / 10.0)), atol=1e-6)
    # shards are reproducible: rows [a, b) of the full snapshot == a shard generated on its own
    c2 = native.Context(64, 2)
    c2.generate_synthetic(n, 1234567, 1000, 1337, 0.0)
    s = c2.download_particles(("x", "h"))
    assert np.array_equal(s["x"], d["x"][1234567:1235567]) and np.array_equal(s["h"], d["h"][1234567:1235567])
    c2.close()
    ctx.close()


@pytest.mark.parametrize("h_value,label", [(20.0, "all-huge"), (3.0, "all-mid")])
def test_record_list_overflow_replay(native, mips, h_value, label):
    """The deferred-footprint lists start small (N/4 mid, N/16 huge records); a frame that needs more
    reruns kernel S in records-only mode after growing them.  The image must not change because of it."""
    n, R = 400000, 1024
    rs = np.random.RandomState(4)
    pos = (rs.normal(size=(n, 3)) * 40.0).astype(np.float32)
    h = np.full(n, h_value, dtype=np.float32)              # P = 2 h R / scale = 204.8 px or 30.7 px
    m = rs.uniform(0.5, 1.5, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    ctx.render(M, sf)                                       # first frame: overflow -> grow -> replay
    st = ctx.stats()
    a = ctx.read_image().astype(np.float64)
    assert st["n_small"] == 0 and (st["n_huge"] if label == "all-huge" else st["n_mid"]) > 65536 * 4
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    ctx.render(M, sf)                                       # second frame: lists are large enough now
    b = ctx.read_image().astype(np.float64)
    ctx.render(M, sf, flags=native.PIPE_GENERIC)
    g = ctx.read_image().astype(np.float64)
    assert rel_close(a[..., 0], g[..., 0], 1e-5) and rel_close(b[..., 0], g[..., 0], 1e-5)
    tol = 1e-5 * np.abs(g[..., 1]).max()
    assert np.abs(a[..., 1] - g[..., 1]).max() <= tol * 50      # channel 1 cancels (signed q): scale by sum|terms| ~ 50x max
    ctx.close()


def test_config5_rgb_2048_full_size(native, mips):
    """BASELINE config 5 at full size: 5e7 star particles, rgb, 2048^2 (reference test_rgb_sph_output,
    tests/test_render_output.py:296-300, checks the mode at test size).  Pipeline (kernels S / M / gather) against the
    generic atomic kernel: value channels within 1e-5, the fragment-count channel and the fragment total exact."""
    n, R = 50_000_000, 2048
    M, sf = camera(200.0)
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=False, with_rgb=True)
    ctx.reorder_spatial(32, 1337)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=native.MODE_RGB)
    st = ctx.stats()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    assert st["n_huge"] > 0 and st["n_mid"] > 0 and st["n_small"] > 0
    fa = st["n_fragments"]
    a = ctx.read_image()
    ctx.render(M, sf, mode=native.MODE_RGB, flags=native.PIPE_GENERIC)
    assert ctx.stats()["n_fragments"] == fa
    g = ctx.read_image()
    assert np.array_equal(a[..., 3], g[..., 3]), "fragment-count channel must be exact"
    for c in range(3):
        assert rel_close(a[..., c].astype(np.float64), g[..., c].astype(np.float64), 1e-5), f"channel {c}"
    ctx.close()


def test_config2_weighted_1e7_full_size(native, mips):
    """BASELINE config 2 at full size: 1e7 particles, density-weighted quantity, 1024^2: pipeline vs the library's own GENERIC
    kernel at this size (the oracle meets the same configuration at full size in test_gpu_baseline_configs.py)."""
    n, R = 10_000_000, 1024
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True)
    ctx.reorder_spatial(32, 1337)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED)
    st = ctx.stats()
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n and st["n_huge"] > 0 and st["n_mega"] == 0
    fa = st["n_fragments"]
    a = ctx.read_image().astype(np.float64)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED, flags=native.PIPE_GENERIC)
    assert ctx.stats()["n_fragments"] == fa
    g = ctx.read_image().astype(np.float64)
    assert rel_close(a[..., 0], g[..., 0], 1e-5)
    # the weighted channel cancels (signed q): absolute tolerance scaled by the sum of |terms|, bounded by |q|max * density
    qmax = 1e-4
    assert (np.abs(a[..., 1] - g[..., 1]) <= 1e-5 * qmax * g[..., 0] + 1e-30).all()
    ctx.close()


def test_rgb_2048_matches_oracle_at_class_boundaries(native, mips):
    """rgb at R = 2048 against the CPU oracle: a few thousand particles whose footprints sit on and around the class
    boundaries 11.3 / 13.5 / 16 (small | mid since round 3) / 22.6 / 45.3 / 64 / 128 / 256 / 384 / 512 / 768 px (kernel S / M / gather
    kernels), plus a wide spread."""
    from oracle import oracle_c
    R, scale = 2048, 200.0
    M, sf = camera(scale)
    rs = np.random.RandomState(11)
    n = 3000
    pos = (rs.normal(size=(n, 3)) * np.array([60.0, 60.0, 30.0])).astype(np.float32)
    bounds = np.array([11.3137, 13.5, 16.0, 22.6274, 45.2548, 64.0, 128.0, 256.0, 384.0, 512.0, 700.0, 768.0])
    P = np.where(rs.uniform(size=n) < 0.6, rs.choice(bounds, size=n) * (1.0 + rs.choice([-1e-6, 0.0, 1e-6, 0.01, -0.01], size=n)),
                 np.exp(rs.uniform(np.log(0.3), np.log(1500.0), size=n)))
    h = (P * scale / (2.0 * R)).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None)
    ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=native.MODE_RGB)
    got = ctx.read_image()
    nf = ctx.stats()["n_fragments"]
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    want, nfrag = oracle_c.splat(x, y, z, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), mode=2, M=M, sf=float(sf), R=R, mips=mips)
    assert nf == nfrag, "coverage decisions differ from the oracle"
    assert np.array_equal(got[..., 3], want[..., 3]), "fragment-count channel must be exact"
    assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.parametrize("label,h_values", [("wide", (60.0,)), ("two-widths", (20.0, 60.0))])
def test_huge_list_overflow_replay_weighted_and_rgb(native, mips, label, h_values):
    """An overflow of the huge-record list replays kernel S (records only) after growing the list.  n > 16 * 65536 so the
    first frame must overflow, grow and replay; weighted (NW = 1) and rgb (NW = 2 weights per record)."""
    n, R = 1_100_000, 1024
    rs = np.random.RandomState(8)
    pos = (rs.normal(size=(n, 3)) * 60.0).astype(np.float32)
    h = rs.choice(np.asarray(h_values, dtype=np.float32), size=n)          # P = 2 h R / scale = 204.8 px or 614.4 px: all kernel H2's
    m = rs.uniform(0.5, 1.5, n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    M, sf = camera(200.0)
    for mode in (native.MODE_WEIGHTED, native.MODE_RGB):
        c2 = native.Context(R, 4)                # fresh context per mode: its record lists start small again
        c2.set_kernel_mips(mips)
        c2.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
        c2.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        c2.render(M, sf, mode=mode)              # first frame: overflow -> grow -> replay
        st = c2.stats()
        a = c2.read_image().astype(np.float64)
        assert st["n_huge"] > 16 * 65536 and st["n_small"] == 0 and st["n_mid"] == 0
        assert st["n_huge"] + st["n_culled"] == n
        c2.render(M, sf, mode=mode)              # second frame: the lists are large enough now
        b = c2.read_image().astype(np.float64)
        c2.render(M, sf, mode=mode, flags=native.PIPE_GENERIC)
        g = c2.read_image().astype(np.float64)
        nch = 3 if mode == native.MODE_RGB else 1
        for c in range(nch):
            assert rel_close(a[..., c], g[..., c], 1e-5) and rel_close(b[..., c], g[..., c], 1e-5), (label, mode, c)
        if mode == native.MODE_RGB:
            assert np.array_equal(a[..., 3], g[..., 3]) and np.array_equal(b[..., 3], g[..., 3])
        c2.close()


def test_in_block_arrangements_keep_cells_and_images(native, mips):
    """tsp_reorder_spatial rearranges every 512-particle block of the Morton order (option reorder_interleave: 2 = by
    descending smoothing length, the default: the 64 particles of a kernel-S wave step then have nearly one footprint width;
    1 = transposed 64 x 8; 0 = Morton order).  The permutation stays a bijection, every (stratum, cell) run keeps exactly its
    own particles (view culling by cell runs), and the render -- image and exact fragment count -- does not depend on it."""
    n, R = 300_000, 512
    M, sf = camera(200.0)
    out = {}
    for inter in (2, 1, 0):
        ctx = native.Context(R, 2)
        ctx.set_kernel_mips(mips)
        ctx.set_option("reorder_interleave", inter)
        ctx.generate_synthetic(n, 0, n, 1337, 0.0)
        perm = ctx.reorder_spatial(8, 1337, want_permutation=True)
        assert np.array_equal(np.sort(perm), np.arange(n))
        h = ctx.download_particles(("h",))["h"]
        ctx.set_option("count_fragments", 1)
        ctx.render(M, sf)
        out[inter] = (perm, ctx.cell_layout()["offsets"], ctx.strata_offsets(), ctx.read_image().astype(np.float64), ctx.stats()["n_fragments"], h)
        ctx.close()
    p0, c0, s0, img0, f0, h0 = out[0]
    sizes = np.diff(c0)
    big = int(np.argmax(sizes))
    b0 = -(-int(c0[big]) // 512) * 512          # an aligned block that lies inside the largest cell run
    assert b0 + 512 <= c0[big + 1]
    r = np.arange(512)
    for inter in (1, 2):
        p, c, s, img, f, h = out[inter]
        assert np.array_equal(c, c0) and np.array_equal(s, s0)
        assert not np.array_equal(p, p0), "the arrangement changed nothing"
        for a, b in zip(c[:-1][::37], c[1:][::37]):          # a sample of the cell runs: the same particles, in another order
            assert np.array_equal(np.sort(p[a:b]), np.sort(p0[a:b]))
        assert np.array_equal(np.sort(p[b0:b0 + 512]), np.sort(p0[b0:b0 + 512]))
        assert f == f0
        assert rel_close(img[..., 0], img0[..., 0], 1e-5)
    # transposition: slot (r mod 8) * 64 + r / 8 holds Morton rank r
    assert np.array_equal(out[1][0][b0 + (r % 8) * 64 + r // 8], p0[b0 + r])
    # by smoothing length: non-increasing inside the block, i.e. every wave step (64 slots) holds neighbours in h
    hb = out[2][5][b0:b0 + 512]
    assert (np.diff(hb) <= 0).all() and hb[0] > hb[-1]
    # ... and inside every (block x cell run) segment elsewhere
    c = out[2][1]
    for a, b in zip(c[:-1][::53], c[1:][::53]):
        for lo in range(int(a) - int(a) % 512, int(b), 512):
            seg = out[2][5][max(lo, int(a)):min(lo + 512, int(b))]
            assert (np.diff(seg) <= 0).all()


def test_vertex_weights_follow_every_upload(native, mips):
    """Kernel S streams m / h^2 (rgb / h^2) formed once per upload: a new mass, smoothing-length or rgb array must reach the
    next frame (the generic kernel divides per fragment and serves as the reference here)."""
    n, R = 50_000, 256
    rs = np.random.RandomState(3)
    pos = (rs.normal(size=(n, 3)) * 40.0).astype(np.float32)
    M, sf = camera(200.0)
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)

    def check(mode):
        ctx.render(M, sf, mode=mode)
        a = ctx.read_image().astype(np.float64)
        ctx.render(M, sf, mode=mode, flags=native.PIPE_GENERIC)
        g = ctx.read_image().astype(np.float64)
        for c in range(3 if mode == native.MODE_RGB else 1):
            assert g[..., c].sum() > 0 and rel_close(a[..., c], g[..., c], 1e-5)

    for trial in range(3):
        h = rs.uniform(0.2, 6.0, n).astype(np.float32)
        m = rs.uniform(0.5, 1.5, n).astype(np.float32) * (10.0 ** trial)
        ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
        check(native.MODE_WEIGHTED)
        for rgb_trial in range(2):
            rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32) * (3.0 ** rgb_trial)
            ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
            check(native.MODE_RGB)
        if trial == 1:
            ctx.reorder_spatial(4, 7)
            check(native.MODE_WEIGHTED)
            check(native.MODE_RGB)
    ctx.close()


def test_stream_batches_and_persistent_workgroups(native, mips):
    """Kernel S's persistent workgroups take batches of chunks from a shared counter: whatever the batch size and the number of
    workgroups per CU (options stream_batch_chunks, stream_blocks_per_cu; 1 workgroup per CU = long serial walks over many
    batches), every chunk is drawn exactly once -- particle and fragment counts identical, density images equal within the
    path's 1e-5."""
    n, R = 6_000_000, 512
    M, sf = camera(200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 4242, 0.0, with_quantity=True)
    ctx.reorder_spatial(8, 4242)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    ref = ctx.read_image().astype(np.float64)
    st0 = ctx.stats()
    assert st0["n_small"] > 0 and st0["n_mid"] > 0 and st0["n_huge"] > 0
    for batch, per_cu in ((4, 0), (8, 1), (64, 2), (8, 16), (4096, 0)):
        ctx.set_option("stream_batch_chunks", batch); ctx.set_option("stream_blocks_per_cu", per_cu)
        ctx.render(M, sf)
        st = ctx.stats()
        for k in ("n_huge", "n_culled", "n_fragments", "n_chunk_culled"):
            assert st[k] == st0[k], (k, batch, per_cu)
        # (which small footprints leave the LDS window for the mid list depends on the chunk sequence of a workgroup)
        assert st["n_small"] + st["n_mid"] == st0["n_small"] + st0["n_mid"], (batch, per_cu)
        assert rel_close(ctx.read_image()[..., 0].astype(np.float64), ref[..., 0], 1e-5), (batch, per_cu)
    ctx.close()


def test_mid_bins_with_more_strips_than_lds_counters(native, mips):
    """Kernel G's binning passes count per strip in LDS up to 8192 strips and with global atomics beyond: a 6000^2 image has
    94 x 188 = 17 672 strips of 64 x 32 pixels.  Against the oracle, exact fragment count included."""
    from oracle import oracle_c
    n, R, scale = 4000, 6000, 100.0
    M, sf = camera(scale)
    rs = np.random.RandomState(77)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, :2] = rs.uniform(-1.05, 1.05, size=(n, 2)) * scale
    P = np.exp(rs.uniform(np.log(8.0), np.log(63.9), n))
    h = (P * scale / (2.0 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.set_option("count_fragments", 1)
    ctx.set_option("p_small_milli", 0)
    ctx.render(M, sf)
    got = ctx.read_image()[..., 0]
    st = ctx.stats()
    assert st["n_mid"] > 3000 and st["n_small"] == 0
    want, nfrag = oracle_c.splat(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), h, m, None, None, mode=0, M=M, sf=sf, R=R, mips=mips)
    assert st["n_fragments"] == nfrag
    assert np.allclose(got, want[..., 0], rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.parametrize("mode_name", ["weighted", "rgb"])
def test_block_draws_in_record_slices(native, mips, mode_name):
    """A block of any size draws (sph.py:306-332): the deferred-record lists go through kernels G and H2 in slices (2^27 mid / 2^30
    huge records by default).  Option slice_records lowers the slice so that 1e6 particles need >= 3 slices of each list; image and
    fragment counts must equal the oracle's, and the unsliced render's, whatever the slicing."""
    from oracle import oracle_c
    n, R = 1_000_000, 512
    rs = np.random.RandomState(21)
    pos = (rs.normal(size=(n, 3)) * 60.0).astype(np.float32)
    # footprints of 2 h R / scale px: a third each below 16 px (kernel S), 16-64 px (kernel G), above 64 px (kernel H2)
    h = rs.choice(np.asarray([1.0, 2.5, 8.0, 11.0, 14.0, 30.0, 70.0], dtype=np.float32), size=n)
    m = rs.uniform(0.5, 1.5, n).astype(np.float32)
    q = rs.uniform(0.1, 1.0, n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    M, sf = camera(200.0)
    rgb_mode = mode_name == "rgb"
    mode = native.MODE_RGB if rgb_mode else native.MODE_WEIGHTED
    ctx = native.Context(R, 4 if rgb_mode else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    if rgb_mode:
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
    else:
        ctx.upload_quantity(q)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf, mode=mode)
    whole, st0 = ctx.read_image().astype(np.float64), ctx.stats()
    n_mid, n_huge = st0["n_mid"], st0["n_huge"]
    assert n_mid > 200_000 and n_huge > 100_000
    slice_records = 32768
    assert n_mid >= 3 * slice_records and n_huge >= 3 * slice_records
    ctx.set_option("slice_records", slice_records)
    ctx.render(M, sf, mode=mode)
    got, st = ctx.read_image().astype(np.float64), ctx.stats()
    assert (st["n_mid"], st["n_huge"], st["n_fragments"]) == (n_mid, n_huge, st0["n_fragments"])
    if rgb_mode:
        want, nfrag = oracle_c.splat(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(),
                                     mode=2, M=M, sf=float(sf), R=R, mips=mips)
        assert np.array_equal(got[..., 3], want[..., 3]) and np.array_equal(whole[..., 3], want[..., 3])
        nch = 3
    else:
        want, nfrag = oracle_c.splat(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), h, m, q, mode=0, M=M, sf=float(sf), R=R, mips=mips)
        nch = 2                      # (q > 0: no cancellation in channel 1)
    assert st["n_fragments"] == nfrag, "coverage decisions differ from the oracle"
    for c in range(nch):
        assert rel_close(got[..., c], want[..., c].astype(np.float64), 1e-5), (mode_name, c)
        assert rel_close(whole[..., c], want[..., c].astype(np.float64), 1e-5), (mode_name, c)
    ctx.close()


@pytest.mark.parametrize("stage", [1, 2])
def test_failed_block_leaves_the_accumulator_as_it_found_it(native, mips, stage):
    """tsp_render draws a block whole or not at all: a failure injected after kernel S (stage 1: its small footprints are already
    in the accumulator) or after kernel G (stage 2) returns the error and restores the float64 accumulator, the float32 image, the
    channel layout and the statistics of the previous call -- for a clearing block and for an accumulating one."""
    n, R = 300_000, 512
    rs = np.random.RandomState(33)
    pos = (rs.normal(size=(n, 3)) * 60.0).astype(np.float32)
    h = rs.choice(np.asarray([1.0, 2.5, 11.0, 30.0], dtype=np.float32), size=n)
    m = rs.uniform(0.5, 1.5, n).astype(np.float32)
    M, sf = camera(200.0)
    M2, sf2 = camera(120.0)
    half = n // 2
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    starts, lens = np.asarray([0], dtype=np.int64), np.asarray([half], dtype=np.int64)
    ctx.render(M, sf, starts, lens, clear=True)                  # frame A: the first half
    a, st_a = ctx.read_image().copy(), ctx.stats()
    for clear, cam in ((True, (M2, sf2)), (False, (M, sf))):
        ctx.set_option("debug_fail_stage", stage)
        with pytest.raises(native.BackendError) as err:
            ctx.render(cam[0], cam[1], np.asarray([half], dtype=np.int64), np.asarray([n - half], dtype=np.int64), clear=clear)
        assert "injected failure" in str(err.value)
        assert np.array_equal(ctx.read_image(), a), "the float32 image changed"
        assert ctx.stats() == st_a, "the statistics changed"
    # the accumulator itself is intact too: the second half added now gives the whole frame
    ctx.render(M, sf, np.asarray([half], dtype=np.int64), np.asarray([n - half], dtype=np.int64), clear=False)
    both = ctx.read_image().astype(np.float64)
    ctx.render(M, sf)
    whole = ctx.read_image().astype(np.float64)
    assert rel_close(both[..., 0], whole[..., 0], 1e-6)
    ctx.close()
