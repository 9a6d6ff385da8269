"""Kernel I (option `integrated_px`, off by default; csrc/tsp_integrated.hip): wide bilinear footprints through their sparse
second differences and two prefix sums of the image.

Its contract is NOT the per-pixel relative one of the default kernels (tests/test_gpu_parity.py): the scattered adds cancel to
the footprint's values only up to float64 rounding, and the texel coordinates are exact rather than the float32-rounded ones of the
oracle's canonical arithmetic, so

  * a pixel differs from the oracle by at most ~1e-6 of the PEAK contribution w * max(T) of the footprints that cover it
    (measured: 2-6e-7), in every channel; what is left below 1e-8 of the pass's largest peak contribution is snapped to 0, so
    pixels no footprint covers stay exactly 0 and a density image has no negative dust;
  * fragment counts are exact;
  * where every pixel lies under many wide footprints (the renders it is meant for) the image is within the north star's 1e-5
    relative per pixel of the exact kernels (measured 2.5e-7 on the 1.25e8-particle snapshot, 1e-6 here).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from topsy_amd import _native
    _native.load_library()
    return _native


def oracle_render(pos, h, a, b, c, mode, M, sf, R, mips):
    from oracle import oracle_c
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    return oracle_c.splat(x, y, z, h, a, b, c, mode=mode, M=M, sf=sf, R=R, mips=mips)


def wide_scene(R, scale, n, seed, pmin=130.0, pmax=6000.0):
    rs = np.random.RandomState(seed)
    P = np.exp(rs.uniform(np.log(pmin), np.log(pmax), n))
    P[:8] = [131.0, 140.0, 255.999, 256.0, 512.0, 1024.0, 2048.0, 4096.0][:min(8, n)]        # node spacings of 4, 8 ... pixels exactly (at R = 1024)
    h = (P * scale / (2.0 * R)).astype(np.float32)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, 0] = rs.uniform(-1.4, 1.4, n) * scale
    pos[:, 1] = rs.uniform(-1.4, 1.4, n) * scale
    pos[:, 2] = rs.uniform(-0.9, 0.9, n) * scale
    pos[::5, :2] = np.round(pos[::5, :2] / (2 * scale / R)) * (2 * scale / R)               # centres on pixel corners: ties
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    return pos, h, m, q, rgb


@pytest.mark.parametrize("mode", ["density", "weighted", "depth", "rgb"])
@pytest.mark.parametrize("R", [200, 1000, 1024])
def test_wide_footprints_against_the_oracle(native, mips, mode, R):
    """A few dozen footprints of 128 ... 6000 px, many of them partly off-screen on every side, each mode: against the oracle
    within 1e-6 of the summed peak contributions (the kernel's contract), exact fragment counts, and every one of them taken by
    kernel I."""
    from oracle import oracle_np
    scale = 100.0
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    n = 40
    pos, h, m, q, rgb = wide_scene(R, scale, n, seed=R)
    peak = float(mips[:4096].max())
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    ctx.set_option("integrated_px", 128)
    ctx.set_option("count_fragments", 1)
    w0 = m.astype(np.float64) / h.astype(np.float64) ** 2
    if mode == "rgb":
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        ctx.render(M, sf, mode=native.MODE_RGB)
        want, nfrag = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
        scales = [float((rgb[:, c].astype(np.float64) / h.astype(np.float64) ** 2).sum()) * peak for c in range(3)]
    elif mode == "depth":
        ctx.render(M, sf, mode=native.MODE_DEPTH)
        want, nfrag = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
        scales = [float(w0.sum()) * peak] * 2          # the depth values lie in [0, 1]
    else:
        if mode == "weighted":
            ctx.upload_quantity(q)
        ctx.render(M, sf, mode=native.MODE_WEIGHTED)
        want, nfrag = oracle_render(pos, h, m, q if mode == "weighted" else None, None, 0, M, sf, R, mips)
        scales = [float(w0.sum()) * peak, float((w0 * np.abs(q)).sum()) * peak]
    got = ctx.read_image()
    st = ctx.stats()
    assert st["n_fragments"] == nfrag
    assert st["n_mega"] == st["n_huge"] > n // 2               # all of the footprints >= 64 px (some are off-screen or outside the z-slab)
    for c, sc in enumerate(scales):
        if mode == "density" and c == 1:
            continue
        err = np.abs(got[..., c].astype(np.float64) - want[..., c])
        assert err.max() <= 1e-6 * sc, (c, err.max() / sc)
        if not (mode == "weighted" and c == 1):
            # what the cancelling adds leave where nothing was drawn is snapped to the exact zero it stands for: no negative dust
            # (the reference's autorange picks the linear scale on any negative value), zero where the oracle is zero
            assert got[..., c].min() >= 0.0
            assert (got[..., c][want[..., c] == 0] == 0).all()
    if mode == "rgb":
        assert np.array_equal(got[..., 3], want[..., 3])         # the fragment-count channel is exact as ever
    ctx.close()


@pytest.mark.parametrize("R", [300, 1024])
def test_kernel_image_that_does_not_vanish_at_its_edges(native, mips, R):
    """A level-0 image with values of 0.3 ... 0.9 on its edge rows and columns (and no symmetry): the jumps at the edges of the
    footprint square -- ~1e-6 of the peak for the SPH kernel, below this file's tolerances -- then carry as much as the
    interior.  Tolerance 5e-6 of the summed peaks: this image is rougher than the SPH kernel (the deviation is the texel-to-texel
    difference times the ~1e-5 texel by which float32 texel coordinates are off)."""
    from oracle import oracle_np
    j, i = np.mgrid[0:64, 0:64].astype(np.float64)
    custom = mips.copy()
    custom[:4096] = (0.6 + 0.3 * np.sin(0.15 * i + 0.3) * np.cos(0.11 * j - 0.2)).astype(np.float32).ravel()
    scale = 100.0
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    pos, h, m, q, rgb = wide_scene(R, scale, 30, seed=R + 1)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(custom)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    ctx.set_option("integrated_px", 128)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    got = ctx.read_image()
    want, nfrag = oracle_render(pos, h, m, q, None, 0, M, sf, R, custom)
    st = ctx.stats()
    assert st["n_fragments"] == nfrag and st["n_mega"] == st["n_huge"] > 10
    w0 = m.astype(np.float64) / h.astype(np.float64) ** 2
    for c, sc in enumerate([float(w0.sum()), float((w0 * np.abs(q)).sum())]):
        err = np.abs(got[..., c].astype(np.float64) - want[..., c])
        assert err.max() <= 5e-6 * sc * 0.9, (c, err.max() / sc)
    ctx.close()


def test_dense_scene_is_within_the_relative_tolerance(native, mips):
    """4e6 synthetic particles at the reference camera: every pixel lies under hundreds of footprints >= 256 px, and the image
    with kernel I is within 1e-5 relative PER PIXEL of the exact kernels' (the north star's tolerance; measured ~1e-6 here and
    2.5e-7 at 1.25e8 particles), over two accumulated render blocks."""
    R, scale, n = 1024, 200.0, 4000000
    M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 1337, 0.0)
    ctx.reorder_spatial(8, 1337)

    def two_blocks():
        ctx.render(M, 1.0 / scale, [0], [n // 2], clear=True)
        ctx.render(M, 1.0 / scale, [n // 2], [n - n // 2], clear=False)
        return ctx.read_image()[..., 0].astype(np.float64), ctx.stats()
    exact, st0 = two_blocks()
    ctx.set_option("integrated_px", 256)
    fast, st1 = two_blocks()
    assert st1["n_mega"] > 1000 and st1["n_mega"] > st0["n_mega"]
    lit = exact > 0.0
    assert lit.mean() > 0.999
    rel = np.abs(fast - exact)[lit] / exact[lit]
    assert rel.max() <= 1e-5, rel.max()
    assert (np.abs(fast[~lit]) <= 1e-12 * exact.max()).all()
    ctx.set_option("integrated_px", 0)           # and off again: the exact kernels, bit for bit up to the atomics' summation order
    again, _ = two_blocks()
    assert np.allclose(again, exact, rtol=1e-6, atol=0)
    ctx.close()


def test_reference_kats_with_kernel_I(native, mips, golden):
    """The reference's own golden vectors (tests/test_render_output.py:161-241, 360-446) at ITS tolerances with every footprint
    >= 128 px on kernel I: the scale-20 cameras of the weighted and bivariate vectors draw footprints of hundreds of pixels."""
    kats = golden["reference_kats.npz"]
    cams = golden["cameras.npz"]
    d = golden["testdata_n1000.npz"]
    ps, m, q = d["pos_smooth"], d["mass"], d["qty"]
    ctx = native.Context(200, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3], m)
    ctx.set_option("integrated_px", 128)

    def cam(name):
        return cams[name + ".transform"].T.copy(), float(cams[name + ".scale_factor"][0])
    M, sf = cam("identity_200")
    ctx.render(M, sf)
    test = ctx.read_image()[::20, ::20, 0].ravel()
    expect = kats["test_sph_output.expect"]
    np.testing.assert_allclose(test, expect, rtol=5e-1)           # reference :237
    assert abs((test / expect).mean() - 1.0) < 0.0015             # reference :240
    assert (test / expect).std() < 0.015                          # reference :241
    ctx.upload_quantity(q)
    M, sf = cam("rot0_0p4_20")
    ctx.render(M, sf)
    assert ctx.stats()["n_mega"] > 100
    im = ctx.read_image()
    np.testing.assert_allclose((im[..., 1] / im[..., 0])[::20, ::20].ravel(),
                               kats["test_sph_weighted_output.expect"], atol=1.5e-7)   # reference :198
    M, sf = cam("rot0_0p5_20")
    ctx.render(M, sf)
    assert ctx.stats()["n_mega"] > 100
    im = ctx.read_image()
    np.testing.assert_allclose(im[::20, ::20, 0].ravel(), kats["test_bivariate_render.expect_den"], rtol=2e-3)
    np.testing.assert_allclose((im[..., 1] / im[..., 0])[::20, ::20].ravel(),
                               kats["test_bivariate_render.expect_qty"], atol=1e-4)
    ctx.close()


def test_through_the_visualizer(monkeypatch):
    """config.INTEGRATED_FOOTPRINT_PX reaches the context of a Visualizer; the 1000-particle reference scene zoomed to
    scale 20 (footprints of hundreds of pixels at 200^2) against the default kernels."""
    import topsy_amd
    from topsy_amd import config
    from topsy_amd.drawreason import DrawReason

    def image():
        vis = topsy_amd.test(1000, render_resolution=200)
        vis.scale = 20.0
        vis.rotate(0.0, 0.4)
        vis.render_sph(DrawReason.EXPORT)
        im = np.array(vis.get_sph_image(), dtype=np.float64)
        n_mega = vis._sph._context.stats()["n_mega"]
        vis.close()
        return im, n_mega
    ref, _ = image()
    monkeypatch.setattr(config, "INTEGRATED_FOOTPRINT_PX", 128)
    got, n_mega = image()
    assert n_mega > 100
    assert np.abs(got - ref).max() <= 1e-5 * ref.max(), np.abs(got - ref).max() / ref.max()


def test_on_several_contexts(native, mips):
    """The option reaches every shard of the in-process multi-GPU driver (here two contexts on one device): the summed image
    of a dense scene is within 1e-5 per pixel of one context's exact render."""
    from topsy_amd import multigpu
    R, scale, n = 512, 200.0, 3000000
    M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
    one = native.Context(R, 2)
    one.set_kernel_mips(mips)
    one.generate_synthetic(n, 0, n, 1337, 0.0)
    one.render(M, 1.0 / scale)
    exact = one.read_image()[..., 0].astype(np.float64)
    one.close()
    two = multigpu.MultiGpuContext(R, 2, [0, 0])
    two.set_kernel_mips(mips)
    two.generate_synthetic(n, 0, n, 1337, 0.0)
    two.set_option("integrated_px", 128)
    two.render(M, 1.0 / scale)
    two.end_frame()
    got = two.read_image()[..., 0].astype(np.float64)
    assert two.stats()["n_mega"] > 1000
    two.close()
    lit = exact > 0
    assert lit.mean() > 0.99
    rel = np.abs(got - exact)[lit] / exact[lit]
    assert rel.max() <= 1e-5, rel.max()


def test_option_range(native, mips):
    ctx = native.Context(128, 2)
    ctx.set_kernel_mips(mips)
    with pytest.raises(Exception):
        ctx.set_option("integrated_px", 64)       # below 128 px the rim of the square is narrower than a pixel
    ctx.set_option("integrated_px", 128)
    ctx.set_option("integrated_px", 0)
    ctx.close()


def test_snap_is_decided_once_per_pixel(native, mips):
    """Weighted render: wherever kernel I writes the density channel as exact 0 it writes density x quantity as exact 0 too (and
    nowhere else), even when max |w| and max |w q| of the pass come from different footprints -- the colormap's g / r must
    never read +-inf at a faint rim.  Switching the option off afterwards releases the second-difference images and the exact
    kernels take over again."""
    from oracle import oracle_np
    R, scale = 512, 100.0
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    # a heavy footprint with a tiny quantity, a light one (1e-7 of the heavy weight) with a huge quantity, far apart
    P = np.array([300.0, 300.0])
    h = (P * scale / (2.0 * R)).astype(np.float32)
    pos = np.array([[-55.0, -55.0, 0.0], [55.0, 55.0, 0.0]], dtype=np.float32)
    m = np.array([1.0, 1e-7], dtype=np.float32)
    q = np.array([1e-6, 1e6], dtype=np.float32)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED)
    exact = ctx.read_image().astype(np.float64)
    ctx.set_option("integrated_px", 128)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED)
    assert ctx.stats()["n_mega"] == 2
    got = ctx.read_image().astype(np.float64)
    assert ((got[..., 0] == 0) == (got[..., 1] == 0)).all(), "a pixel keeps or loses BOTH channels"
    lit = got[..., 0] > 0
    assert lit.sum() > 20000 and (exact[..., 0][lit] > 0).all()
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = got[..., 1][lit] / got[..., 0][lit]
    assert np.isfinite(ratio).all()
    ctx.set_option("integrated_px", 0)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED)
    assert np.array_equal(ctx.read_image().astype(np.float64), exact) or np.allclose(ctx.read_image(), exact, rtol=1e-6, atol=0)
    ctx.close()
