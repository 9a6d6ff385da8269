"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.md section 5 / north_star): density channel <= 1e-5 relative; weighted channel
absolute tolerance scaled by the per-pixel sum of |terms| (it can cancel); colormap uint8 bit-exact
on the identical float buffer.
"""
import numpy as np
import pytest

from conftest import make_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from topsy_amd import _native
    _native.load_library()
    return _native


def oracle_render(pos, h, a, b, c, mode, M, sf, R, mips, ranges=None):
    from oracle import oracle_c
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    return oracle_c.splat(x, y, z, h, a, b, c, mode=mode, M=M, sf=sf, R=R, mips=mips, ranges=ranges)


def abs_terms_image(pos, h, m, q, M, sf, R, mips):
    """sum of |val * q| per pixel: the scale of the weighted channel's rounding noise."""
    img, _ = oracle_render(pos, h, m, np.abs(q), None, 0, M, sf, R, mips)
    return img[..., 1]


def check_2ch(got, want, abs_terms, rtol=1e-5):
    d0 = np.abs(got[..., 0] - want[..., 0])
    assert (d0 <= rtol * np.abs(want[..., 0]) + 1e-30).all(), \
        f"density channel: max rel err {np.max(d0 / np.maximum(np.abs(want[..., 0]).astype(np.float64), 1e-300))}"
    d1 = np.abs(got[..., 1] - want[..., 1])
    assert (d1 <= rtol * abs_terms + 1e-30).all(), "weighted channel beyond atol scaled by sum|terms|"


CAMERAS = [
    ("identity", np.eye(3), np.zeros(3), 200.0),
    ("zoom_rot", None, np.array([1.5, -2.0, 0.25]), 35.0),
    ("wide", None, np.zeros(3), 900.0),
]


def _rot(a, b):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    rx = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
    ry = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
    return rx @ ry


@pytest.mark.parametrize("pipe", ["generic", "default"])
@pytest.mark.parametrize("cam", CAMERAS, ids=[c[0] for c in CAMERAS])
@pytest.mark.parametrize("R", [200, 1024])
def test_weighted_matches_oracle(native, mips, cam, R, pipe):
    from oracle import oracle_np
    name, rot, off, scale = cam
    rot = _rot(0.3, -0.7) if rot is None else rot
    M, sf = oracle_np.transform_matrix(rot, off, scale)
    pos, h, m, q, _ = make_cloud(20000, seed=3)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    flags = native.PIPE_GENERIC if pipe == "generic" else native.PIPE_DEFAULT
    ctx.render(M, sf, mode=native.MODE_WEIGHTED, flags=flags)
    got = ctx.read_image()
    want, _ = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
    check_2ch(got, want, abs_terms_image(pos, h, m, q, M, sf, R, mips), rtol=1e-5)
    # density-only (q = NULL): channel 1 must be exactly zero
    ctx.upload_quantity(None)
    ctx.render(M, sf, mode=native.MODE_WEIGHTED, flags=flags)
    got = ctx.read_image()
    assert (got[..., 1] == 0).all()
    assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.parametrize("pipe", ["generic", "default"])
def test_rgb_and_depth_match_oracle(native, mips, pipe):
    from oracle import oracle_np
    R = 256
    M, sf = oracle_np.transform_matrix(_rot(0.1, 0.2), np.zeros(3), 120.0)
    pos, h, m, q, rgb = make_cloud(8000, seed=5)
    flags = native.PIPE_GENERIC if pipe == "generic" else native.PIPE_DEFAULT
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None)
    ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
    ctx.render(M, sf, mode=native.MODE_RGB, flags=flags)
    got = ctx.read_image()
    want, _ = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
    assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
    assert np.array_equal(got[..., 3], want[..., 3]), "fragment-count channel must be exact"
    ctx.close()
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.render(M, sf, mode=native.MODE_DEPTH, flags=flags)
    got = ctx.read_image()
    want, _ = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
    assert np.allclose(got, want, rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.parametrize("pipe", ["generic", "default"])
def test_ranges_and_accumulate(native, mips, pipe):
    """(start, len) blocks as the reference's indirect draws: clear on the first block only."""
    from oracle import oracle_np
    R = 128
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 150.0)
    pos, h, m, q, _ = make_cloud(5000, seed=9)
    flags = native.PIPE_GENERIC if pipe == "generic" else native.PIPE_DEFAULT
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    starts = np.array([10, 700, 2500, 4990, 6000, -5]); lens = np.array([300, 1, 1700, 50, 10, 3])
    ctx.render(M, sf, starts, lens, clear=True, flags=flags)
    got = ctx.read_image()
    cs = np.array([10, 700, 2500, 4990]); cl = np.array([300, 1, 1700, 10])
    want, _ = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips, ranges=(cs, cl))
    at = abs_terms_image(pos, h, m, q, M, sf, R, mips)
    check_2ch(got, want, at)
    # second block without clearing adds on top
    ctx.render(M, sf, np.array([0]), np.array([10]), clear=False, flags=flags)
    got2 = ctx.read_image()
    want2, _ = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips,
                             ranges=(np.array([0, 10, 700, 2500, 4990]), np.array([10, 300, 1, 1700, 10])))
    check_2ch(got2, want2, at, rtol=1e-5)
    # empty selection: clear only
    ctx.render(M, sf, np.array([], dtype=np.int64), np.array([], dtype=np.int64), clear=True, flags=flags)
    assert (ctx.read_image() == 0).all()
    ctx.close()


def test_edge_cases(native, mips):
    from oracle import oracle_np
    R = 64
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 10.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    # zero particles
    e = np.zeros(0, dtype=np.float32)
    ctx.upload_particles(e, e, e, e, e)
    ctx.render(M, sf)
    assert (ctx.read_image() == 0).all()
    # NaN / inf / zero / negative smoothing, out-of-slab z, off-screen: nothing drawn, nothing crashes
    x = np.array([0, 0, 0, 0, 0, 1e6, np.nan, 0, 0.3], dtype=np.float32)
    y = np.zeros_like(x)
    z = np.array([0, 0, 0, 0, 50, 0, 0, np.inf, 0], dtype=np.float32)
    h = np.array([np.nan, np.inf, 0, -1, 1, 1, 1, 1, 0.7], dtype=np.float32)
    m = np.ones_like(x)
    ctx.upload_particles(x, y, z, h, m)
    ctx.render(M, sf)
    got = ctx.read_image()
    pos = np.stack([x, y, z], axis=1)
    want, _ = oracle_render(pos, h, m, None, None, 0, M, sf, R, mips)
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=1e-5, atol=0)
    assert got[..., 0].sum() > 0      # the last particle is the only one drawn
    ctx.close()


def test_errors(native, mips):
    ctx = native.Context(32, 2)
    x = np.zeros(4, dtype=np.float32)
    ctx.upload_particles(x, x, x, x + 1, x + 1)
    with pytest.raises(native.BackendError, match="tsp_set_kernel_mips"):
        ctx.render(np.eye(4), 1.0)
    ctx.set_kernel_mips(mips)
    with pytest.raises(native.BackendError):
        ctx.render(np.eye(4), 1.0, mode=native.MODE_RGB)      # 2-channel context
    with pytest.raises(native.BackendError):
        native.Context(32, 3)
    with pytest.raises(ValueError):
        ctx.upload_quantity(np.zeros(3, dtype=np.float32))
    ctx.close()


def test_colormap_bit_exact(native, mips, golden):
    """kernel B / B' vs the oracle on the identical float buffer: uint8 must match bit-for-bit."""
    from oracle import oracle_c, oracle_np
    R = 200
    d = golden["testdata_n1000.npz"]
    luts = golden["colormap_luts.npz"]
    ps, m, q = d["pos_smooth"], d["mass"], d["qty"]
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 200.0)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3], m)
    ctx.upload_quantity(q)
    ctx.render(M, sf)
    img = ctx.read_image()
    cases = [(True, False, -12.8, -8.2), (False, False, 0.0, 2e-9), (True, True, -6.0, -4.0), (False, True, -1e-4, 1e-4)]
    for name in ("twilight_shifted", "viridis"):
        for log, weighted, vmin, vmax in cases:
            got = ctx.colormap_scalar(luts[name], vmin, vmax, log, weighted)
            want = oracle_c.colormap_scalar(img, luts[name], vmin, vmax, log, weighted)
            assert np.array_equal(got, want), (name, log, weighted)
            got_h = ctx.colormap_scalar_host(img, luts[name], vmin, vmax, log, weighted)
            assert np.array_equal(got_h, want)
    # special values: zeros (log -> -inf -> t=0), 0/0 (NaN -> t=0), negatives, inf, denormals
    rs = np.random.RandomState(0)
    sp = np.zeros((R, R, 2), dtype=np.float32)
    sp[..., 0] = np.exp(rs.uniform(-90, 80, size=(R, R)))
    sp[..., 1] = rs.normal(size=(R, R)) * sp[..., 0]
    sp[0, :50] = 0.0
    sp[1, :50, 0] = -1.0
    sp[2, :50, 0] = np.inf
    sp[3, :50, 0] = 1e-42
    sp[4, :50] = np.nan
    ctx.write_image(sp)
    for log, weighted, vmin, vmax in [(True, False, -30, 30), (False, False, 0, 1), (True, True, -3, 1), (False, True, -2, 2)]:
        got = ctx.colormap_scalar(luts["viridis"], vmin, vmax, log, weighted)
        want = oracle_c.colormap_scalar(sp, luts["viridis"], vmin, vmax, log, weighted)
        assert np.array_equal(got, want), (log, weighted)
    ctx.close()
    # rgb map
    ctx = native.Context(R, 4)
    rgb4 = np.zeros((R, R, 4), dtype=np.float32)
    rgb4[..., :3] = np.exp(rs.uniform(-20, 5, size=(R, R, 3)))
    rgb4[0, :20, :3] = 0
    rgb4[1, :20, 0] = np.nan
    ctx.write_image(rgb4)
    for gamma in (1.0, 0.5, 2.2):
        got = ctx.colormap_rgb(-6.0, -1.0, gamma)
        want = oracle_c.colormap_rgb(rgb4, -6.0, -1.0, gamma)
        assert np.array_equal(got, want), gamma
        gf = ctx.colormap_rgb(-6.0, -1.0, gamma, as_float=True)
        wf = oracle_c.colormap_rgb(rgb4, -6.0, -1.0, gamma, as_float=True)
        assert np.array_equal(gf, wf, equal_nan=True), gamma
        assert np.array_equal(ctx.colormap_rgb_host(rgb4, -6.0, -1.0, gamma), want)
    ctx.close()


def test_reference_kats_through_hip(native, mips, golden):
    """The reference's own golden vectors (tests/test_render_output.py) at its own tolerances, HIP path."""
    from oracle import oracle_np
    kats = golden["reference_kats.npz"]
    cams = golden["cameras.npz"]
    d = golden["testdata_n1000.npz"]
    ps, m, q = d["pos_smooth"], d["mass"], d["qty"]
    R = 200
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3], m)

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
    im = ctx.read_image()
    np.testing.assert_allclose((im[..., 1] / im[..., 0])[::20, ::20].ravel(),
                               kats["test_sph_weighted_output.expect"], atol=1.5e-7)   # reference :198
    M, sf = cam("rot0_0p5_20")
    ctx.render(M, sf)
    im = ctx.read_image()
    np.testing.assert_allclose(im[::20, ::20, 0].ravel(), kats["test_bivariate_render.expect_den"], rtol=2e-3)
    np.testing.assert_allclose((im[..., 1] / im[..., 0])[::20, ::20].ravel(),
                               kats["test_bivariate_render.expect_qty"], atol=1e-4)
    ctx.close()


@pytest.mark.parametrize("R", [1, 2, 8, 33, 65])
def test_tiny_and_odd_resolutions(native, mips, R):
    """Resolutions below / not a multiple of every tile size (64-px window, 64x32 and 128x64 tiles)."""
    from oracle import oracle_np
    M, sf = oracle_np.transform_matrix(_rot(0.2, 0.1), np.zeros(3), 90.0)
    pos, h, m, q, _ = make_cloud(3000, seed=21)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.upload_quantity(q)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    got = ctx.read_image()
    want, nfrag = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
    assert ctx.stats()["n_fragments"] == nfrag
    check_2ch(got, want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
    ctx.close()


def test_single_particles_and_ties(native, mips, golden):
    """n = 1 (the reference loader's special case), a footprint thousands of pixels wide, footprints whose
    edges fall exactly on pixel centres, and particles exactly on the z-slab faces."""
    from oracle import oracle_np
    R = 128
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 64.0)       # 1 px = 1 length unit
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    d = golden["testdata_n1.npz"]
    cases = [
        (d["pos_smooth"][:, :3], d["pos_smooth"][:, 3], d["mass"]),                                    # TestDataLoader(1)
        (np.array([[3.0, -7.0, 0.0]], np.float32), np.array([4000.0], np.float32), np.ones(1, np.float32)),    # P = 8000 px
        # half-widths that put footprint edges exactly on pixel centres (|d| < half is strict)
        (np.array([[0.5, 0.5, 0.0], [0.0, 0.0, 0.0], [10.5, -3.5, 0.0]], np.float32), np.array([1.0, 0.75, 2.25], np.float32),
         np.ones(3, np.float32)),
        # z exactly on the slab faces (clip z = 0 and 1 are kept), just outside (dropped)
        (np.array([[1.0, 1.0, 64.0], [5.0, 5.0, -64.0], [9.0, 9.0, 64.00001], [20.0, 20.0, -64.00001]], np.float32),
         np.full(4, 3.0, np.float32), np.ones(4, np.float32)),
    ]
    for pos, h, m in cases:
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
        for flags in (native.PIPE_DEFAULT, native.PIPE_GENERIC):
            ctx.set_option("count_fragments", 1)
            ctx.render(M, sf, flags=flags)
            got = ctx.read_image()
            want, nfrag = oracle_render(pos, h.astype(np.float32), m.astype(np.float32), None, None, 0, M, sf, R, mips)
            assert ctx.stats()["n_fragments"] == nfrag
            assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
            ctx.set_option("count_fragments", 0)
            ctx.render(M, sf, flags=flags)                  # with the exact corner culling active
            assert np.allclose(ctx.read_image()[..., 0], want[..., 0], rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.parametrize("mode", ["weighted", "rgb", "depth"])
def test_scattered_small_footprints_leave_the_window(native, mips, mode):
    """Unordered particles with footprints of 0-11 px spread over a 1024^2 image: a 512-particle chunk spans far
    more than kernel S's 64-px LDS window, so most of them take the MID-list route (kernel G, mip 3); the counter
    channel of those rgb footprints comes from the rectangle sum.  Everything must still match the oracle."""
    from oracle import oracle_np
    R = 1024
    M, sf = oracle_np.transform_matrix(_rot(0.4, 0.15), np.zeros(3), 100.0)
    rs = np.random.RandomState(77)
    n = 60000
    pos = rs.uniform(-95.0, 95.0, size=(n, 3)).astype(np.float32)
    h = np.exp(rs.uniform(np.log(0.01), np.log(1.05), size=n)).astype(np.float32)      # P = 2 h R / scale <= 10.8 px
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    if mode == "rgb":
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        ctx.render(M, sf, mode=native.MODE_RGB)
        want, _ = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
        got = ctx.read_image()
        assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
        assert np.array_equal(got[..., 3], want[..., 3])
    elif mode == "depth":
        ctx.render(M, sf, mode=native.MODE_DEPTH)
        want, _ = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
        assert np.allclose(ctx.read_image(), want, rtol=1e-5, atol=0)
    else:
        ctx.upload_quantity(q)
        ctx.render(M, sf, mode=native.MODE_WEIGHTED)
        want, _ = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
        check_2ch(ctx.read_image(), want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
    st = ctx.stats()
    assert st["n_huge"] == 0 and st["n_mid"] > n // 4 and st["n_small"] > 0          # all are "small" by width
    assert st["n_small"] + st["n_mid"] + st["n_culled"] == n
    ctx.close()


@pytest.mark.parametrize("rule", [1, 2], ids=["bilinear_mip0", "bilinear_mip"])
def test_alternative_sampling_rules(native, mips, rule):
    """TSP_SAMPLE_BILINEAR_MIP0 / _MIP (diagnostic, generic kernel): same restatement as the oracle's `sampling`
    argument, and really different from the reference rule."""
    from oracle import oracle_np, oracle_c
    R = 256
    M, sf = oracle_np.transform_matrix(_rot(0.25, -0.1), np.zeros(3), 120.0)
    pos, h, m, q, _ = make_cloud(6000, seed=13)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(x, y, z, h, m)
    ctx.upload_quantity(q)
    flag = native.SAMPLE_BILINEAR_MIP0 if rule == 1 else native.SAMPLE_BILINEAR_MIP
    ctx.render(M, sf, flags=flag)                     # implies the generic kernel
    got = ctx.read_image()
    want, _ = oracle_c.splat(x, y, z, h, m, q, mode=0, M=M, sf=sf, R=R, mips=mips, sampling=rule)
    terms, _ = oracle_c.splat(x, y, z, h, m, np.abs(q), mode=0, M=M, sf=sf, R=R, mips=mips, sampling=rule)
    check_2ch(got, want, terms[..., 1])
    ref, _ = oracle_c.splat(x, y, z, h, m, q, mode=0, M=M, sf=sf, R=R, mips=mips)
    assert np.abs(want[..., 0] / np.maximum(ref[..., 0], 1e-30) - 1.0).max() > 1e-3
    ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_randomised_views(native, mips, seed):
    """Random resolution, camera, smoothing-length range and mode per seed; the three-class pipeline must agree
    with the oracle on the image (1e-5) and on the exact fragment count, including particles with degenerate
    attributes (zero / negative / non-finite h, non-finite positions) that the reference's rasteriser drops."""
    from oracle import oracle_np
    rs = np.random.RandomState(1000 + seed)
    R = int(rs.choice([17, 64, 100, 129, 255, 300, 512]))
    scale = float(np.exp(rs.uniform(np.log(5.0), np.log(400.0))))
    M, sf = oracle_np.transform_matrix(_rot(rs.uniform(-3, 3), rs.uniform(-3, 3)), rs.normal(size=3) * 5.0, scale)
    n = 4000
    pos = (rs.normal(size=(n, 3)) * rs.uniform(5.0, 60.0, size=3)).astype(np.float32)
    hmax = scale * rs.choice([0.05, 0.5, 3.0])           # footprints up to 0.1 / 1 / 6 image widths
    h = np.exp(rs.uniform(np.log(hmax * 1e-4), np.log(hmax), size=n)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    # degenerate particles
    h[:6] = [0.0, -1.0, np.nan, np.inf, 1e-30, 1e30]
    pos[6, 0] = np.nan; pos[7, 1] = np.inf; pos[8, 2] = -np.inf
    mode = ["weighted", "rgb", "depth"][seed % 3]
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    ctx.set_option("count_fragments", 1)
    if mode == "rgb":
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        ctx.render(M, sf, mode=native.MODE_RGB)
        want, nfrag = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
        got = ctx.read_image()
        assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
        assert np.array_equal(got[..., 3], want[..., 3])
    elif mode == "depth":
        ctx.render(M, sf, mode=native.MODE_DEPTH)
        want, nfrag = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
        assert np.allclose(ctx.read_image(), want, rtol=1e-5, atol=0)
    else:
        ctx.upload_quantity(q)
        ctx.render(M, sf, mode=native.MODE_WEIGHTED)
        want, nfrag = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
        check_2ch(ctx.read_image(), want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
    assert ctx.stats()["n_fragments"] == nfrag
    # and once more with the exact disc culling active (no fragment statistics)
    ctx.set_option("count_fragments", 0)
    md = {"weighted": native.MODE_WEIGHTED, "rgb": native.MODE_RGB, "depth": native.MODE_DEPTH}[mode]
    ctx.render(M, sf, mode=md)
    got = ctx.read_image()
    if mode == "weighted":
        check_2ch(got, want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
    else:
        assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
        if mode == "rgb":
            assert np.array_equal(got[..., 3], want[..., 3])
    ctx.close()


@pytest.mark.parametrize("mode", ["weighted", "depth", "rgb"])
@pytest.mark.parametrize("R", [200, 1024])
def test_gather_kernel_class_boundaries(native, mips, mode, R):
    """Footprints right at the class boundary of the tile-gather kernel -- 64 px (nearest mip 0 -> bilinear: kernel G -> H2; a
    texel row per pixel row, the one case where rounding may skip a texel row) -- and at the widths where rounds 1-4 switched
    kernels (128 / 256 / 384 / 512 / 768 px), at arbitrary sub-pixel centres, partly off-screen, against the oracle: image
    within 1e-5 and the exact fragment count.  R = 200 leaves partial tiles and strips on both axes."""
    from oracle import oracle_np
    scale = 100.0
    M, sf = oracle_np.transform_matrix(_rot(0.0, 0.0), np.zeros(3), scale)
    widths = np.array([63.99, 64.0, 64.0001, 64.001, 64.5, 65.0, 90.0, 127.9, 127.999, 128.0, 128.001, 200.3, 255.9, 255.999, 256.0,
                       256.001, 300.0, 383.999, 384.0, 384.001, 511.9, 511.999, 512.0, 512.001, 700.0, 767.999, 768.0, 768.001, 1023.0, 1024.0, 3000.0,
                       20000.0], dtype=np.float64)
    rs = np.random.RandomState(77)
    reps = 6
    P = np.repeat(widths, reps)
    n = len(P)
    h = (P * scale / (2.0 * R)).astype(np.float32)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, 0] = rs.uniform(-1.3, 1.3, n) * scale
    pos[:, 1] = rs.uniform(-1.3, 1.3, n) * scale
    pos[:, 2] = rs.uniform(-0.9, 0.9, n) * scale
    pos[::7, :2] = np.round(pos[::7, :2] / (2 * scale / R)) * (2 * scale / R)      # centres on pixel corners: ties
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    for count in (1, 0):                       # with fragment statistics (no disc culling), then with the exact culling
        ctx.set_option("count_fragments", count)
        if mode == "rgb":
            ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
            ctx.render(M, sf, mode=native.MODE_RGB)
            want, nfrag = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
            got = ctx.read_image()
            assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0)
            assert np.array_equal(got[..., 3], want[..., 3])
        elif mode == "depth":
            ctx.render(M, sf, mode=native.MODE_DEPTH)
            want, nfrag = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
            assert np.allclose(ctx.read_image(), want, rtol=1e-5, atol=0)
        else:
            ctx.upload_quantity(q)
            ctx.render(M, sf, mode=native.MODE_WEIGHTED)
            want, nfrag = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
            check_2ch(ctx.read_image(), want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
        st = ctx.stats()
        if count:
            assert st["n_fragments"] == nfrag
        wide = int((h.astype(np.float64) * 2.0 * R / scale >= 64.0).sum())
        assert wide // 3 < st["n_huge"] <= wide     # (some of them are off-screen or outside the z-slab)
    # kernel H2's other strip shape / occupancy builds (what other record counts select, and the A/B builds), exact culling on
    for variant in {"rgb": (4,), "weighted": (4,), "depth": (4,)}[mode]:
        ctx.set_option("huge_variant", variant)
        if mode == "rgb":
            ctx.render(M, sf, mode=native.MODE_RGB)
            got = ctx.read_image()
            assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0), variant
            assert np.array_equal(got[..., 3], want[..., 3]), variant
        elif mode == "depth":
            ctx.render(M, sf, mode=native.MODE_DEPTH)
            assert np.allclose(ctx.read_image(), want, rtol=1e-5, atol=0), variant
        else:
            ctx.render(M, sf, mode=native.MODE_WEIGHTED)
            check_2ch(ctx.read_image(), want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
        ctx.set_option("huge_variant", 1)
    ctx.close()


def test_gather_kernels_fold_their_accumulators(native, mips):
    """More than 2048 footprints per wave strip: kernel H2 flushes its float32 accumulators to the float64 target every 2048
    footprints (forced here by one workgroup per tile); density stays within 1e-5 of the oracle for every strip shape /
    occupancy build, the exact fragment count included, whatever the number of workgroups per tile."""
    from oracle import oracle_np
    R, scale, n = 160, 100.0, 5200
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    rs = np.random.RandomState(5)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, :2] = rs.uniform(-0.8, 0.8, size=(n, 2)) * scale
    P = np.where(np.arange(n) % 2 == 0, rs.uniform(70.0, 400.0, n), rs.uniform(520.0, 3000.0, n))
    h = (P * scale / (2.0 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.set_option("huge_split", 1)
    ctx.render(M, sf)
    got = ctx.read_image()
    assert ctx.stats()["n_mega"] == 0 and ctx.stats()["n_huge"] == n
    want, nfrag = oracle_render(pos, h, m, None, None, 0, M, sf, R, mips)
    assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
    # H2's strip shapes / occupancies: 64x32 strips at 6 / 8 / 7 waves per SIMD (7 = what large record counts select,
    # row factors fetched group by group), 64x16 at 7 / 8
    for variant in (2, 7, 4, 5, 6):
        ctx.set_option("huge_variant", variant)
        ctx.render(M, sf)
        assert np.allclose(ctx.read_image()[..., 0], want[..., 0], rtol=1e-5, atol=0), variant
    ctx.set_option("huge_variant", 1)
    ctx.set_option("count_fragments", 1)
    for split in (8, 24):                      # several workgroups per tile
        ctx.set_option("huge_split", split)
        ctx.render(M, sf)
        assert np.allclose(ctx.read_image()[..., 0], want[..., 0], rtol=1e-5, atol=0), split
        assert ctx.stats()["n_fragments"] == nfrag, split
    ctx.close()


@pytest.mark.parametrize("mode", ["weighted", "rgb"])
def test_huge_records_binned_by_band_or_not(native, mips, mode):
    """Kernel H2 scans the huge records of its 64-row image band (bins filled by one pass) or, when the bins would not fit the
    memory budget (option huge_band_mib; 0 = never bin), the whole list: same image, same exact fragment count, against the
    oracle; R = 300 leaves a partial last band, footprints reach from one band to all of them."""
    from oracle import oracle_np
    R, scale, n = 300, 100.0, 6000
    M, sf = oracle_np.transform_matrix(_rot(0.1, 0.0), np.zeros(3), scale)
    rs = np.random.RandomState(17)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, :2] = rs.uniform(-1.2, 1.2, size=(n, 2)) * scale
    pos[:, 2] = rs.uniform(-0.5, 0.5, n) * scale
    P = np.exp(rs.uniform(np.log(64.0), np.log(1500.0), n))
    P[:40] = [64.0, 64.0001, 127.99, 128.0] * 10          # class boundary and band-edge widths
    h = (P * scale / (2.0 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    if mode == "rgb":
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        want, nfrag = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
        md = native.MODE_RGB
    else:
        ctx.upload_quantity(q)
        want, nfrag = oracle_render(pos, h, m, q, None, 0, M, sf, R, mips)
        md = native.MODE_WEIGHTED
    for mib in (6144, 0, 1):                   # binned, never binned, a budget the bins do not fit (6000 records x 5 bands x 24 B < 1 MiB: binned after all)
        for count in (1, 0):
            ctx.set_option("huge_band_mib", mib); ctx.set_option("count_fragments", count)
            ctx.render(M, sf, mode=md)
            got = ctx.read_image()
            if mode == "rgb":
                assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0), mib
                assert np.array_equal(got[..., 3], want[..., 3]), mib
            else:
                check_2ch(got, want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
            if count:
                assert ctx.stats()["n_fragments"] == nfrag, mib
            assert ctx.stats()["n_huge"] > 4096
    ctx.close()


def test_asymmetric_kernel_lut_uses_full_tables(native, mips):
    """Kernel G keeps only one quadrant of every mip level in LDS when the uploaded LUT is mirror-symmetric bit for bit
    (the reference's radial kernel is); any other LUT must go through the full tables.  Both against the oracle."""
    from oracle import oracle_np
    M, sf = oracle_np.transform_matrix(_rot(0.2, -0.4), np.zeros(3), 120.0)
    pos, h, m, q, _ = make_cloud(20000, seed=9)
    rs = np.random.RandomState(2)
    skew = mips.copy()
    skew *= (1.0 + 0.05 * rs.uniform(size=skew.shape)).astype(np.float32)      # no symmetry left, corners no longer zero
    for lut in (mips, skew):
        ctx = native.Context(300, 2)
        ctx.set_kernel_mips(lut)
        ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
        ctx.upload_quantity(q)
        ctx.render(M, sf)
        want, _ = oracle_render(pos, h, m, q, None, 0, M, sf, 300, lut)
        check_2ch(ctx.read_image(), want, abs_terms_image(pos, h, m, q, M, sf, 300, lut))
        ctx.close()


@pytest.mark.parametrize("mode", ["density", "weighted", "rgb", "depth"])
def test_mid_footprints_gather(native, mips, mode):
    """The footprints below 64 px that kernel S defers are drawn by kernel N (four records per wave step on 16-column strips)
    and / or kernel G (one record per wave step on 64-column strips), split by option mid_narrow_px_milli (64000 = all by N, the
    default; 0 = all by G; 24000 = both): against the oracle, exact fragment count included, for every work-item size, with more than 2048 footprints per wave strip
    (float32 accumulators folded into the float64 target), R = 300 (partial last tiles) and widths on the class boundaries and
    on every mip-level threshold."""
    from oracle import oracle_np
    R, scale, n = 300, 100.0, 30000
    M, sf = oracle_np.transform_matrix(_rot(0.15, -0.1), np.zeros(3), scale)
    rs = np.random.RandomState(23)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, :2] = rs.uniform(-1.1, 1.1, size=(n, 2)) * scale
    pos[:6000, :2] = rs.uniform(-0.1, 0.1, size=(6000, 2)) * scale          # a dense patch: > 2048 footprints on one strip
    pos[:, 2] = rs.uniform(-0.5, 0.5, n) * scale
    P = np.exp(rs.uniform(np.log(10.0), np.log(63.9), n))
    P[:48] = [63.999, 32.0, 32.0001, 16.0, 16.0001, 11.3137, 11.32, 15.99] * 6
    h = (P * scale / (2.0 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    ctx = native.Context(R, 4 if mode == "rgb" else 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None if mode == "rgb" else m)
    if mode == "rgb":
        ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
        want, nfrag = oracle_render(pos, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), 2, M, sf, R, mips)
        md = native.MODE_RGB
    elif mode == "depth":
        want, nfrag = oracle_render(pos, h, m, None, None, 1, M, sf, R, mips)
        md = native.MODE_DEPTH
    else:
        if mode == "weighted":
            ctx.upload_quantity(q)
        want, nfrag = oracle_render(pos, h, m, q if mode == "weighted" else None, None, 0, M, sf, R, mips)
        md = native.MODE_WEIGHTED
    ctx.set_option("p_small_milli", 0)         # everything below 64 px goes to the mid list
    for variant, items in ((64000, 0), (64000, 64), (64000, 8192), (0, 0), (0, 64), (0, 8192), (24000, 0), (45254, 256)):
        for count in (1, 0):
            ctx.set_option("mid_narrow_px_milli", variant)
            ctx.set_option("mid_item_records", items); ctx.set_option("count_fragments", count)
            ctx.render(M, sf, mode=md)
            got = ctx.read_image()
            st = ctx.stats()
            assert st["n_mid"] > 25000 and st["n_small"] == 0
            if mode == "rgb":
                assert np.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=0), (variant, items)
                assert np.array_equal(got[..., 3], want[..., 3]), (variant, items)
            elif mode == "weighted":
                check_2ch(got, want, abs_terms_image(pos, h, m, q, M, sf, R, mips))
            else:
                assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0), (variant, items)
                if mode == "depth":
                    assert np.allclose(got[..., 1], want[..., 1], rtol=1e-5, atol=1e-30), (variant, items)
            if count:
                assert st["n_fragments"] == nfrag, (variant, items)
    ctx.close()


@pytest.mark.parametrize("mode", ["weighted", "rgb"])
def test_mid_footprints_with_weights_that_are_not_finite(native, mips, mode):
    """Kernel N draws four records per wave step and lets a slot read zeros in the pixel rows its record does not cover:
    "0 x weight" must not become NaN there when a weight is infinite or NaN.  The fill pass flags such a list and the kernel
    then draws slot by slot.  Pixels the odd particles do not cover must equal the render without them; the pixels they cover
    are not finite in either kernel (inf x k, or NaN where k = 0), as in the generic kernel."""
    from oracle import oracle_np
    R, scale, n = 256, 100.0, 8000
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    rs = np.random.RandomState(41)
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, :2] = rs.uniform(-1.0, 1.0, size=(n, 2)) * scale
    P = np.exp(rs.uniform(np.log(17.0), np.log(60.0), n))
    h = (P * scale / (2.0 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    rgb = rs.uniform(0.1, 1.0, size=(n, 3)).astype(np.float32)
    odd = np.arange(0, n, 400)                       # 20 odd particles spread over the image
    m_odd, rgb_odd = m.copy(), rgb.copy()
    m_odd[odd[::2]] = np.inf; m_odd[odd[1::2]] = np.nan
    rgb_odd[odd[::2], 1] = np.inf; rgb_odd[odd[1::2], 2] = np.nan
    keep = np.ones(n, dtype=bool); keep[odd] = False
    md = native.MODE_RGB if mode == "rgb" else native.MODE_WEIGHTED
    images = {}
    for label, sel, mm, cc in (("clean", keep, m, rgb), ("odd", np.ones(n, dtype=bool), m_odd, rgb_odd)):
        for split in (64000, 0):
            ctx = native.Context(R, 4 if mode == "rgb" else 2)
            ctx.set_kernel_mips(mips)
            ctx.set_option("mid_narrow_px_milli", split)
            ctx.upload_particles(pos[sel, 0], pos[sel, 1], pos[sel, 2], h[sel], None if mode == "rgb" else mm[sel])
            if mode == "rgb":
                ctx.upload_rgb(cc[sel, 0].copy(), cc[sel, 1].copy(), cc[sel, 2].copy())
            ctx.render(M, sf, mode=md)
            assert ctx.stats()["n_mid"] == int(sel.sum())
            images[(label, split)] = ctx.read_image().astype(np.float64)
            ctx.close()
    clean = images[("clean", 64000)]
    nch = 3 if mode == "rgb" else 1
    for split in (64000, 0):
        got = images[("odd", split)]
        finite = np.isfinite(got[..., :nch]).all(axis=-1)
        # pixels no odd footprint square reaches: exactly the clean render's sums (same records, same order of accumulation classes)
        half = (P[odd] / 2.0)[:, None]
        px = np.arange(R) + 0.5
        pcx = (pos[odd, 0] / scale + 1.0) * R / 2.0; pcy = (1.0 - pos[odd, 1] / scale) * R / 2.0
        cx = np.abs(px[None, :] - pcx[:, None]) < half + 1e-3; cy = np.abs(px[None, :] - pcy[:, None]) < half + 1e-3
        touched = np.einsum("ky,kx->yx", cy.astype(np.int64), cx.astype(np.int64)) > 0
        assert finite[~touched].all(), "a pixel outside every odd footprint is not finite"
        assert np.allclose(got[~touched][:, :nch], clean[~touched][:, :nch], rtol=1e-5, atol=0), split
        assert (~finite).sum() > 0
    # (which of the TOUCHED pixels are finite may differ between the kernels: both skip the strips that lie outside the disc inscribed
    # in a footprint square -- where the kernel value is exactly 0 and "0 x inf" would be NaN -- and their strips differ in shape)
