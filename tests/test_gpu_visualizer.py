"""The reference's own render tests (tests/test_render_output.py, test_render_mode.py,
test_colormap.py) restated against the topsy_amd Visualizer on the GPU, with its golden vectors
and its tolerances."""
import numpy as np
import numpy.testing as npt
import pytest

import topsy_amd
from topsy_amd import config
from topsy_amd.drawreason import DrawReason

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[False, True], ids=["plain", "with_cells"])
def vis(request):
    v = topsy_amd.test(1000, render_resolution=200, with_cells=request.param)
    v.scale = 200.0
    yield v
    v.close()


@pytest.fixture(scope="module")
def kats(golden):
    return golden["reference_kats.npz"]


def test_render(vis, kats):                       # reference test_render :27-65
    result = vis.get_sph_presentation_image()
    assert result.dtype == np.uint8 and result.shape == (200, 200, 4)
    npt.assert_allclose(result[::20, ::20].ravel().astype(int), kats["test_render.reference_result"], atol=5)


def test_hdr_rgb_render(kats):                    # reference test_hdr_rgb_render :69-141
    v = topsy_amd.test(1000, render_resolution=200, render_mode="rgb-hdr")
    v.scale = 20.0
    v.colormap.update_parameters({"min_mag": 38.0, "max_mag": 40.0})
    result = v.get_sph_presentation_image()[..., :3]
    assert result.dtype == np.float16
    npt.assert_allclose(result[::20, ::20].ravel().astype(np.float64), kats["test_hdr_rgb_render.result_ref"], atol=1e-2)
    v.close()


def test_particle_pos_smooth(vis):                # reference :144-159
    if hasattr(vis.data_loader, "_cell_layout"):
        return
    npt.assert_allclose(vis.data_loader.get_pos_smooth()[::100][:2],
                        [[1.6189760e+01, -4.0728635e-01, -1.8409515e+01, 2.0848181e+01],
                         [-3.6236227e-01, 1.9854842e-02, -3.4908600e+00, 1.2997785e+00]], rtol=1e-7)


def test_sph_weighted_output(vis, kats):          # reference :161-198
    vis.quantity_name = "test-quantity"
    vis.scale = 20.0
    vis.rotate(0.0, 0.4)
    vis.render_sph(DrawReason.EXPORT)
    result = vis.get_sph_image()
    assert result.shape == (200, 200)
    npt.assert_allclose(result[::20, ::20].flatten(), kats["test_sph_weighted_output.expect"], atol=1.5e-7)


def test_sph_output(vis, kats):                   # reference :200-241
    vis.render_sph(DrawReason.EXPORT)
    result = vis.get_sph_image()
    assert result.shape == (200, 200)
    test, expect = result[::20, ::20].flatten(), kats["test_sph_output.expect"]
    npt.assert_allclose(test, expect, rtol=5e-1)
    assert abs((test / expect).mean() - 1.0) < 0.0015
    assert (test / expect).std() < 0.015


def test_rotated_sph_output(vis):                 # reference :280-293
    vis.draw(reason=DrawReason.EXPORT)
    unrotated = vis.get_sph_image()
    vis.rotation_matrix = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    vis.draw(reason=DrawReason.EXPORT)
    npt.assert_allclose(unrotated.T[:, ::-1], vis.get_sph_image(), rtol=5e-2)


def test_rgb_sph_output():                        # reference :296-300
    v = topsy_amd.test(1000, render_resolution=200, render_mode="rgb")
    assert v.get_sph_image().shape == (200, 200, 3)
    assert v._sph.get_image().shape == (200, 200, 4)
    v.close()


def test_depth_output(kats):                      # reference :302-343
    v = topsy_amd.test(1000, render_resolution=200)
    v.scale = 20.0
    v.rotation_matrix = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]], dtype=np.float32)
    v.render_sph(DrawReason.EXPORT)
    before = v.get_sph_image().copy()
    result = v._sph.get_depth_image(DrawReason.EXPORT)
    npt.assert_allclose(result[::20, ::20].ravel(), kats["test_depth_output.expect"], atol=1e-1)
    # the depth pass used the shared render target: the next read re-renders the scene itself
    npt.assert_allclose(v.get_sph_image(), before, rtol=1e-5)
    v.close()


def test_bivariate_splat_values(kats):            # reference test_bivariate_render :345-446 (splat part)
    v = topsy_amd.test(1000, render_resolution=200)
    v.quantity_name = "test-quantity"
    v.scale = 20.0
    v.rotate(0.0, 0.5)
    v.render_sph(DrawReason.EXPORT)
    raw = v._sph.get_image()
    npt.assert_allclose(raw[::20, ::20, 0].ravel(), kats["test_bivariate_render.expect_den"], rtol=2e-3)
    npt.assert_allclose(v.get_sph_image()[::20, ::20].ravel(), kats["test_bivariate_render.expect_qty"], atol=1e-4)
    v.close()


def test_render_mode_switching_and_errors():      # reference tests/test_render_mode.py
    v = topsy_amd.test(1000, render_resolution=64)
    v.scale = 20.0
    for mode in ("univariate", "bivariate", "rgb", "rgb-hdr", "univariate"):
        v.render_mode = mode
        result, pres = v.get_sph_image(), v.get_sph_presentation_image()
        assert pres.dtype == (np.float16 if mode.endswith("hdr") else np.uint8) and pres.shape == (64, 64, 4)
        assert result.shape == ((64, 64, 3) if mode.startswith("rgb") else ((64, 64, 2) if mode == "bivariate" else (64, 64)))
    with pytest.raises(ValueError, match="Invalid render_mode 'invalid'"):
        v.render_mode = "invalid"
    assert v.render_mode == "univariate"
    with pytest.raises(ValueError):
        v.render_mode = "surface"
    assert v.render_mode == "univariate"
    with pytest.raises(ValueError, match="Unable to get quantity"):
        v.quantity_name = "no-such-quantity"
    v.close()


def test_progressive_frames_fold_mass_scale():
    """CHANGE frame draws a prefix, REFINE frames add the rest without clearing; the colormap sees
    N/N_drawn (reference sph.py:306-332, implementation.py:427-453, visualizer.py:386-402)."""
    v = topsy_amd.test(200000, render_resolution=128)
    v.scale = 100.0
    full = v._sph.get_image().copy()
    rp = v._sph._render_progression
    rp._recommended_num_particles_to_render = 50000
    # pretend every block costs 40 ms of GPU time, so the 1/30 s budget admits one block per frame
    timer = v._sph._render_timer
    real_add = timer.add_block
    timer.add_block = lambda ms, wall_seconds=None: real_add(40.0)
    v.invalidate()
    v.draw(DrawReason.CHANGE)
    # the library reordered the particles into strata: the block was rounded to whole strata (unbiased sample)
    bounds = v.particle_buffers.block_boundaries
    drawn = rp._start_index
    assert len(bounds) == config.SPATIAL_ORDER_STRATA + 1 and bounds[0] == 0 and bounds[-1] == 200000 and drawn in set(bounds.tolist())
    assert abs(drawn - 50000) < 200000 // config.SPATIAL_ORDER_STRATA + 1000
    assert v._sph.last_render_mass_scale == pytest.approx(200000 / drawn, rel=1e-12)
    partial = v._sph.get_image()             # scaled by N/N_drawn
    ratio = partial[..., 0].sum() / full[..., 0].sum()
    assert 0.97 < ratio < 1.03
    assert v._pending_draw == DrawReason.REFINE
    while v._sph.needs_refine():
        v.draw(DrawReason.REFINE)
    assert v._sph.last_render_mass_scale == 1.0
    npt.assert_allclose(v._sph.get_image()[..., 0], full[..., 0], rtol=2e-5, atol=1e-20)
    v.draw(DrawReason.PRESENTATION_CHANGE)
    v.close()


def test_depth_query_between_progressive_frames():
    """A depth query goes through the shared render target (the reference's DepthSPH owns a texture of its own,
    sph.py:443-446).  The frames after it -- REFINE is the normal next one because _pending_draw stays REFINE after
    a partial CHANGE frame, and PRESENTATION_CHANGE after a colormap edit -- must redraw the scene instead of
    adding blocks onto / re-colouring the depth image."""
    v = topsy_amd.test(200000, render_resolution=128)
    v.scale = 100.0
    full = v._sph.get_image().copy()
    full_rgba = v.draw(DrawReason.EXPORT).copy()
    rp = v._sph._render_progression
    rp._recommended_num_particles_to_render = 50000
    timer = v._sph._render_timer
    real_add = timer.add_block
    timer.add_block = lambda ms, wall_seconds=None: real_add(40.0)        # one block per interactive frame
    v.invalidate()
    v.draw(DrawReason.CHANGE)
    assert v._pending_draw == DrawReason.REFINE and v._sph.needs_refine()
    depth = v.get_depth_image()
    assert depth.shape == (128, 128)
    assert v._pending_draw == DrawReason.CHANGE       # the visualizer knows the target was overwritten
    # even if the caller insists on REFINE, the renderer starts the frame again rather than refining the depth image
    v.draw(DrawReason.REFINE)
    for _ in range(64):
        if not v._sph.needs_refine():
            break
        v.draw(DrawReason.REFINE)
    assert v._sph.last_render_mass_scale == 1.0
    npt.assert_allclose(v._sph.get_image()[..., 0], full[..., 0], rtol=2e-5, atol=1e-20)
    # PRESENTATION_CHANGE straight after a depth query re-renders too (it would colour the depth image otherwise)
    timer.add_block = real_add
    v._sph.get_depth_image(DrawReason.EXPORT)
    out = v.draw(DrawReason.PRESENTATION_CHANGE)
    while v._sph.needs_refine():
        out = v.draw(DrawReason.REFINE)
    assert np.abs(out.astype(int) - full_rgba.astype(int)).max() <= 1
    v.close()


@pytest.mark.parametrize("mode", ["density", "weighted-average"])
@pytest.mark.parametrize("log_scale", [True, False], ids=["log", "linear"])
def test_colormap_vs_matplotlib(mode, log_scale):  # reference tests/test_colormap.py:35-105
    from matplotlib import colors, cm
    import matplotlib
    v = topsy_amd.test(100, render_resolution=200)
    img = np.empty((200, 200, 2), dtype=np.float32)
    img[:, :, 0] = np.logspace(-3, 0, 200)
    img[:, :, 1] = np.linspace(0, 1, 200)[:, np.newaxis] * img[:, :, 0]
    weighted = mode == "weighted-average"
    vmin, vmax = ((-2.0 if weighted else -3.0), 0.0) if log_scale else (0.0, 1.0)
    v.colormap.update_parameters({"type": "density", "weighted_average": weighted, "vmin": vmin, "vmax": vmax,
                                  "log": log_scale})
    image = v.colormap.sph_raw_output_to_image(img)
    assert image.shape == (200, 200, 4) and image.dtype == np.uint8
    content = v.colormap.sph_raw_output_to_content(img)
    with np.errstate(divide="ignore"):
        content = np.log10(content) if log_scale else content
    mpl = cm.ScalarMappable(norm=colors.Normalize(vmin=vmin, vmax=vmax),
                            cmap=matplotlib.colormaps[v.colormap.get_parameter("colormap_name")])
    npt.assert_allclose(image.astype(int), (mpl.to_rgba(content) * 255).astype(np.uint8).astype(int), atol=5)
    v.close()


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize("mode,quantity", [("univariate", None), ("univariate", "test-quantity"), ("rgb", None), ("rgb-hdr", None),
                                           ("bivariate", "test-quantity"), ("bivariate", None)])
def test_device_autorange_equals_host_autorange(mode, quantity):
    """autorange from device-side order statistics == autorange(get_image()) on the host."""
    v = topsy_amd.test(20000, render_resolution=256, render_mode=mode)
    v.scale = 40.0
    if quantity:
        v.quantity_name = quantity
    v.render_sph(DrawReason.EXPORT)
    for S in (1.0, 3.7):
        v._sph.last_render_mass_scale = S
        v.colormap.update_parameters({"vmin": 0.0, "vmax": 1.0})
        v.colormap.autorange(v._sph.get_image())
        host = v.colormap.get_parameters()
        v.colormap.update_parameters({"vmin": 0.0, "vmax": 1.0})
        v.colormap.autorange_on_device(S)
        dev = v.colormap.get_parameters()
        for k in ("vmin", "vmax", "log", "ui_range_linear", "ui_range_log", "density_vmin", "density_vmax", "ui_range_density"):
            if k in host:
                assert np.array_equal(np.asarray(host[k]), np.asarray(dev[k])), (k, host[k], dev[k])
    v.close()


def test_bivariate_render(kats):                  # reference test_bivariate_render :345-446, in full
    from oracle import oracle_c, oracle_np
    v = topsy_amd.test(1000, render_resolution=200, render_mode="bivariate")
    v.quantity_name = "test-quantity"
    v.scale = 20.0
    v.rotate(0.0, 0.5)
    v.render_sph(DrawReason.EXPORT)
    results = v.get_sph_image()
    mapped = v.get_sph_presentation_image()
    assert results.shape == (200, 200, 2) and mapped.shape == (200, 200, 4) and mapped.dtype == np.uint8
    npt.assert_allclose(results[::20, ::20, 0].ravel(), kats["test_bivariate_render.expect_den"], rtol=2e-3)
    npt.assert_allclose(results[::20, ::20, 1].ravel(), kats["test_bivariate_render.expect_qty"], atol=1e-4)
    npt.assert_allclose(mapped[::20, ::20].ravel().astype(int), kats["test_bivariate_render.expect_rgba"], atol=5)
    # and bit-exact against the oracle on the identical float buffer
    p = v.colormap.get_parameters()
    raw = v._sph._context.read_image()[..., :2]
    want = oracle_c.colormap_bivariate(raw, oracle_np.bivariate_lut(p["colormap_name"]), np.float32(p["vmin"]), np.float32(p["vmax"]),
                                       np.float32(p["density_vmin"]), np.float32(p["density_vmax"]), p["log"], p["weighted_average"])
    assert np.array_equal(mapped, want)
    v.close()


@pytest.mark.parametrize("log_scale", [True, False], ids=["log", "linear"])
def test_bivariate_colormap_vs_software(log_scale):   # reference tests/test_colormap.py:107-141 (software model, atol 5)
    from scipy.interpolate import RegularGridInterpolator
    v = topsy_amd.test(100, render_resolution=200)
    img = np.empty((200, 200, 2), dtype=np.float32)
    img[:, :, 0] = np.logspace(-3, 0, 200)
    img[:, :, 1] = np.linspace(0, 1, 200)[:, np.newaxis] * img[:, :, 0]
    vmin, vmax = (-2.0, 0.0) if log_scale else (0.0, 1.0)
    v.colormap.update_parameters({"type": "bivariate", "weighted_average": True, "vmin": vmin, "vmax": vmax,
                                  "density_vmin": -3.0, "density_vmax": 0.0, "log": log_scale})
    image = v.colormap.sph_raw_output_to_image(img)
    mapping = v.colormap._impl._generate_mapping_rgba_f32(1000)
    with np.errstate(divide="ignore"):
        w = img[..., 1] / img[..., 0]
        sd = (np.log10(img[..., 0]) + 3.0) / 3.0
        sv = (np.log10(w) - vmin) / (vmax - vmin) if log_scale else (w - vmin) / (vmax - vmin)
    pts = np.linspace(0, 1, 1000)
    soft = np.clip(RegularGridInterpolator((pts, pts), mapping, method="linear")(np.clip(np.stack((sv, sd), axis=-1), 0, 1)), 0, 1)
    npt.assert_allclose(image.astype(int), (soft * 255).astype(np.uint8).astype(int), atol=5)
    v.close()


def test_periodic_sph_output(kats):               # reference test_periodic_sph_output :243-278
    from oracle import oracle_np
    v = topsy_amd.test(1000, render_resolution=200, periodic_tiling=True)
    assert v.scale == 50.0                        # periodic TestDataLoader: box 100 -> initial half-width 50
    v.scale = 200.0
    v.render_sph(DrawReason.EXPORT)
    result = v.get_sph_image()
    npt.assert_allclose(result[::20, ::20].flatten(), kats["test_periodic_sph_output.expect"], rtol=1e-1)
    # rotated, non-integer shifts with fading weights: HIP post-pass == oracle bit for bit
    v.rotate(0.3, 0.2)
    v.scale = 130.0
    from topsy_amd import periodic_sph, sph
    sph.SPH.render(v._sph, DrawReason.EXPORT)     # the splat alone (two renders differ in the last bit: atomics)
    raw = v._sph._context.read_image()
    off, w = oracle_np.periodic_instances(v.rotation_matrix, 100.0 / 130.0)
    poff, pw = periodic_sph.instance_offsets_and_weights(v.rotation_matrix, 100.0 / 130.0)
    assert np.array_equal(off, poff) and np.array_equal(w, pw)
    v._sph._context.tile_periodic(poff, pw)       # what PeriodicSPH.render does next
    tiled = v._sph._context.read_image()
    assert np.array_equal(tiled, oracle_np.periodic_tile(raw, off, w))
    # and the product's own frame (render + tiling) agrees with it up to the summation order of the splat
    v.render_sph(DrawReason.EXPORT)
    npt.assert_allclose(v._sph._context.read_image(), tiled, rtol=1e-5, atol=1e-30)
    # a depth query goes through the shared render target; a PRESENTATION_CHANGE afterwards must redraw the periodic
    # frame (splat + tiling) instead of re-presenting the depth pass's image
    v._sph.get_depth_image()
    assert v._sph.render(DrawReason.PRESENTATION_CHANGE) is True
    npt.assert_allclose(v._sph._context.read_image(), tiled, rtol=1e-5, atol=1e-30)
    assert v._sph.render(DrawReason.PRESENTATION_CHANGE) is False      # now the frame is resident: nothing to do
    v.close()


def test_cell_progression_multi_range_blocks():
    """RenderProgressionWithCells hands out up to n_cells (start, len) ranges per block (reference
    progressive_render.py:152-187); partial frames + refinement must add up to the one-block EXPORT image."""
    v = topsy_amd.test(60000, render_resolution=160, with_cells=True)
    v.scale = 80.0
    full = v._sph.get_image().copy()                      # EXPORT: a single block spanning everything
    rp = v._sph._render_progression
    assert rp.get_max_particle_regions_per_block() == 1000
    rp._recommended_num_particles_to_render = 7000
    timer = v._sph._render_timer
    real_add = timer.add_block
    timer.add_block = lambda ms, wall_seconds=None: real_add(40.0)           # one block per frame
    v.invalidate()
    v.draw(DrawReason.CHANGE)
    starts, lens = v.particle_buffers.current_ranges()
    assert len(starts) > 100 and lens.sum() < 60000       # many per-cell ranges, a fraction of the particles
    frames = 1
    while v._sph.needs_refine():
        v.draw(DrawReason.REFINE)
        frames += 1
    assert frames > 3 and v._sph.last_render_mass_scale == 1.0
    npt.assert_allclose(v._sph.get_image(), full, rtol=1e-5, atol=1e-25)
    # view-sphere culling: zoomed in, cells outside the sphere are skipped yet the visible image is unchanged
    v.scale = 6.0
    v.position_offset = np.array([-6.0, -10.0, 0.0])      # centre on the (6, 10, 0) blob
    v.invalidate()
    v.draw(DrawReason.CHANGE)
    while v._sph.needs_refine():
        v.draw(DrawReason.REFINE)
    assert rp.get_fraction_volume_selected() < 0.5
    culled = v._sph.get_image().copy()
    rp.select_all()
    v._sph._context.render(*v._sph._get_transform_params(), clear=True)
    everything = v._sph._context.read_image()
    # particles in culled cells lie outside the 1.2 x scale sphere plus a cell diagonal: their footprints may
    # still graze the view, so compare where the culled render has substantial signal
    mask = culled[..., 0] > 1e-3 * culled[..., 0].max()
    assert mask.sum() > 1000
    npt.assert_allclose(culled[..., 0][mask], everything[..., 0][mask], rtol=0.05)
    v.close()


def test_first_interactive_block_is_a_whole_stratum():
    """A device-resident snapshot of 5e6 particles: the first interactive block (1e5 requested) is rounded up to
    one stratum (5e6 / config.SPATIAL_ORDER_STRATA), and that preview -- scaled by N/N_drawn -- carries the mass of the full image to 2 %,
    which a spatially compact index range would not."""
    v = topsy_amd.synthetic_on_device(5_000_000, render_resolution=256)
    v.scale = 200.0
    full = v._sph.get_image().copy()
    rp = v._sph._render_progression
    bounds = v.particle_buffers.block_boundaries
    assert len(bounds) == config.SPATIAL_ORDER_STRATA + 1 and (np.diff(bounds) > 0).all() and bounds[-1] == 5_000_000
    assert abs(np.diff(bounds) / (5_000_000 / config.SPATIAL_ORDER_STRATA) - 1.0).max() < 0.02        # uniform random strata
    timer = v._sph._render_timer
    real_add = timer.add_block
    timer.add_block = lambda ms, wall_seconds=None: real_add(40.0)           # one block per frame
    rp._recommended_num_particles_to_render = 100000      # the reference's first-frame guess (config.py:7)
    v.invalidate()
    v.draw(DrawReason.CHANGE)
    assert rp._start_index == bounds[1]
    preview = v._sph.get_image()
    assert abs(preview[..., 0].sum() / full[..., 0].sum() - 1.0) < 0.02
    # the preview is not confined to a corner: its centre of light agrees with the full image's to 2 px
    jj, ii = np.mgrid[0:256, 0:256]
    for img in (preview, full):
        w = img[..., 0] / img[..., 0].sum()
        img_c = ((w * ii).sum(), (w * jj).sum())
        if img is preview: c0 = img_c
    assert abs(c0[0] - img_c[0]) < 2.0 and abs(c0[1] - img_c[1]) < 2.0
    v.close()


def test_rgb_from_band_magnitudes_on_device():
    """rgb render from SSP band magnitudes: the device contraction (tsp_upload_band_magnitudes) equals the reference's host
    formula (loader.py:112-121, oracle_np.band_contraction) to one float32 ulp, with and without the load-time reordering,
    and the image equals the one rendered from host-computed rgb masses."""
    from oracle import oracle_np
    from topsy_amd import _native, kernel_lut, loader, visualizer
    rs = np.random.RandomState(11)
    n = 40000
    pos = (rs.normal(size=(n, 3)) * 20.0).astype(np.float32)
    h = np.exp(rs.uniform(np.log(0.05), np.log(8.0), n)).astype(np.float32)
    mags = {b: rs.uniform(2.0, 14.0, n) for b in "IVU"}
    mags["I"][::211] = np.nan
    ld = loader.ArrayDataLoader(None, pos=pos, smooth=h, mass=np.ones(n, np.float32), band_magnitudes=mags)
    m, w = ld.get_band_magnitudes()
    want = oracle_np.band_contraction(m, w)
    assert np.array_equal(want, ld.get_rgb_masses())
    for reorder in (False, True):
        ctx = _native.Context(128, 4)
        ctx.set_kernel_mips(kernel_lut.kernel_mips())
        ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None)
        perm = ctx.reorder_spatial(8, 3, want_permutation=True) if reorder else np.arange(n)
        ctx.upload_band_magnitudes(m, w)
        d = ctx.download_particles(("r", "g", "b"))
        got = np.stack([d["r"], d["g"], d["b"]], axis=1)
        npt.assert_allclose(got, want[np.asarray(perm)], rtol=1.2e-7, atol=0)       # float64 pow: libm vs device, then one rounding
        assert (got[np.isnan(m[0][np.asarray(perm)]), 0] == 0.0).all()
        # a general 3 x 3 mixing matrix
        w2 = rs.uniform(0.0, 1.0, size=(3, 3))
        ctx.upload_band_magnitudes(m, w2)
        d = ctx.download_particles(("r", "g", "b"))
        npt.assert_allclose(np.stack([d["r"], d["g"], d["b"]], axis=1), oracle_np.band_contraction(m, w2)[np.asarray(perm)], rtol=3e-7, atol=0)
        ctx.close()
    # through the product path: Visualizer in rgb mode over the loader with magnitudes == over one with host rgb arrays
    imgs = []
    for kw in ({"band_magnitudes": mags}, {"rgb": want}):
        v = visualizer.Visualizer(data_loader_class=loader.ArrayDataLoader, data_loader_kwargs=dict(pos=pos, smooth=h, mass=np.ones(n, np.float32), **kw),
                                 render_resolution=128, render_mode="rgb")
        v.scale = 80.0
        imgs.append(v._sph.get_image().copy())
        v.close()
    npt.assert_allclose(imgs[0][..., :3], imgs[1][..., :3], rtol=1e-5, atol=0)
    assert np.array_equal(imgs[0][..., 3], imgs[1][..., 3])


def test_band_contraction_on_device_pinned_by_reference_fixture(golden):
    """tsp_upload_band_magnitudes against the reference's own get_rgb_masses output (tests/golden/band_magnitudes.npz,
    reference src/topsy/loader.py:112-121): one float32 ulp (device pow vs libm, both float64, one rounding); NaN -> 0,
    inf and 0 reproduced exactly."""
    from topsy_amd import _native, kernel_lut
    g = golden["band_magnitudes.npz"]
    order = g["particle_order"]
    m = np.stack([g[b + "_mag"][order] for b in "IVU"])
    w = np.diag([0.5, 1.0, 1.0])
    want = g["rgb"]
    n = len(order)
    rs = np.random.RandomState(2)
    pos = rs.normal(size=(n, 3)).astype(np.float32)
    for reorder in (False, True):
        ctx = _native.Context(64, 4)
        ctx.set_kernel_mips(kernel_lut.kernel_mips())
        ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], np.ones(n, np.float32), None)
        perm = np.asarray(ctx.reorder_spatial(4, 9, want_permutation=True)) if reorder else np.arange(n)
        ctx.upload_band_magnitudes(m, w)
        d = ctx.download_particles(("r", "g", "b"))
        got = np.stack([d["r"], d["g"], d["b"]], axis=1)
        exp = want[perm]
        special = ~np.isfinite(exp) | (exp == 0)
        assert np.array_equal(got[special], exp[special])
        npt.assert_allclose(got[~special], exp[~special], rtol=1.2e-7, atol=0)
        ctx.close()


def test_view_culling_on_the_device_ordering():
    """SURVEY section 8 f3 on the library's own ordering: a snapshot WITHOUT a host cell layout is reordered by
    tsp_reorder_spatial (strata x Morton), the per-stratum cell runs come back through tsp_get_cell_offsets, and the plain
    RenderProgression culls by view sphere like the reference's cell progression (progressive_render.py:207-220,
    cell_layout.py:26-31).  Zoomed onto the (6, 10, 0) blob at scale 6, fewer than half of the particles are visited and
    the image is unchanged where it has signal."""
    from topsy_amd import cell_layout
    n = 4_000_000
    v = topsy_amd.synthetic_on_device(n, render_resolution=256)
    cells = v.particle_buffers.device_cells
    assert isinstance(cells, cell_layout.StratifiedCells) and cells.get_num_cells() >= 8
    lay = v.particle_buffers.context.cell_layout()
    off = lay["offsets"]
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 0).all()
    assert np.array_equal(off[::lay["cells_per_axis"] ** 3], v.particle_buffers.block_boundaries)
    # the cell runs really hold the particles of their cell: check a few runs against the downloaded positions
    d = v.particle_buffers.context.download_particles(("x", "y", "z"))
    pos = np.stack([d["x"], d["y"], d["z"]], axis=1).astype(np.float64)
    ncell = lay["cells_per_axis"] ** 3
    rs = np.random.RandomState(0)
    for e in rs.choice(len(off) - 1, 40, replace=False):
        a, b = off[e], off[e + 1]
        if b == a:
            continue
        code = e % ncell
        cxyz = np.array([sum(((code >> (3 * j + ax)) & 1) << j for j in range(4)) for ax in range(3)])
        lo = lay["box_lo"] + cxyz * lay["cell_width"]
        assert (pos[a:b] >= lo - 1e-3 * lay["cell_width"]).all() and (pos[a:b] <= lo + lay["cell_width"] * 1.001).all()
    # the whole view: nothing culled, plain (start, count) blocks
    v.scale = 200.0
    v.draw(DrawReason.CHANGE)
    while v._sph.needs_refine():
        v.draw(DrawReason.REFINE)
    rp = v._sph._render_progression
    assert rp.get_fraction_volume_selected() > 0.5
    timer = v._sph._render_timer
    real_add = timer.add_block
    timer.add_block = lambda ms, wall_seconds=None: real_add(40.0)           # one block per interactive frame

    def zoom(centre, scale):
        v.scale = scale
        v.position_offset = -np.asarray(centre, dtype=np.float64)
        rp._recommended_num_particles_to_render = 300000  # (a block that spans everything is drawn whole, as in the reference)
        v.draw(DrawReason.CHANGE)
        starts, lens = v.particle_buffers.current_ranges()
        n_ranges = len(starts)
        visited = int(np.sum(lens))
        frames = 1
        while v._sph.needs_refine():
            v.draw(DrawReason.REFINE)
            visited += int(np.sum(v.particle_buffers.current_ranges()[1]))
            frames += 1
            assert frames < 1000
        assert v._sph.last_render_mass_scale == 1.0
        culled = v._sph.get_image().copy()
        frac = rp.get_fraction_volume_selected()
        rp.select_all()
        v._sph._context.render(*v._sph._get_transform_params(), clear=True)
        everything = v._sph._context.read_image()
        # particles in culled cells lie outside the 1.2 x scale sphere plus a cell diagonal: their footprints may still
        # graze the view, so compare where the culled render has substantial signal
        mask = culled[..., 0] > 1e-3 * culled[..., 0].max()
        assert mask.sum() > 1000
        npt.assert_allclose(culled[..., 0][mask], everything[..., 0][mask], rtol=0.05)
        return visited, frac, n_ranges

    # zoomed onto the (6, 10, 0) blob (the case of the host-cell test above): fewer than half of the CELLS are picked; the
    # dense disc 10 kpc away lies inside the selection margin (1.2 x scale + one cell diagonal, as the reference picks
    # cells), so about 3/4 of the particles of this snapshot are still visited
    visited, frac, n_ranges = zoom([6.0, 10.0, 0.0], 6.0)
    assert frac < 0.5 and n_ranges > 1 and visited < 0.8 * n, (visited, frac)
    # zoomed onto the outskirts: a small fraction of the particles is visited
    visited, frac, n_ranges = zoom([45.0, 40.0, 10.0], 6.0)
    assert frac < 0.1 and visited < 0.1 * n, (visited, frac)
    v.close()
