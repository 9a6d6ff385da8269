"""Chunk culling (option "chunk_cull", on by default; include/topsy_splat.h tsp_stats.n_chunk_culled): kernel S skips, unread,
the 512-particle chunks whose bounding box cannot reach the view.  It must change nothing but the time: the same image
(partial sums arrive in another order: 1e-5), the same class counts, the same fragment count, every particle accounted
for -- zoomed, rotated and off-centre cameras, ranges that are not chunk-aligned, degenerate particles inside the chunks,
re-uploaded positions (the bounds follow) and the oracle itself on a zoomed view.

The role in the reference: optional view culling of whole cells before the draw (src/topsy/progressive_render.py:207-220,
src/topsy/cell_layout.py:26-31); here it is finer (512 particles) and exact boxes are tested.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNT_KEYS = ("n_particles", "n_small", "n_mid", "n_huge", "n_culled", "n_fragments")


@pytest.fixture(scope="module")
def native():
    from topsy_amd import _native
    _native.load_library()
    return _native


def view(scale, rot=None, centre=(0.0, 0.0, 0.0)):
    from oracle import oracle_np
    return oracle_np.transform_matrix(np.eye(3) if rot is None else rot, -np.asarray(centre, dtype=np.float64), scale)


def rot(a, b):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    return np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])


def max_rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    den = np.maximum(np.abs(a), np.abs(b))
    lit = den > 0
    return float((np.abs(a - b)[lit] / den[lit]).max()) if lit.any() else 0.0


def both(ctx, M, sf, mode, starts=None, lens=None):
    """the same call with and without chunk culling: images, stats"""
    out = []
    for cull in (1, 0):
        ctx.set_option("chunk_cull", cull)
        ctx.render(M, sf, starts, lens, mode=mode)
        out.append((ctx.read_image(), ctx.stats()))
    ctx.set_option("chunk_cull", 1)
    return out


def check_same(on, off, expect_culling=True, signed_q_max=None):
    (img1, st1), (img0, st0) = on, off
    assert st0["n_chunk_culled"] == 0
    if expect_culling:
        assert st1["n_chunk_culled"] > 0, "no chunk was culled: the test does not test anything"
    assert st1["n_chunk_culled"] <= st1["n_culled"]
    for k in COUNT_KEYS:
        assert st1[k] == st0[k], (k, st1[k], st0[k])
    assert st1["n_small"] + st1["n_mid"] + st1["n_huge"] + st1["n_culled"] == st1["n_particles"]
    # two renders of one scene differ by the order in which float32 partial sums and float64 atomics arrive (a few 1e-7)
    if signed_q_max is None:
        assert max_rel(img1, img0) <= 1e-5
        assert np.array_equal(img1 == 0, img0 == 0)
    else:       # the weighted channel cancels (signed q): within 1e-5 of |q|max x density, as in test_gpu_scale.py
        assert max_rel(img1[..., 0], img0[..., 0]) <= 1e-5
        assert np.array_equal(img1[..., 0] == 0, img0[..., 0] == 0)
        assert (np.abs(img1[..., 1].astype(np.float64) - img0[..., 1]) <= 1e-5 * signed_q_max * img0[..., 0].astype(np.float64) + 1e-30).all()


CAMS = [("camera A", 200.0, None, (0, 0, 0)), ("zoom x8", 25.0, None, (0, 0, 0)), ("off-centre zoom", 30.0, None, (150.0, -90.0, 40.0)),
        ("rotated zoom", 40.0, (0.7, -0.4), (20.0, 10.0, -60.0)), ("far away", 50.0, None, (5000.0, 0.0, 0.0)),
        ("wide", 2000.0, (0.3, 0.2), (0, 0, 0))]


@pytest.mark.parametrize("cam", CAMS, ids=[c[0] for c in CAMS])
def test_culled_render_equals_unculled(native, mips, cam):
    _, scale, angles, centre = cam
    n, R = 6_000_000, 512
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 99, 0.0, with_quantity=True)
    ctx.reorder_spatial(8, 99)
    ctx.set_option("count_fragments", 1)
    M, sf = view(scale, None if angles is None else rot(*angles), centre)
    on, off = both(ctx, M, sf, native.MODE_WEIGHTED)
    # (camera A and the wide view see the whole snapshot -- 10 sigma of its widest component: nothing to cull, nothing may change)
    check_same(on, off, expect_culling=(cam[0] not in ("camera A", "wide")), signed_q_max=1e-4)
    if cam[0] == "far away":
        # (all but the few chunks that hold one of the snapshot's outermost particles: smoothing lengths of hundreds of units)
        assert on[1]["n_chunk_culled"] > 0.99 * n
    ctx.close()


def test_unaligned_ranges_and_modes(native, mips):
    """Ranges that start and end inside chunks (a chunk of the call then spans two bounds blocks), depth and rgb renders."""
    n, R = 5_000_000, 300
    M, sf = view(30.0, rot(0.2, 0.5), (40.0, 40.0, 0.0))
    starts = np.array([1000, 2_500_077, 4_100_001]); lens = np.array([1_200_333, 1_400_000, 899_999])
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 5, 0.0)
    ctx.reorder_spatial(4, 5)
    ctx.set_option("count_fragments", 1)
    for mode in (native.MODE_WEIGHTED, native.MODE_DEPTH):
        on, off = both(ctx, M, sf, mode, starts, lens)
        check_same(on, off)
        assert on[1]["n_particles"] == int(lens.sum())
    ctx.close()
    ctx = native.Context(R, 4)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 5, 0.0, with_rgb=True)
    ctx.reorder_spatial(4, 5)
    ctx.set_option("count_fragments", 1)
    on, off = both(ctx, M, sf, native.MODE_RGB, starts, lens)
    check_same(on, off)
    ctx.close()


def test_degenerate_particles_and_reupload(native, mips):
    """NaN / infinite coordinates and smoothing lengths inside otherwise ordinary chunks: such a chunk is kept or dropped, its
    finite particles are drawn either way; new positions bring new bounds."""
    rs = np.random.RandomState(3)
    n, R = 3_000_000, 256
    # spatially ordered by construction: blocks of 512 neighbours along a space-filling jitter of a 3-D lattice
    cells = rs.uniform(-300.0, 300.0, size=(n // 512 + 1, 3)).astype(np.float32)
    pos = (np.repeat(cells, 512, axis=0)[:n] + rs.normal(size=(n, 3)).astype(np.float32) * 2.0).astype(np.float32)
    h = np.exp(rs.uniform(np.log(0.02), np.log(3.0), size=n)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    bad = rs.choice(n, size=300, replace=False)
    pos[bad[:60], 0] = np.nan; pos[bad[60:120], 1] = np.inf; pos[bad[120:180], 2] = -np.inf
    h[bad[180:220]] = np.nan; h[bad[220:240]] = np.inf; h[bad[240:270]] = -1.0; h[bad[270:]] = 0.0
    h[rs.choice(n, size=40, replace=False)] = 80.0          # a few very wide footprints: their chunks reach the view from far away
    M, sf = view(40.0, rot(-0.3, 0.9), (30.0, -20.0, 10.0))
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    ctx.set_option("count_fragments", 1)
    on, off = both(ctx, M, sf, native.MODE_WEIGHTED)
    check_same(on, off)
    # the oracle on the same view (it visits every particle)
    from oracle import oracle_c
    want, nfrag = oracle_c.splat(np.ascontiguousarray(pos[:, 0]), np.ascontiguousarray(pos[:, 1]), np.ascontiguousarray(pos[:, 2]),
                                 h, m, None, None, mode=0, M=M, sf=sf, R=R, mips=mips)
    assert on[1]["n_fragments"] == nfrag
    assert max_rel(on[0][..., 0], want[..., 0]) <= 1e-5
    # move everything: the old bounds would cull what is now in view
    pos2 = pos.copy(); pos2[:, 0] += 250.0
    ctx.upload_particles(pos2[:, 0], pos2[:, 1], pos2[:, 2], h, m)
    on2, off2 = both(ctx, M, sf, native.MODE_WEIGHTED)
    check_same(on2, off2)
    assert on2[1]["n_fragments"] != on[1]["n_fragments"]
    ctx.close()


def test_small_calls_are_not_culled(native, mips):
    """Fewer than 4096 chunks in a call: the extra launch is not worth it, the statistics say so."""
    n, R = 1_000_000, 256
    ctx = native.Context(R, 2)
    ctx.set_kernel_mips(mips)
    ctx.generate_synthetic(n, 0, n, 7, 0.0)
    ctx.reorder_spatial(4, 7)
    M, sf = view(20.0)
    ctx.render(M, sf)
    st = ctx.stats()
    assert st["n_chunk_culled"] == 0 and st["n_culled"] > 0
    ctx.close()
