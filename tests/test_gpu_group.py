"""The C-level device group (tsp_group_*, include/topsy_splat.h): several contexts behind one handle with the host-thread
choreography inside the library -- index-range shards (reference split arithmetic: src/topsy/split_buffers.py:26-38,78-116),
concurrent per-shard tsp_render, ONE image reduce per frame.  On a single-GPU box the contexts share device 0 and the reduce goes
through the host; with two GPUs the same calls use RCCL (skipped otherwise)."""
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


def close(a, b, rtol=1e-5):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return (np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b)) + 1e-30).all()


@pytest.mark.parametrize("devices", [[0, 0, 0], [0, 1]])
def test_group_frame_equals_one_context(native, mips, devices):
    if len(set(devices)) > 1 and native.device_count() < len(set(devices)):
        pytest.skip("needs two GPUs")
    n, R = 3_000_001, 512
    M, sf = camera(200.0)
    one = native.Context(R, 2)
    one.set_kernel_mips(mips)
    one.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True)
    one.set_option("count_fragments", 1)
    one.render(M, sf)
    want = one.read_image()
    st1 = one.stats()
    grp = native.Group(R, 2, devices)
    assert grp.size == len(devices) and grp.uses_rccl == (len(set(devices)) == len(devices))
    grp.set_kernel_mips(mips)
    grp.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=True)
    assert grp.num_particles == n
    assert [grp.member(g).num_particles for g in range(grp.size)] == [(n * (g + 1)) // grp.size - (n * g) // grp.size for g in range(grp.size)]
    assert [grp.shard_range(g) for g in range(grp.size)] == [((n * g) // grp.size, (n * (g + 1)) // grp.size - (n * g) // grp.size) for g in range(grp.size)]
    grp.set_option("count_fragments", 1)
    ms = grp.render(M, sf)
    assert ms > 0.0
    grp.end_frame()
    assert grp.end_frame() == 0.0                      # nothing rendered since: no second reduce
    got = grp.root.read_image()
    st = grp.stats()
    for k in ("n_particles", "n_fragments", "n_culled"):
        assert st[k] == st1[k], k
    assert st["n_small"] + st["n_mid"] + st["n_huge"] + st["n_culled"] == n
    assert close(got[..., 0], want[..., 0])
    assert (np.abs(got[..., 1].astype(np.float64) - want[..., 1]) <= 1e-5 * 1e-4 * want[..., 0].astype(np.float64) + 1e-30).all()
    # the colormap runs on the root after the reduce, like on a single context
    import matplotlib
    lut = matplotlib.colormaps["viridis"](np.linspace(0.001, 0.999, 1000)).astype(np.float32)
    a = grp.root.colormap_scalar(lut, -10.0, -5.0, True, False)
    b = one.colormap_scalar(lut, -10.0, -5.0, True, False)
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all()
    # a frame in blocks with a presentation (reduce) in between: REFINE continues from every shard's own accumulator
    cut = n // 3 + 17
    grp.render(M, sf, [0], [cut], clear=True)
    grp.end_frame()
    part = grp.root.read_image()
    one.render(M, sf, [0], [cut])
    assert close(part[..., 0], one.read_image()[..., 0])
    grp.render(M, sf, [cut, 5], [2**62, 0], clear=False)           # "to the end" + an empty range
    grp.end_frame()
    assert close(grp.root.read_image()[..., 0], want[..., 0])
    # ... and the accumulators stayed shard-local and unrounded: the two-block frame is the one-block frame of the group to the
    # last bit of float32 accumulation order (same shards, same float64 partial sums, rounded once), on either collective
    two_blocks = grp.root.read_image().copy()
    grp.render(M, sf)
    grp.end_frame()
    one_block = grp.root.read_image()
    assert np.abs(two_blocks[..., 0].astype(np.float64) - one_block[..., 0]).max() <= 2e-7 * one_block[..., 0].max()
    with pytest.raises(native.BackendError, match="already reduced"):
        grp.root.set_reduced_image(one_block)              # the presentation image is handed over once per frame
    # a block that touches only the last shard still clears the others
    grp.render(M, sf, [n - 1000], [1000], clear=True)
    grp.end_frame()
    one.render(M, sf, [n - 1000], [1000])
    assert close(grp.root.read_image()[..., 0], one.read_image()[..., 0])
    assert grp.stats()["n_particles"] == 1000
    grp.close()
    one.close()


def test_group_uploads_and_errors(native, mips):
    rs = np.random.RandomState(2)
    n, R = 100_003, 256
    pos = (rs.normal(size=(n, 3)) * 30.0).astype(np.float32)
    h = np.exp(rs.uniform(np.log(0.05), np.log(30.0), n)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    M, sf = camera(100.0)
    one = native.Context(R, 2)
    one.set_kernel_mips(mips)
    one.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    one.upload_quantity(q)
    one.render(M, sf)
    want = one.read_image().astype(np.float64)
    grp = native.Group(R, 2, [0, 0])
    grp.set_kernel_mips(mips)
    grp.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, m)
    grp.upload_quantity(q)
    grp.reorder_spatial(8, 3)
    grp.render(M, sf)
    grp.end_frame()
    got = grp.root.read_image().astype(np.float64)
    assert close(got[..., 0], want[..., 0])
    scale = np.abs(want[..., 1]).max()
    assert np.abs(got[..., 1] - want[..., 1]).max() <= 1e-4 * scale          # signed quantity: cancelling sums, other order
    with pytest.raises(native.BackendError, match="negative length"):
        grp.render(M, sf, [0], [-5])
    # a member touched behind the group's back (another channel layout) is refused on the calling thread, before any collective
    grp.render(M, sf)
    grp.member(1).set_reduced_image(np.zeros((R, R, 2), dtype=np.float32))
    with pytest.raises(native.BackendError, match="context 1 was already reduced"):
        grp.end_frame()
    grp.close()
    one.close()
    # rgb through the group: band magnitudes cut per shard == the one-context contraction
    mags = rs.uniform(2.0, 9.0, size=(3, n))
    W = np.diag([0.5, 1.0, 1.0])
    one = native.Context(R, 4)
    one.set_kernel_mips(mips)
    one.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None)
    one.upload_band_magnitudes(mags, W)
    one.render(M, sf, mode=native.MODE_RGB)
    want = one.read_image().astype(np.float64)
    grp = native.Group(R, 4, [0, 0, 0])
    grp.set_kernel_mips(mips)
    grp.upload_particles(pos[:, 0], pos[:, 1], pos[:, 2], h, None)
    grp.upload_band_magnitudes(mags, W)
    grp.render(M, sf, mode=native.MODE_RGB)
    grp.end_frame()
    got = grp.root.read_image().astype(np.float64)
    for c in range(3):
        assert close(got[..., c], want[..., c])
    assert np.array_equal(got[..., 3], want[..., 3])
    grp.close()
    one.close()
    with pytest.raises(native.BackendError):
        native.Group(R, 2, [0, 10_000])                    # no such device: nothing leaks, the error names it
