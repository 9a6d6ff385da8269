"""The in-process multi-GPU driver (topsy_amd/multigpu.py) on CPU: `_native.Context` is replaced by a stand-in whose
render is the CPU oracle (test infrastructure), so what runs for real is the product's host logic -- ParticleBuffers,
SPH.render, RenderProgression, the shard arithmetic, the per-block fan-out and the one reduce per frame.
Reference behaviour matched: one frame = the blocks of the render progression, cleared on the first block only, the mass
scale N / N_drawn global (src/topsy/sph.py:306-332, src/topsy/visualizer.py:386-405)."""
import numpy as np
import pytest


class FakeContext:
    """Subset of _native.Context that ParticleBuffers / SPH / MultiGpuContext use, backed by the oracle."""
    log = []

    def __init__(self, resolution, n_channels, device_id=0):
        self.resolution, self.n_channels, self.device_id = resolution, n_channels, device_id
        self.active_channels = 2
        self.image = np.zeros((resolution, resolution, 2), dtype=np.float32)
        self.q = None
        self.n = 0
        self.n_renders = 0
        self.reordered = False
        FakeContext.log.append(self)

    n_gpus = 1

    def set_kernel_mips(self, mips, n0=64, n_levels=4):
        self.mips = np.asarray(mips, dtype=np.float32)

    def upload_particles(self, x, y, z, h, mass=None):
        self.x, self.y, self.z, self.h = (np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, z, h))
        self.m = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32)
        self.n = len(self.x)

    def upload_quantity(self, q):
        self.q = None if q is None else np.ascontiguousarray(q, dtype=np.float32)

    def reorder_spatial(self, n_strata=1, seed=1337, want_permutation=False):
        self.reordered = True
        self.n_strata = n_strata
        return np.arange(self.n, dtype=np.int64) if want_permutation else None

    def strata_offsets(self):
        if not self.reordered:
            return np.empty(0, dtype=np.int64)
        return np.linspace(0, self.n, self.n_strata + 1).astype(np.int64)

    def cell_layouts(self):
        return []

    @property
    def num_particles(self):
        return self.n

    def set_option(self, name, value):
        pass

    def render(self, matrix, scale_factor, starts=None, lens=None, clear=True, mode=0, flags=0):
        from oracle import oracle_c
        if clear:
            self.image[:] = 0
        if starts is None:
            starts, lens = [0], [self.n]
        starts, lens = np.asarray(starts, dtype=np.int64), np.asarray(lens, dtype=np.int64)
        self.last_ranges = (starts.copy(), lens.copy())
        self.n_renders += 1
        self.reduced = False
        if self.n and lens.sum() > 0:
            oracle_c.splat(self.x, self.y, self.z, self.h, self.m, self.q, mode=0, M=matrix, sf=float(scale_factor),
                           R=self.resolution, mips=self.mips, ranges=(starts, lens), out=self.image)
        return self.ms_per_block

    ms_per_block = 1.0

    def end_frame(self, root=0):
        return 0.0

    # a stand-in communicator with the contract of tsp_comm_*: every rank calls the reduce once per frame (the ranks meet
    # at a barrier, as the collective does), a second reduce of the same frame is refused
    groups = {}

    @staticmethod
    def comm_unique_id():
        import os
        return os.urandom(128)

    def comm_init(self, n_ranks, rank, unique_id):
        import threading
        g = FakeContext.groups.setdefault(unique_id, {"members": {}, "barrier": threading.Barrier(n_ranks)})
        g["members"][rank] = self
        self.group, self.rank, self.reduced = g, rank, False

    def comm_reduce_image(self, root=0):
        assert not self.reduced, "already reduced"
        self.group["barrier"].wait(timeout=30)
        if self.rank == root:
            total = self.image.astype(np.float64)
            for r, c in self.group["members"].items():
                if r != root:
                    total += c.image
            self.presented = total.astype(np.float32)
        self.group["barrier"].wait(timeout=30)
        self.reduced = True
        self.n_reduces = getattr(self, "n_reduces", 0) + 1
        return 0.25

    def read_image(self):
        if getattr(self, "reduced", False) and hasattr(self, "presented"):
            return self.presented.copy()          # the float32 presentation copy; the accumulator stays rank-local
        return self.image.copy()

    def write_image(self, img):
        self.image = np.ascontiguousarray(img, dtype=np.float32).copy()

    def set_reduced_image(self, total):
        # tsp_set_reduced_image: the caller's sum becomes the presentation copy, the accumulator stays this shard's
        assert not getattr(self, "reduced", False), "already reduced"
        self.presented = np.ascontiguousarray(total, dtype=np.float32).copy()
        self.reduced = True

    def stats(self):
        return {"n_particles": 0, "ms_total": 1.0}

    def download_particles(self, names=("x", "y", "z", "h", "mass")):
        src = {"x": self.x, "y": self.y, "z": self.z, "h": self.h, "mass": self.m, "q": self.q}
        return {k: src[k].copy() for k in names}

    def comm_destroy(self):
        self.comm_destroyed = True

    def close(self):
        pass


@pytest.fixture
def fake_backend(monkeypatch):
    from topsy_amd import _native
    FakeContext.log = []
    FakeContext.ms_per_block = 1.0
    monkeypatch.setattr(_native, "Context", FakeContext)
    return FakeContext


class _Vis:
    """The part of the Visualizer that SPH touches."""
    periodicity_scale = np.inf

    def __init__(self, n, resolution, device_ids, with_cells=False):
        from topsy_amd import loader, particle_buffers
        self.data_loader = loader.TestDataLoader(None, n, with_cells=with_cells)
        self.particle_buffers = particle_buffers.ParticleBuffers(self.data_loader, resolution, 0, 1, device_ids=device_ids)


def _frame(vis, reason, resolution, scale=60.0):
    from topsy_amd import sph
    s = sph.SPH(vis, resolution)
    s.scale = scale
    s.render(reason)
    return s


@pytest.mark.parametrize("n_gpus", [2, 3])
def test_sph_render_on_several_contexts_matches_one(fake_backend, n_gpus):
    from topsy_amd import multigpu
    from topsy_amd.drawreason import DrawReason
    n, R = 20010, 64
    one = _Vis(n, R, None)
    s1 = _frame(one, DrawReason.EXPORT, R)
    want = s1.get_image()
    many = _Vis(n, R, list(range(n_gpus)))
    ctx = many.particle_buffers.context
    assert isinstance(ctx, multigpu.MultiGpuContext) and ctx.collective == "rccl" and ctx.n_gpus == n_gpus
    sm = _frame(many, DrawReason.EXPORT, R)
    got = sm.get_image()
    np.testing.assert_allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
    # every particle went to exactly one shard, shards are index ranges, every shard saw every block
    shards = ctx.contexts
    assert sum(c.n for c in shards) == n
    assert [c.n for c in shards] == list(np.diff(ctx._bounds))
    assert len({c.n_renders for c in shards}) == 1
    assert all(c.n_reduces == 1 for c in shards), "exactly one reduce per frame on every rank"
    sm.get_image()
    assert all(c.n_reduces == 1 for c in shards), "a second read-back must not reduce again"
    assert sm.last_render_mass_scale == 1.0


def test_progressive_frames_refine_without_double_counting(fake_backend):
    """Interactive frame (a prefix), REFINE frames (clear = False) until complete: after each frame the presented image is
    the sum of the shards' partial images -- the reduced copy never leaks back into a shard's accumulator."""
    from topsy_amd.drawreason import DrawReason
    n, R = 30010, 48
    one = _Vis(n, R, None)
    want = _frame(one, DrawReason.EXPORT, R).get_image()
    many = _Vis(n, R, [0, 0])                   # two contexts on one device: the host collective by construction
    ctx = many.particle_buffers.context
    assert ctx.collective == "host"
    from topsy_amd import sph
    s = sph.SPH(many, R)
    s.scale = 60.0
    s._render_progression._recommended_num_particles_to_render = 7000
    fake_backend.ms_per_block = 25.0           # one block uses up the 1/30 s budget of an interactive frame
    s.render(DrawReason.CHANGE)
    assert s.needs_refine()
    frames = 1
    partial = s.get_image()                    # scaled by N / N_drawn: roughly the final image already
    assert 0.5 < partial[..., 0].sum() / want[..., 0].sum() < 2.0
    while s.needs_refine():
        s._render_progression._recommended_num_particles_to_render = 9000
        s.render(DrawReason.REFINE)
        s.get_image()                           # a presentation between the frames (triggers the lazy reduce path too)
        frames += 1
        assert frames < 20
    assert frames >= 3
    got = s.get_image()
    assert s.last_render_mass_scale == 1.0
    np.testing.assert_allclose(got[..., 0], want[..., 0], rtol=2e-5, atol=0)


def test_multi_context_block_boundaries_and_ranges(fake_backend):
    from topsy_amd import multigpu
    ctx = multigpu.MultiGpuContext(32, 4, [0, 1, 2])
    n = 1000
    rs = np.random.RandomState(3)
    x = rs.normal(size=n).astype(np.float32)
    ctx.set_kernel_mips(np.zeros(5440, dtype=np.float32))
    ctx.upload_particles(x, x, x, np.ones(n, np.float32), np.ones(n, np.float32))
    assert ctx.num_particles == n and list(ctx._bounds) == [0, 333, 666, 1000]
    perm = ctx.reorder_spatial(30, 1, want_permutation=True)
    assert np.array_equal(perm, np.arange(n))                                   # the stand-in keeps the order; shards are offset correctly
    offs = ctx.strata_offsets()
    assert offs[0] == 0 and offs[-1] == n and (np.diff(offs) > 0).all() and {333, 666} <= set(offs.tolist())
    assert len(offs) == 3 * 10 + 1                                              # ceil(30 / 3) strata per shard
    # a block that touches only the middle shard: the others still take part with an empty selection (clear must reach them)
    ctx.render(np.eye(4, dtype=np.float32), 1.0, [400], [100], clear=True)
    r = [c.last_ranges for c in ctx.contexts]
    assert r[0][1].sum() == 0 and r[2][1].sum() == 0 and (r[1][0].tolist(), r[1][1].tolist()) == ([67], [100])
    ctx.render(np.eye(4, dtype=np.float32), 1.0, [300, 660], [100, 10], clear=False)
    r = [c.last_ranges for c in ctx.contexts]
    assert (r[0][0].tolist(), r[0][1].tolist()) == ([300], [33])
    assert (r[1][0].tolist(), r[1][1].tolist()) == ([0, 327], [67, 6])
    assert (r[2][0].tolist(), r[2][1].tolist()) == ([0], [4])
    ctx.close()


def test_interleaved_assignment_index_maps(fake_backend):
    """Block-cyclic shards: every shard's global -> local index map is monotone, so a global (start, len) range is ONE local
    range per shard (what global_to_split_monotonic needs of a split, split_buffers.py:78-116), the shards partition every
    range, and uploads / downloads / quantity swaps address the same particles."""
    from topsy_amd import multigpu
    n, B, G = 10_037, 64, 3
    ctx = multigpu.MultiGpuContext(32, 4, [0, 1, 2], assignment="interleaved", interleave_block=B)
    rs = np.random.RandomState(5)
    x = np.arange(n, dtype=np.float32)                       # x = the particle's global index
    ctx.set_kernel_mips(np.zeros(5440, dtype=np.float32))
    ctx.upload_particles(x, x, x, np.ones(n, np.float32), np.ones(n, np.float32))
    assert ctx.num_particles == n and sum(c.n for c in ctx.contexts) == n
    for g, c in enumerate(ctx.contexts):
        own = c.x.astype(np.int64)
        assert (np.diff(own) > 0).all() and ((own // B) % G == g).all()
        assert max(abs(c.n - n / G) for c in ctx.contexts) <= B
    q = rs.normal(size=n).astype(np.float32)
    ctx.upload_quantity(q)
    for c in ctx.contexts:
        assert np.array_equal(c.q, q[c.x.astype(np.int64)])
    back = ctx.download_particles(("x", "q"))
    assert np.array_equal(back["x"], x) and np.array_equal(back["q"], q)
    for _ in range(50):
        k = rs.randint(1, 6)
        starts = np.sort(rs.randint(0, n, size=k)).astype(np.int64)
        lens = rs.randint(0, 700, size=k).astype(np.int64)
        lens = np.minimum(lens, n - starts)
        ctx.render(np.eye(4, dtype=np.float32), 1.0, starts, lens, clear=True)
        drawn = []
        for c in ctx.contexts:
            ls, ll = c.last_ranges
            assert len(ls) <= k
            for a, b in zip(ls, ll):
                drawn.append(c.x[a:a + b].astype(np.int64))
        drawn = np.sort(np.concatenate(drawn)) if drawn else np.empty(0, dtype=np.int64)
        want = np.sort(np.concatenate([np.arange(a, a + b) for a, b in zip(starts, lens)] + [np.empty(0, dtype=np.int64)]))
        assert np.array_equal(drawn, want)
    # after a library reordering the index space is the concatenation of the shards; the permutation maps back to the caller's order
    perm = ctx.reorder_spatial(6, 1, want_permutation=True)
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(perm, np.concatenate([c.x.astype(np.int64) for c in ctx.contexts]))
    ctx.render(np.eye(4, dtype=np.float32), 1.0, [0], [ctx.contexts[0].n], clear=True)
    r = [c.last_ranges for c in ctx.contexts]
    assert r[0][1].sum() == ctx.contexts[0].n and r[1][1].sum() == 0 and r[2][1].sum() == 0
    q2 = rs.normal(size=n).astype(np.float32)
    ctx.upload_quantity(q2)                                   # still given in the caller's order
    for c in ctx.contexts:
        assert np.array_equal(c.q, q2[c.x.astype(np.int64)])
    ctx.close()
    assert all(getattr(c, "comm_destroyed", False) for c in ctx.contexts)
    with pytest.raises(AttributeError):
        ctx.write_image                                        # root-only mutations are not forwarded


def test_cell_sorted_loader_is_sharded_interleaved_and_renders_the_same_image(fake_backend):
    """A loader with its own cell layout hands over spatially sorted particles (reference loader.py:88-97): 'auto' deals them
    block-cyclically, the cell progression's (start, len) ranges reach every shard, and the frame equals the one-context frame."""
    from topsy_amd import multigpu
    from topsy_amd.drawreason import DrawReason
    n, R = 30_000, 64
    one = _Vis(n, R, None, with_cells=True)
    want = _frame(one, DrawReason.EXPORT, R).get_image()
    many = _Vis(n, R, [0, 1, 2, 3], with_cells=True)
    ctx = many.particle_buffers.context
    assert isinstance(ctx, multigpu.MultiGpuContext) and ctx.assignment == "interleaved"
    assert max(c.n for c in ctx.contexts) - min(c.n for c in ctx.contexts) <= ctx.interleave_block
    got = _frame(many, DrawReason.EXPORT, R).get_image()
    np.testing.assert_allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
    plain = _Vis(n, R, [0, 1, 2, 3], with_cells=False)
    assert plain.particle_buffers.context.assignment == "contiguous"
