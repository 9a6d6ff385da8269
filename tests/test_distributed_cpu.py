"""world_size-2 gloo test (CPU): the index-range sharding of render blocks and the sum-reduce of
the partial images reproduce the single-rank image.  The per-shard images come from the CPU oracle
(test infrastructure) because there is no GPU here; what is under test is the product's sharding
arithmetic (topsy_amd.distributed) and the N>1 reduction contract (sum, float32, root 0)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from conftest import make_cloud
    from oracle import oracle_c, oracle_np
    from topsy_amd import distributed, kernel_lut, progressive_render
    from topsy_amd.drawreason import DrawReason
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, R = 30011, 96
        pos, h, m, q, _ = make_cloud(n, seed=11)
        M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 180.0)
        mips = kernel_lut.kernel_mips()
        start, length = distributed.shard_range(n, rank, world)
        sl = slice(start, start + length)
        xs, ys, zs = (np.ascontiguousarray(pos[sl, k]) for k in range(3))
        # frame = the block sequence the (unchanged) RenderProgression hands out, here an EXPORT frame
        # split into several blocks plus a multi-range block as the cell-based progression produces
        rp = progressive_render.RenderProgression(n)
        rp.start_frame(DrawReason.EXPORT)
        blocks = [([0], [7000]), ([7000, 9000, 20000], [2000, 11000, 10011])]
        image = np.zeros((R, R, 2), dtype=np.float32)
        drawn = 0
        for starts, lens in blocks:
            s, l = distributed.intersect_ranges(starts, lens, start, length)
            assert (s >= 0).all() and (s + l <= length).all()
            drawn += int(l.sum())
            if len(s):
                oracle_c.splat(xs, ys, zs, h[sl], m[sl], q[sl], mode=0, M=M, sf=sf, R=R, mips=mips, ranges=(s, l), out=image)
        t = torch.from_numpy(image)
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)        # the one collective of the path
        total = torch.tensor([drawn])
        dist.all_reduce(total)
        assert int(total) == n, "every particle must be drawn by exactly one shard"
        if rank == 0:
            np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
        # the RCCL unique id travels through the same out-of-band channel in production
        payload = distributed.torch_broadcaster(dist)(b"x" * 128 if rank == 0 else None)
        assert payload == b"x" * 128
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_render_matches_single(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_cloud
    from oracle import oracle_c, oracle_np
    from topsy_amd import kernel_lut
    pos, h, m, q, _ = make_cloud(30011, seed=11)
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 180.0)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    want, _ = oracle_c.splat(x, y, z, h, m, q, mode=0, M=M, sf=sf, R=96, mips=kernel_lut.kernel_mips())
    got = np.load(tmp_path / "reduced.npy")
    np.testing.assert_allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)
    scale, _ = oracle_c.splat(x, y, z, h, m, np.abs(q), mode=0, M=M, sf=sf, R=96, mips=kernel_lut.kernel_mips())
    assert (np.abs(got[..., 1] - want[..., 1]) <= 1e-5 * scale[..., 1] + 1e-30).all()


def test_shard_arithmetic():
    from topsy_amd import distributed as d
    for n, g in [(10, 3), (1000000007, 8), (5, 8), (0, 2)]:
        b = d.shard_bounds(n, g)
        assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
    # blocks intersected with shards partition the blocks exactly
    rs = np.random.RandomState(0)
    n, g = 100000, 4
    starts = np.sort(rs.randint(0, n, 50)); lens = rs.randint(0, 3000, 50)
    lens = np.minimum(lens, n - starts)
    covered = np.zeros(n, dtype=np.int32); want = np.zeros(n, dtype=np.int32)
    for s, l in zip(starts, lens):
        want[s:s + l] += 1
    for r in range(g):
        s0, ln = d.shard_range(n, r, g)
        ss, ll = d.intersect_ranges(starts, lens, s0, ln)
        for s, l in zip(ss, ll):
            covered[s0 + s:s0 + s + l] += 1
    assert np.array_equal(covered, want)
    assert d.intersect_ranges([5], [10], 100, 50)[0].size == 0


def test_sharded_renderer_whole_set_block():
    """starts = lens = None means the whole snapshot, as in tsp_render: every rank draws exactly its shard."""
    from topsy_amd import distributed as d

    class FakeContext:
        def render(self, matrix, scale_factor, starts, lens, clear, mode, flags):
            self.call = (np.asarray(starts).tolist(), np.asarray(lens).tolist(), clear)
            return 0.0

    n, g = 1000003, 3
    drawn = 0
    for r in range(g):
        ctx = FakeContext()
        sr = d.ShardedRenderer(ctx, n, r, g)
        sr.render_block(np.eye(4), 1.0, None, None, clear=True)
        assert ctx.call == ([0], [sr.shard_len], True)
        drawn += sr.shard_len
    assert drawn == n
