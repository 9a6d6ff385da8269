"""Two ranks on two GPUs: index-range shards + ONE RCCL sum-reduce reproduce the single-GPU image.
Skipped on a 1-GPU box (RCCL refuses two ranks on one device, tools/rccl_two_ranks_one_gpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from topsy_amd import _native, distributed, kernel_lut
    dist.init_process_group("gloo", rank=rank, world_size=world)         # host channel for the 128-byte RCCL id only
    try:
        n, R = 2_000_000, 512
        ctx = _native.Context(R, 2, device_id=rank)
        ctx.set_kernel_mips(kernel_lut.kernel_mips())
        start, length = distributed.shard_range(n, rank, world)
        ctx.generate_synthetic(n, start, length, 1337, 0.0)
        ctx.reorder_spatial(32, 1337)
        distributed.init_comm(ctx, rank, world, distributed.torch_broadcaster(dist))
        M = np.eye(4, dtype=np.float32)
        M[:3, :3] /= 150.0
        M[2, :] = [0.0, 0.0, 0.5 / 150.0, 0.5]
        sr = distributed.ShardedRenderer(ctx, n, rank, world)
        # a frame of two blocks (global index ranges, clipped to the shard), then the one reduce
        sr.render_block(M, 1.0 / 150.0, [0], [n // 3], clear=True)
        sr.render_block(M, 1.0 / 150.0, [n // 3], [n - n // 3], clear=False)
        ms = sr.reduce(root=0)
        assert ms >= 0.0
        with pytest.raises(_native.BackendError, match="already reduced"):
            sr.reduce(root=0)
        if rank == 0:
            np.save(os.path.join(out_dir, "reduced.npy"), ctx.read_image())
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_gpu_sharded_render_matches_single_gpu(tmp_path):
    from topsy_amd import _native, kernel_lut
    if _native.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    n, R = 2_000_000, 512
    ctx = _native.Context(R, 2, device_id=0)
    ctx.set_kernel_mips(kernel_lut.kernel_mips())
    ctx.generate_synthetic(n, 0, n, 1337, 0.0)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] /= 150.0
    M[2, :] = [0.0, 0.0, 0.5 / 150.0, 0.5]
    ctx.render(M, 1.0 / 150.0)
    want = ctx.read_image()
    ctx.close()
    got = np.load(tmp_path / "reduced.npy")
    assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0)


def _reference_frames(vis):
    """What the UI layers would ask of a Visualizer: an export-quality image, its colormapped presentation, autorange."""
    from topsy_amd.drawreason import DrawReason
    vis.scale = 60.0
    vis.rotate(0.2, 0.3)
    vis.render_sph(DrawReason.EXPORT)
    img = vis._sph.get_image().copy()
    rgba = vis.get_sph_presentation_image().copy()
    return img, rgba, dict(vis.colormap.get_parameters())


@pytest.mark.parametrize("device_ids", [[0, 0], [0, 0, 0]])
def test_visualizer_on_several_contexts_of_one_device(device_ids):
    """The multi-GPU driver behind the Visualizer (topsy_amd/multigpu.py) with every context on device 0 -- RCCL refuses two
    ranks on one device, so the shards are summed through the host, but everything else is the real path: sharded
    upload, per-shard load-time reordering, concurrent tsp_render calls from one thread per context, one reduce per
    frame, colormap / autorange on the first context.  Equals the one-context Visualizer within 1e-5."""
    import topsy_amd
    n, R = 200000, 256
    one = topsy_amd.test(n, render_resolution=R)
    want_img, want_rgba, want_params = _reference_frames(one)
    one.close()
    many = topsy_amd.test(n, render_resolution=R, device_ids=device_ids)
    ctx = many.particle_buffers.context
    assert ctx.n_gpus == len(device_ids) and ctx.collective == "host" and ctx.num_particles == n
    got_img, got_rgba, got_params = _reference_frames(many)
    assert np.allclose(got_img[..., 0], want_img[..., 0], rtol=1e-5, atol=0)
    assert abs(got_params["vmin"] - want_params["vmin"]) < 1e-4 and abs(got_params["vmax"] - want_params["vmax"]) < 1e-4
    assert (np.abs(got_rgba.astype(int) - want_rgba.astype(int)) <= 1).all()
    # weighted quantity + progressive refinement: the frame is completed by REFINE frames without double counting
    from topsy_amd.drawreason import DrawReason
    many.quantity_name = "test-quantity"
    many._sph._render_progression._recommended_num_particles_to_render = 30000
    many.draw(DrawReason.CHANGE)
    guard = 0
    while many._sph.needs_refine():
        many.draw(DrawReason.REFINE)
        guard += 1
        assert guard < 200
    done = many._sph.get_image().copy()
    many.render_sph(DrawReason.EXPORT)
    exp = many._sph.get_image()
    assert np.allclose(done[..., 0], exp[..., 0], rtol=2e-5, atol=0)
    many.close()


def test_visualizer_two_gpus_rccl():
    """topsy_amd.test(n, n_gpus=2): two devices driven from one process, RCCL sum-reduce over xGMI.  Skipped on a
    single-GPU box."""
    import topsy_amd
    from topsy_amd import _native
    if _native.device_count() < 2:
        pytest.skip("needs two GPUs")
    n, R = 2_000_000, 512
    one = topsy_amd.test(n, render_resolution=R)
    want_img, want_rgba, _ = _reference_frames(one)
    one.close()
    two = topsy_amd.test(n, render_resolution=R, n_gpus=2)
    assert two.particle_buffers.context.collective == "rccl"
    got_img, got_rgba, _ = _reference_frames(two)
    assert np.allclose(got_img[..., 0], want_img[..., 0], rtol=1e-5, atol=0)
    assert (np.abs(got_rgba.astype(int) - want_rgba.astype(int)) <= 1).all()
    two.close()


def test_rgb_and_depth_on_several_contexts_of_one_device():
    """rgb mode (band magnitudes contracted on every shard's device), the depth pass and the periodic tiling through the
    multi-context driver equal the one-context results."""
    import topsy_amd
    from topsy_amd import loader, visualizer
    from topsy_amd.drawreason import DrawReason
    rs = np.random.RandomState(4)
    n, R = 60000, 128
    pos = (rs.normal(size=(n, 3)) * 20.0).astype(np.float32)
    h = np.exp(rs.uniform(np.log(0.05), np.log(8.0), n)).astype(np.float32)
    mags = {b: rs.uniform(2.0, 14.0, n) for b in "IVU"}
    imgs, depths = [], []
    for dev in (None, [0, 0]):
        v = visualizer.Visualizer(data_loader_class=loader.ArrayDataLoader,
                                  data_loader_kwargs=dict(pos=pos, smooth=h, mass=np.ones(n, np.float32), band_magnitudes=mags),
                                  render_resolution=R, render_mode="rgb", device_ids=dev)
        v.scale = 80.0
        imgs.append(v._sph.get_image().copy())
        depths.append(v.get_depth_image().copy())
        v.close()
    assert np.allclose(imgs[0][..., :3], imgs[1][..., :3], rtol=1e-5, atol=0)
    assert np.array_equal(imgs[0][..., 3], imgs[1][..., 3])
    fin = np.isfinite(depths[0]) & np.isfinite(depths[1])
    assert fin.sum() > 1000 and np.allclose(depths[0][fin], depths[1][fin], rtol=1e-4, atol=1e-3)
    # periodic tiling on the first context after the reduce
    outs = []
    for dev in (None, [0, 0]):
        v = topsy_amd.test(2000, render_resolution=R, periodic_tiling=True, device_ids=dev)
        v.scale = 150.0
        v.render_sph(DrawReason.EXPORT)
        outs.append(v.get_sph_image().copy())
        v.close()
    assert np.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-30)


def test_interleaved_shards_balance_a_cell_sorted_snapshot():
    """A loader with its own cell layout hands the particles over sorted by spatial cell (reference loader.py:88-97): contiguous
    index-range shards are then spatial slabs -- the dense core on one GPU, the fragment-heavy outskirts on others.  The
    block-cyclic assignment ('interleaved', what 'auto' picks for such loaders) gives every shard the same mix: per-shard GPU
    time max / mean <= 1.3, the image equal to the one-context image at 1e-5.  The four contexts share device 0 here, so
    the shards render ONE AFTER ANOTHER for clean per-shard timings (a one-thread pool); split arithmetic preserved:
    split_buffers.py:78-116."""
    from concurrent.futures import ThreadPoolExecutor
    import topsy_amd
    from topsy_amd.drawreason import DrawReason
    n, R = 4_000_000, 1024
    one = topsy_amd.test(n, render_resolution=R, with_cells=True)
    one.scale = 200.0
    one.render_sph(DrawReason.EXPORT)
    want = one._sph.get_image().copy()
    one.close()
    ratios = {}
    for assignment in ("contiguous", "interleaved", None):
        v = topsy_amd.test(n, render_resolution=R, with_cells=True, device_ids=[0, 0, 0, 0], shard_assignment=assignment)
        ctx = v.particle_buffers.context
        assert ctx.assignment == (assignment or "interleaved")          # None = config 'auto' = interleaved for a cell-sorted loader
        ctx._pool = ThreadPoolExecutor(max_workers=1)                   # sequential shards: timings do not disturb each other
        v.scale = 200.0
        best = None
        for _ in range(3):
            v.render_sph(DrawReason.EXPORT)
            ms = np.array([s["ms_total"] for s in ctx.per_shard_stats()])
            best = ms if best is None else np.minimum(best, ms)
        got = v._sph.get_image().copy()
        assert np.allclose(got[..., 0], want[..., 0], rtol=1e-5, atol=0), assignment
        ratios[assignment or "auto"] = float(best.max() / best.mean())
        v.close()
    print("per-shard ms_total max/mean:", ratios)
    assert ratios["interleaved"] <= 1.3 and ratios["auto"] <= 1.3
    assert ratios["contiguous"] > ratios["interleaved"], "the cell-sorted input is what the interleaved assignment is for"
