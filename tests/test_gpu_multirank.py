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
