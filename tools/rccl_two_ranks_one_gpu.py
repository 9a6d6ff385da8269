import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist, numpy as np
from topsy_amd import _native, kernel_lut
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = _native.Context(128, 2, device_id=0); ctx.set_kernel_mips(kernel_lut.kernel_mips())
n = 100000; per = n // world
ctx.generate_synthetic(n, rank * per, per, 3, 0.0)
ids = [ctx.comm_unique_id() if rank == 0 else None]; dist.broadcast_object_list(ids, src=0)
try:
    ctx.comm_init(world, rank, ids[0])
    M = np.eye(4, dtype=np.float32); M[:3, :3] /= 100.0; M[2, :] = [0, 0, 0.005, 0.5]
    ctx.render(M, 0.01); ms = ctx.comm_reduce_image(0)
    img = ctx.read_image()
    if rank == 0:
        full = _native.Context(128, 2, device_id=0); full.set_kernel_mips(kernel_lut.kernel_mips()); full.generate_synthetic(n, 0, n, 3, 0.0); full.render(M, 0.01)
        ref = full.read_image()
        print("2-rank reduce on one GPU: max rel err", float(np.max(np.abs(img[..., 0] - ref[..., 0]) / np.maximum(ref[..., 0], 1e-30))), "ms", ms)
except Exception as e:
    print("rank", rank, "RCCL on a shared GPU:", str(e)[:300])
dist.destroy_process_group()
