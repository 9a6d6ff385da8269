"""Multi-GPU driver: one process per GPU, particles sharded by global index range, ONE sum-reduce
of the float32 image per frame over RCCL/xGMI.  The reference is single-GPU;
the sharding arithmetic follows its SplitBuffers (src/topsy/split_buffers.py:26-38, 78-116):
contiguous ranges [g*N/G, (g+1)*N/G) and (start, len) blocks intersected with each shard.

The collective lives inside libtopsy_splat (tsp_comm_*); this module only derives shard ranges and
moves the 128-byte RCCL unique id between ranks (any out-of-band channel works: torch.distributed,
MPI, a file)."""
import numpy as np


def shard_bounds(n_total, world_size):
    """Start index of every shard plus the end: shard g owns [b[g], b[g+1])."""
    return np.array([(g * int(n_total)) // world_size for g in range(world_size + 1)], dtype=np.int64)


def shard_range(n_total, rank, world_size):
    b = shard_bounds(n_total, world_size)
    return int(b[rank]), int(b[rank + 1] - b[rank])


def intersect_ranges(starts, lens, shard_start, shard_len):
    """Clip global (start, len) blocks to one shard and re-base them to shard-local indices
    (what global_to_split_monotonic does per physical buffer in the reference)."""
    starts = np.asarray(starts, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    lo = np.maximum(starts, shard_start)
    hi = np.minimum(starts + lens, shard_start + shard_len)
    keep = hi > lo
    return (lo[keep] - shard_start), (hi[keep] - lo[keep])


def init_comm(context, rank, world_size, broadcast_bytes):
    """Create the RCCL communicator of `context`.  `broadcast_bytes(payload_or_None) -> bytes`
    must return rank 0's payload on every rank."""
    uid = context.comm_unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid)
    context.comm_init(world_size, rank, uid)


def torch_broadcaster(dist, src=0):
    def bcast(payload):
        box = [payload]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return bcast


class ShardedRenderer:
    """Frame driver for N ranks: every rank renders its index-range shard of each block, the partial
    images are summed on `root`, and the progressive mass scale stays global (N_total / N_drawn)."""

    def __init__(self, context, n_total, rank, world_size):
        self.context = context
        self.n_total = int(n_total)
        self.rank, self.world_size = rank, world_size
        self.shard_start, self.shard_len = shard_range(n_total, rank, world_size)

    def render_block(self, matrix, scale_factor, starts, lens, clear, mode=0, flags=0):
        """Render this rank's share of one block.  starts = lens = None means the whole snapshot (as in tsp_render)."""
        if starts is None and lens is None:
            starts, lens = [0], [self.n_total]
        s, l = intersect_ranges(starts, lens, self.shard_start, self.shard_len)
        return self.context.render(matrix, scale_factor, s, l, clear=clear, mode=mode, flags=flags)

    def reduce(self, root=0):
        """Sum the partial images on `root` (every rank when root < 0).  Exactly once per frame, after the frame's
        last render_block: the library refuses a second reduce of the same frame (it would double-count), and any
        later render_block -- a REFINE block included -- rebuilds the local partial image, which must then be
        reduced again before presentation (include/topsy_splat.h, tsp_comm_reduce_image)."""
        if self.world_size > 1:
            return self.context.comm_reduce_image(root)
        return 0.0
