"""Several GPUs behind ONE Visualizer: an in-process group of libtopsy_splat contexts with the interface of
`_native.Context`, so that SPH / ParticleBuffers / the colormap drive G GPUs exactly as they drive one.

The reference's frame path is one process and one device (src/topsy/visualizer.py:386-405, src/topsy/sph.py:306-332);
its only notion of "several buffers" is SplitBuffers (src/topsy/split_buffers.py:26-38, 78-116), which cuts the particle
set into contiguous index ranges and intersects every draw range with them.  This module does the same across GPUs
(SURVEY.md section 8e):

  * particles are sharded by contiguous global index range [g N / G, (g + 1) N / G) -- one tsp_context per device;
  * a render block's (start, len) ranges are intersected with every shard and the shards render CONCURRENTLY (one host
    thread per context; ctypes releases the GIL and every tsp_* call selects its own device);
  * the frame ends with ONE sum-reduce of the float32 image onto the first context (`end_frame`): RCCL over xGMI
    (tsp_comm_reduce_image), or -- when two contexts share a device, which RCCL refuses, i.e. on a single-GPU test box --
    a read-back / add / write-back through the host;
  * everything that looks at the finished image (read-back, colormap, autorange, periodic tiling) runs on the first
    context; the progressive mass scale N / N_drawn stays global because the render progression above is unchanged.

bench.py's one-process-per-GPU launch (torch.distributed.run) uses `distributed.ShardedRenderer` instead; both share
the shard arithmetic of `distributed`.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native, distributed

_COUNT_KEYS = ("n_particles", "n_small", "n_mid", "n_huge", "n_culled", "n_fragments", "n_mega")


class MultiGpuContext:
    def __init__(self, resolution, n_channels, device_ids):
        device_ids = [int(d) for d in device_ids]
        if len(device_ids) < 2:
            raise ValueError("MultiGpuContext needs at least two contexts (use _native.Context for one GPU)")
        self.device_ids = device_ids
        self.contexts = [_native.Context(resolution, n_channels, d) for d in device_ids]
        self.resolution = int(resolution)
        self.n_channels = int(n_channels)
        self.device_id = device_ids[0]
        self._pool = ThreadPoolExecutor(max_workers=len(device_ids), thread_name_prefix="tsp-gpu")
        self._bounds = np.zeros(len(device_ids) + 1, dtype=np.int64)
        self._needs_reduce = False          # the shards hold partial images that have not been summed onto the root yet
        self._root_partial = None           # host collective: the root's own partial image while it holds the sum
        self.last_reduce_ms = 0.0
        # RCCL needs one device per rank; contexts that share a device (single-GPU boxes, tests) sum through the host
        self.collective = "rccl" if len(set(device_ids)) == len(device_ids) else "host"
        if self.collective == "rccl":
            uid = _native.Context.comm_unique_id()
            world = len(self.contexts)
            self._map(lambda g, c: c.comm_init(world, g, uid))      # ncclCommInitRank blocks until every rank has joined

    # ---- plumbing ------------------------------------------------------------------------------------------------------
    @property
    def root(self):
        return self.contexts[0]

    @property
    def n_gpus(self):
        return len(self.contexts)

    def _map(self, fn):
        futures = [self._pool.submit(fn, g, c) for g, c in enumerate(self.contexts)]
        return [f.result() for f in futures]

    def _shard(self, g):
        return int(self._bounds[g]), int(self._bounds[g + 1] - self._bounds[g])

    def close(self):
        for c in getattr(self, "contexts", []):
            c.close()
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name):
        # whatever is not sharded looks at the finished frame: it belongs to the root context, after the reduce
        if name.startswith("_") or name in ("contexts",):
            raise AttributeError(name)
        target = getattr(self.contexts[0], name)
        if callable(target):
            def on_root(*args, **kwargs):
                self.end_frame()
                return target(*args, **kwargs)
            return on_root
        return target

    @property
    def active_channels(self):
        return self.contexts[0].active_channels

    @active_channels.setter
    def active_channels(self, value):
        for c in self.contexts:
            c.active_channels = value

    # ---- data: every attribute is cut at the same shard bounds -------------------------------------------------------------
    def set_kernel_mips(self, mips, n0=64, n_levels=4):
        self._map(lambda g, c: c.set_kernel_mips(mips, n0, n_levels))

    def upload_particles(self, x, y, z, h, mass=None):
        n = len(x)
        self._bounds = distributed.shard_bounds(n, self.n_gpus)

        def up(g, c):
            a, ln = self._shard(g)
            s = slice(a, a + ln)
            c.upload_particles(x[s], y[s], z[s], h[s], None if mass is None else mass[s])
        self._map(up)

    def _upload_sliced(self, method, arrays, axis=0):
        def up(g, c):
            a, ln = self._shard(g)
            s = slice(a, a + ln)
            getattr(c, method)(*[(v if v is None else (v[s] if axis == 0 else v[:, s])) for v in arrays])
        self._map(up)

    def upload_quantity(self, q):
        if q is None:
            self._map(lambda g, c: c.upload_quantity(None))
        else:
            self._upload_sliced("upload_quantity", [np.asarray(q)])

    def upload_rgb(self, r, g, b):
        self._upload_sliced("upload_rgb", [np.asarray(r), np.asarray(g), np.asarray(b)])

    def upload_band_magnitudes(self, mags, weights):
        mags = np.asarray(mags, dtype=np.float64)

        def up(g, c):
            a, ln = self._shard(g)
            c.upload_band_magnitudes(mags[:, a:a + ln], weights)
        self._map(up)

    def generate_synthetic(self, n_total, first=0, count=None, seed=1337, h_cap=0.0, with_quantity=False, with_rgb=False):
        count = n_total - first if count is None else count
        self._bounds = distributed.shard_bounds(count, self.n_gpus)

        def gen(g, c):
            a, ln = self._shard(g)
            c.generate_synthetic(n_total, first + a, ln, seed, h_cap, with_quantity, with_rgb)
        self._map(gen)

    def reorder_spatial(self, n_strata=1, seed=1337, want_permutation=False):
        """Load-time ordering, shard by shard: every shard is cut into ceil(n_strata / G) strata, so the snapshot as a
        whole keeps about n_strata block boundaries (see strata_offsets)."""
        per_shard = max(1, math.ceil(n_strata / self.n_gpus))
        perms = self._map(lambda g, c: c.reorder_spatial(per_shard, seed, want_permutation) if self._shard(g)[1] > 0 else
                          (np.empty(0, dtype=np.int64) if want_permutation else None))
        if not want_permutation:
            return None
        return np.concatenate([np.asarray(p, dtype=np.int64) + self._bounds[g] for g, p in enumerate(perms)])

    def strata_offsets(self):
        """Global indices at which a render block may end: every shard's stratum offsets, shifted to its index range."""
        offs = self._map(lambda g, c: c.strata_offsets())
        if any(len(o) == 0 for g, o in enumerate(offs) if self._shard(g)[1] > 0):
            return np.empty(0, dtype=np.int64)
        parts = [o[:-1] + self._bounds[g] for g, o in enumerate(offs) if len(o)]
        return np.concatenate(parts + [self._bounds[-1:]]).astype(np.int64)

    def cell_layouts(self):
        """One cell grid per shard (each over its own bounding box), offsets shifted to global indices."""
        out = []
        for g, lays in enumerate(self._map(lambda g, c: c.cell_layouts() if self._shard(g)[1] > 0 else [])):
            for lay in lays:
                lay = dict(lay)
                lay["offsets"] = np.asarray(lay["offsets"], dtype=np.int64) + self._bounds[g]
                out.append(lay)
        return out

    def download_particles(self, names=("x", "y", "z", "h", "mass")):
        parts = self._map(lambda g, c: c.download_particles(names) if self._shard(g)[1] > 0 else
                          {k: np.empty(0, dtype=np.float32) for k in names})
        return {k: np.concatenate([p[k] for p in parts]) for k in names}

    @property
    def num_particles(self):
        return int(self._bounds[-1])

    def set_option(self, name, value):
        self._map(lambda g, c: c.set_option(name, value))

    # ---- render ------------------------------------------------------------------------------------------------------------
    def render(self, matrix, scale_factor, starts=None, lens=None, clear=True, mode=_native.MODE_WEIGHTED, flags=_native.PIPE_DEFAULT):
        """One render block on every shard at once; returns the slowest shard's GPU milliseconds.  A shard that the block's
        ranges do not touch still takes part (an empty selection): `clear` must reach every partial image."""
        if starts is None:
            starts, lens = [0], [self.num_particles]
        starts = np.asarray(starts, dtype=np.int64)
        lens = np.asarray(lens, dtype=np.int64)
        if self._root_partial is not None:
            # host collective: the root's image currently holds the SUM; give it back its own partial frame before it
            # accumulates (clear = False) -- a cleared frame needs nothing restored
            if not clear:
                self.contexts[0].write_image(self._root_partial)
            self._root_partial = None

        def go(g, c):
            a, ln = self._shard(g)
            s, l = distributed.intersect_ranges(starts, lens, a, ln)
            if len(s) == 0:
                s, l = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
            return c.render(matrix, scale_factor, s, l, clear=clear, mode=mode, flags=flags)
        ms = max(self._map(go))
        self._needs_reduce = True
        return ms

    def end_frame(self, root=0):
        """Sum the partial images onto the first context -- the ONE exchange step of a frame; a no-op when nothing was
        rendered since the last call.  Returns the milliseconds it took (GPU time for RCCL)."""
        if not self._needs_reduce:
            return 0.0
        self._needs_reduce = False
        if self.collective == "rccl":
            self.last_reduce_ms = max(self._map(lambda g, c: c.comm_reduce_image(0)))
        else:
            import time
            t = time.perf_counter()
            parts = self._map(lambda g, c: c.read_image())
            self._root_partial = parts[0]
            total = parts[0].astype(np.float64)
            for p in parts[1:]:
                total += p
            self.contexts[0].write_image(total.astype(np.float32))
            self.last_reduce_ms = (time.perf_counter() - t) * 1e3
        return self.last_reduce_ms

    def stats(self):
        """Counters summed over the shards, times of the slowest shard."""
        per = self._map(lambda g, c: c.stats())
        out = {}
        for k in per[0]:
            vals = [p[k] for p in per]
            out[k] = sum(vals) if k in _COUNT_KEYS else max(vals)
        out["ms_reduce"] = self.last_reduce_ms
        return out
