"""Several GPUs behind ONE Visualizer: an in-process group of libtopsy_splat contexts with the interface of
`_native.Context`, so that SPH / ParticleBuffers / the colormap drive G GPUs exactly as they drive one.

The reference's frame path is one process and one device (src/topsy/visualizer.py:386-405, src/topsy/sph.py:306-332);
its only notion of "several buffers" is SplitBuffers (src/topsy/split_buffers.py:26-38, 78-116), which cuts the particle
set into contiguous index ranges and intersects every draw range with them.  This module does the same across GPUs
(SURVEY.md section 8e):

  * particles are sharded by contiguous global index range [g N / G, (g + 1) N / G) -- one tsp_context per device;
    for input that arrives SPATIALLY SORTED (a loader with its own cell layout, src/topsy/loader.py:88-97: index ranges are
    then spatial slabs, and one GPU would get the dense core while another gets the fragment-heavy outskirts) the
    assignment is block-cyclic instead (`assignment="interleaved"`, SURVEY.md section 8e's fallback): blocks of
    `interleave_block` consecutive particles are dealt to the shards in turn, each shard keeps its blocks in order, so a
    global (start, len) range still maps to ONE local range per shard (the per-shard index map is monotone -- the property
    global_to_split_monotonic relies on, split_buffers.py:78-116);
  * a render block's (start, len) ranges are intersected with every shard and the shards render CONCURRENTLY (one host
    thread per context; ctypes releases the GIL and every tsp_* call selects its own device);
  * the frame ends with ONE sum-reduce of the float32 image onto the first context (`end_frame`): RCCL over xGMI
    (tsp_comm_reduce_image), or -- when two contexts share a device, which RCCL refuses, i.e. on a single-GPU test box --
    a read-back / add / write-back through the host.  The host collective is for test boxes only: it keeps a FLOAT32 copy
    of the root's partial image and writes it back before a REFINE block, so the root's float64 accumulation is rounded to
    float32 once per frame there (RCCL leaves every accumulator untouched);
  * everything that looks at the finished image (read-back, colormap, autorange, periodic tiling) runs on the first
    context; the progressive mass scale N / N_drawn stays global because the render progression above is unchanged.

bench.py's one-process-per-GPU launch (torch.distributed.run) uses `distributed.ShardedRenderer` instead; both share
the shard arithmetic of `distributed`.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native, distributed

_COUNT_KEYS = ("n_particles", "n_small", "n_mid", "n_huge", "n_culled", "n_fragments", "n_mega",
               "n_fragments_stream", "n_fragments_mid", "n_fragments_huge", "n_fragments_mega", "n_chunk_culled")


class MultiGpuContext:
    def __init__(self, resolution, n_channels, device_ids, assignment="contiguous", interleave_block=4096):
        device_ids = [int(d) for d in device_ids]
        if len(device_ids) < 2:
            raise ValueError("MultiGpuContext needs at least two contexts (use _native.Context for one GPU)")
        if assignment not in ("contiguous", "interleaved"):
            raise ValueError(f"unknown shard assignment '{assignment}'")
        self.assignment = assignment
        self.interleave_block = int(interleave_block)
        self._cyclic = False                # the resident particles were dealt block-cyclically (interleaved upload)
        self._cyclic_ranges = False         # ... and global indices still refer to the caller's order (no library reordering)
        self._n_uploaded = 0
        self.device_ids = device_ids
        self.contexts = [_native.Context(resolution, n_channels, d) for d in device_ids]
        self.resolution = int(resolution)
        self.n_channels = int(n_channels)
        self.device_id = device_ids[0]
        self._pool = ThreadPoolExecutor(max_workers=len(device_ids), thread_name_prefix="tsp-gpu")
        self._bounds = np.zeros(len(device_ids) + 1, dtype=np.int64)
        self._needs_reduce = False          # the shards hold partial images that have not been summed onto the root yet
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

    # block-cyclic assignment: block b of `interleave_block` consecutive particles lives on shard b mod G, blocks in order
    def _cyclic_local(self, i, g):
        """Number of shard g's particles with global index < i (i may be an array): the monotone global -> local map."""
        i = np.asarray(i, dtype=np.int64)
        B, G = self.interleave_block, self.n_gpus
        nb, rem = i // B, i % B
        return (nb // G + ((nb % G) > g)) * B + np.where((nb % G) == g, rem, 0)

    def _cyclic_owned(self, g):
        """Global indices of shard g's particles in its local order."""
        B, G, n = self.interleave_block, self.n_gpus, self._n_uploaded
        blocks = np.arange(g, (n + B - 1) // B, G, dtype=np.int64)
        idx = (blocks[:, None] * B + np.arange(B, dtype=np.int64)[None, :]).ravel()
        return idx[idx < n]

    def _take(self, v, g, axis=0):
        """Shard g's part of a caller-ordered array."""
        if v is None:
            return None
        if self._cyclic:
            return np.take(v, self._cyclic_owned(g), axis=axis)
        a, ln = self._shard(g)
        return v[a:a + ln] if axis == 0 else v[:, a:a + ln]

    def _local_ranges(self, starts, lens, g):
        if self._cyclic_ranges:
            lo, hi = self._cyclic_local(starts, g), self._cyclic_local(starts + lens, g)
            keep = hi > lo
            return lo[keep], (hi - lo)[keep]
        a, ln = self._shard(g)
        return distributed.intersect_ranges(starts, lens, a, ln)

    def close(self):
        # the communicators go first, explicitly and rank by rank, in the calling thread (close() also runs from __del__ at
        # interpreter shutdown, when the worker pool can no longer take work)
        if getattr(self, "collective", None) == "rccl":
            for c in getattr(self, "contexts", []):
                try:
                    c.comm_destroy()
                except Exception:
                    pass
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

    # Whatever looks at the FINISHED frame runs on the first context after the reduce.  An explicit list: these calls read
    # the presented image (tile_periodic rewrites the presentation copy only: the float64 accumulator keeps the raw render,
    # include/topsy_splat.h).  Anything that would change one shard's accumulator alone (write_image, upload_*) is not
    # forwarded: it has a sharded implementation below or does not exist on several GPUs.
    _ROOT_OPERATIONS = ("read_image", "colormap_scalar", "colormap_rgb", "colormap_set_lut2d", "colormap_bivariate",
                        "colormap_bivariate_host", "colormap_scalar_host", "colormap_rgb_host", "content_sort",
                        "content_values", "tile_periodic", "measure_read_bandwidth")

    def __getattr__(self, name):
        if name in MultiGpuContext._ROOT_OPERATIONS:
            target = getattr(self.contexts[0], name)

            def on_root(*args, **kwargs):
                self.end_frame()
                return target(*args, **kwargs)
            return on_root
        raise AttributeError(f"'{name}' is not available on a multi-GPU context")

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
        self._n_uploaded = n
        self._cyclic = self._cyclic_ranges = self.assignment == "interleaved"
        if self._cyclic:
            sizes = [int(self._cyclic_local(n, g)) for g in range(self.n_gpus)]
            self._bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        else:
            self._bounds = distributed.shard_bounds(n, self.n_gpus)
        arrays = [np.asarray(v) if v is not None else None for v in (x, y, z, h, mass)]
        self._map(lambda g, c: c.upload_particles(*[self._take(v, g) for v in arrays]))

    def _upload_sliced(self, method, arrays, axis=0):
        self._map(lambda g, c: getattr(c, method)(*[self._take(v, g, axis) for v in arrays]))

    def upload_quantity(self, q):
        if q is None:
            self._map(lambda g, c: c.upload_quantity(None))
        else:
            self._upload_sliced("upload_quantity", [np.asarray(q)])

    def upload_rgb(self, r, g, b):
        self._upload_sliced("upload_rgb", [np.asarray(r), np.asarray(g), np.asarray(b)])

    def upload_band_magnitudes(self, mags, weights):
        mags = np.asarray(mags, dtype=np.float64)

        self._map(lambda g, c: c.upload_band_magnitudes(np.ascontiguousarray(self._take(mags, g, axis=1)), weights))

    def generate_synthetic(self, n_total, first=0, count=None, seed=1337, h_cap=0.0, with_quantity=False, with_rgb=False):
        count = n_total - first if count is None else count
        # the generator's index bijection makes every index range a uniform sample: contiguous shards are balanced
        self._cyclic = self._cyclic_ranges = False
        self._n_uploaded = count
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
        # from here on a global index means "position in the concatenation of the shards' new orders", whatever the upload
        # assignment was (later quantity / rgb uploads still arrive in the caller's order and are cut as at upload)
        cyclic_upload = self._cyclic
        self._cyclic_ranges = False
        if not want_permutation:
            return None
        if cyclic_upload:
            return np.concatenate([self._cyclic_owned(g)[np.asarray(p, dtype=np.int64)] for g, p in enumerate(perms)])
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
        if self._cyclic_ranges:          # back into the caller's order
            out = {k: np.empty(self._n_uploaded, dtype=np.float32) for k in names}
            for g, p in enumerate(parts):
                own = self._cyclic_owned(g)
                for k in names:
                    out[k][own] = p[k]
            return out
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
        def go(g, c):
            s, l = self._local_ranges(starts, lens, g)
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
            # host collective: the float32 partial images added in rank order (what ncclReduce(sum, float32) computes up to
            # association); the sum becomes the root's PRESENTATION image only -- its float64 accumulator keeps its own
            # partial sums, as after the in-place RCCL reduce, so REFINE blocks continue unrounded
            parts = self._map(lambda g, c: c.read_image())
            total = parts[0].copy()
            for p in parts[1:]:
                total += p
            self.contexts[0].set_reduced_image(total)
            self.last_reduce_ms = (time.perf_counter() - t) * 1e3
        return self.last_reduce_ms

    def per_shard_stats(self):
        """tsp_stats of every shard's last render block (load balance: compare their ms_total)."""
        return self._map(lambda g, c: c.stats())

    def stats(self):
        """Counters summed over the shards, times of the slowest shard."""
        per = self.per_shard_stats()
        out = {}
        for k in per[0]:
            vals = [p[k] for p in per]
            out[k] = sum(vals) if k in _COUNT_KEYS else max(vals)
        out["ms_reduce"] = self.last_reduce_ms
        return out
