"""Resident particle data of one visualizer: the HBM-side counterpart of the reference's
ParticleBuffers (src/topsy/particle_buffers.py).

The reference packs AoS vertex buffers (pos_smooth f32x4, mass_and_quantity f32x3, rgb f32x3;
:84-118) split into <= 2^27-particle physical buffers and rewrites indirect draw records per
block (:76-82).  Here the attributes are uploaded once as SoA float32 arrays through the C-ABI
(tsp_upload_*), there is no buffer splitting (one HIP allocation per attribute), and a block's
(starts, lens) ranges are handed straight to tsp_render.
"""
import logging

import numpy as np

from . import _native, config, kernel_lut

logger = logging.getLogger(__name__)
_UNSET = object()


class ParticleBuffers:
    def __init__(self, loader, resolution, device_id=0, max_draw_calls_per_buffer=1, device_ids=None,
                 shard_assignment=None):
        self._loader = loader
        self.quantity_name = None
        self._quantity_on_device = _UNSET
        self._have_rgb = False
        self._last_ranges = (None, None)
        self._max_draw_calls_per_buffer = max_draw_calls_per_buffer
        self.block_boundaries = None          # stratum offsets when the library reordered the particles
        self.device_cells = None              # ... and the cells of that ordering (view culling)
        # one 4-channel-capable context serves SPH, DepthSPH and RGBSPH (the active channel count
        # follows the render mode); several devices: one context each behind the same interface, the particles
        # sharded by index range and the image summed once per frame (multigpu.py)
        if device_ids is not None and len(device_ids) > 1:
            from . import multigpu
            assignment = shard_assignment or config.MULTI_GPU_SHARD_ASSIGNMENT
            if assignment == "auto":
                assignment = "interleaved" if hasattr(loader, "_cell_layout") else "contiguous"
            self.context = multigpu.MultiGpuContext(resolution, 4, device_ids, assignment=assignment,
                                                    interleave_block=config.MULTI_GPU_INTERLEAVE_BLOCK)
        else:
            self.context = _native.Context(resolution, 4, device_id if not device_ids else device_ids[0])
        self.context.set_kernel_mips(kernel_lut.kernel_mips())
        self._upload_geometry()

    def _upload_geometry(self):
        ld = self._loader
        if getattr(ld, "on_device", False):
            self.context.generate_synthetic(ld.n_total, ld.first, ld.count, ld.seed, ld.h_cap, with_quantity=True,
                                            with_rgb=True)
            self._have_rgb = True
            if ld.spatial_order:
                self.context.reorder_spatial(self._num_strata(ld.count), ld.seed)
                self._note_ordering()
            return
        logger.info("Uploading position+smoothing+mass arrays")
        ps = ld.get_pos_smooth()
        self.context.upload_particles(ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3], ld.get_mass())
        # Load-time spatial ordering (the counterpart of the reference's cell sort at load, loader.py:88-97).
        # A loader WITH a cell layout hands out per-cell index ranges in its own order, which is already
        # cell-coherent and must be kept; otherwise the library may reorder: strata keep index prefixes
        # unbiased for the plain RenderProgression, and later quantity/rgb uploads are permuted by the library.
        if not hasattr(ld, "_cell_layout") and len(ld) > 1:
            self.context.reorder_spatial(self._num_strata(len(ld)), 1337)
            self._note_ordering()

    def _note_ordering(self):
        from . import cell_layout
        self.block_boundaries = self.context.strata_offsets()
        self.device_cells = cell_layout.StratifiedCells.from_context(self.context)

    @staticmethod
    def _num_strata(n):
        """Strata of the load-time ordering: at least SPATIAL_ORDER_STRATA, more for large snapshots so that one
        stratum (the smallest unbiased block) stays below MAX_PARTICLES_PER_STRATUM (measured at 1.25e8 particles:
        128 strata cost 0.1 ms of a 40 ms frame and 0.7 ms of the 5 ms h-capped frame, 400 strata 0.5 ms)."""
        return int(min(max(config.SPATIAL_ORDER_STRATA, -(-n // config.MAX_PARTICLES_PER_STRATUM)), config.SPATIAL_ORDER_MAX_STRATA))

    def __len__(self):
        return len(self._loader)

    # -- per-mode attribute residency (get_mass_and_quantity_buffers / get_rgb_buffers) --------
    def ensure_quantity(self):
        """Make the resident quantity channel match `quantity_name` (None = density render, q = 0)."""
        if self._quantity_on_device is not _UNSET and self._quantity_on_device == self.quantity_name:
            return
        if getattr(self._loader, "on_device", False):
            # the generated quantity stays resident; a density render just stops reading it
            if self.quantity_name not in (None, "test-quantity"):
                raise KeyError(self.quantity_name)
            self.context.set_option("use_quantity", 0 if self.quantity_name is None else 1)
        elif self.quantity_name is None:
            self.context.upload_quantity(None)
        else:
            self.context.upload_quantity(np.asarray(self._loader.get_named_quantity(self.quantity_name), dtype=np.float32))
        self._quantity_on_device = self.quantity_name

    def ensure_rgb(self):
        if not self._have_rgb:
            bands = getattr(self._loader, "get_band_magnitudes", lambda: None)()
            if bands is not None:
                logger.info("Contracting band magnitudes to rgb on the device")
                self.context.upload_band_magnitudes(*bands)
                self._have_rgb = True
                return
            logger.info("Uploading rgb arrays")
            rgb = np.asarray(self._loader.get_rgb_masses(), dtype=np.float32)
            rgb = np.where(np.isnan(rgb), np.float32(0.0), rgb)      # reference loader.py:120
            self.context.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
            self._have_rgb = True

    # -- per-block ranges (update_particle_ranges, particle_buffers.py:76-82) -----------------
    def update_particle_ranges(self, particle_mins, particle_lens):
        self._last_ranges = (np.asarray(particle_mins, dtype=np.int64), np.asarray(particle_lens, dtype=np.int64))

    def current_ranges(self):
        return self._last_ranges
