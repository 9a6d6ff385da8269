"""SPH renderers: the host-side mirror of the reference's SPH / RGBSPH / DepthSPH classes
(src/topsy/sph.py) on top of the HIP splat kernels.

What is kept: constructor signature, the attribute surface the Visualizer reads and writes
(rotation_matrix, position_offset, scale, min_pixels, max_pixels, has_rendered,
last_render_mass_scale, last_render_fps, _render_progression, _render_resolution), and the
methods render / invalidate / needs_refine / get_image / get_depth_image / get_output_texture.
What is replaced: the wgpu pipeline (shader module, bind groups, vertex layouts, indirect
draws) by tsp_render calls on a resident SoA particle set.
"""
import copy
import logging

import numpy as np

from . import _native, config
from .drawreason import DrawReason
from .util import GpuFrameTimer

logger = logging.getLogger(__name__)


class RenderTarget:
    """Opaque handle to the device-resident float32 image (what get_output_texture() returns in
    place of a wgpu texture; understood by topsy_amd.colormap only)."""

    def __init__(self, context, n_channels, fmt):
        self.context = context
        self.n_channels = n_channels
        self.format = fmt
        self.width = self.height = context.resolution


class SPH:
    render_format = "rg32float"      # reference sph.py:23
    _nchannels_input = 2
    _nchannels_output = 2
    _output_dtype = np.float32
    _buffer_name = "mass_and_quantity"
    _mode = _native.MODE_WEIGHTED

    def __init__(self, visualizer, render_resolution, wrapping=False, share_render_progression=None):
        logger.info(f"Creating SPH renderer with resolution {render_resolution}")
        self._visualizer = visualizer
        self._render_resolution = render_resolution
        self._wrapping = wrapping
        self._context = visualizer.particle_buffers.context
        if self._context.resolution != render_resolution:
            raise ValueError("render resolution differs from the resident render target")
        self._render_texture = RenderTarget(self._context, self._nchannels_output, self.render_format)
        self._render_timer = GpuFrameTimer()
        if share_render_progression is not None:
            self._render_progression = share_render_progression
        else:
            self._render_progression = visualizer.data_loader.get_render_progression()
            boundaries = getattr(visualizer.particle_buffers, "block_boundaries", None)
            if boundaries is not None and hasattr(self._render_progression, "set_block_boundaries"):
                self._render_progression.set_block_boundaries(boundaries)
                # view culling on the library's own ordering (per-stratum Morton cell runs)
                self._render_progression.set_device_cells(getattr(visualizer.particle_buffers, "device_cells", None))
        self.scale = config.DEFAULT_SCALE
        self.min_pixels = 0.0
        self.max_pixels = np.inf
        self.rotation_matrix = np.eye(3)
        self.position_offset = np.zeros(3)
        self.has_rendered = False
        self.last_render_mass_scale = 1.0
        self.pipeline_flags = _native.PIPE_DEFAULT

    # -- camera (reference sph.py:268-299) ----------------------------------------------------
    def _get_transform_params(self):
        """Returns (M, scale_factor): row-major float32 4x4 with clip = M @ (x, y, z, 1)
        -- the transpose of the reference's uploaded `transform` -- and 1/scale."""
        translate = np.eye(4)
        translate[:3, 3] = self.position_offset
        # webgpu clip space has z in [0, 1]: squash z by 1/2 and centre it on 0.5
        to_clip = np.diag([1.0, 1.0, 0.5, 1.0])
        to_clip[2, 3] = 0.5
        rot_scale = np.zeros((4, 4))
        rot_scale[:3, :3] = np.asarray(self.rotation_matrix) / self.scale
        rot_scale[3, 3] = 1.0
        return (to_clip @ rot_scale @ translate).astype(np.float32), np.float32(1.0 / self.scale)

    # -- frame driver (reference sph.py:301-335) ----------------------------------------------
    def invalidate(self, draw_reason=DrawReason.CHANGE):
        if draw_reason not in (DrawReason.REFINE, DrawReason.PRESENTATION_CHANGE):
            self.has_rendered = False

    def _prepare_buffers(self):
        self._visualizer.particle_buffers.ensure_quantity()

    def _target_is_mine(self):
        """True while the shared device render target still holds THIS renderer's last frame.  SPH, RGBSPH and
        DepthSPH render into one resident target (the reference gives each its own texture, sph.py:56-63,
        443-446), so a depth query or another renderer's frame leaves an image that must not be refined,
        re-coloured or read as this renderer's."""
        return self.has_rendered and getattr(self._visualizer.particle_buffers, "last_renderer", None) is self

    def render(self, draw_reason=DrawReason.CHANGE):
        """One frame of render blocks.  Returns True when the render target was (re)drawn, False when the resident
        image is simply presented again (PRESENTATION_CHANGE with this renderer's frame still in the target)."""
        if draw_reason in (DrawReason.REFINE, DrawReason.PRESENTATION_CHANGE) and not self._target_is_mine():
            # nothing of ours to refine / re-present: start the frame again
            draw_reason = DrawReason.CHANGE
        if draw_reason == DrawReason.PRESENTATION_CHANGE:
            return False
        rp = self._render_progression
        if draw_reason != DrawReason.REFINE:
            rp.select_sphere(-np.asarray(self.position_offset), self.scale * 1.2)
            self._transform = self._get_transform_params()
        self._prepare_buffers()
        M, sf = self._transform
        buffers = self._visualizer.particle_buffers
        clear = rp.start_frame(draw_reason)
        n_blocks = 0
        while block := rp.get_block(self._render_timer.total_time_in_frame()):
            n_blocks += 1
            # a block is charged its wall-clock time (host work included), as the reference's TimeGpuOperation does
            with self._render_timer.block() as timed:
                buffers.update_particle_ranges(*block)
                starts, lens = buffers.current_ranges()
                timed.gpu_ms = self._context.render(M, sf, starts, lens, clear=clear, mode=self._mode, flags=self.pipeline_flags)
            rp.end_block(self._render_timer.total_time_in_frame())
            clear = False
        # several GPUs: the one exchange step of the frame, the sum-reduce of the partial images (no-op on one GPU)
        with self._render_timer.block() as timed:
            timed.gpu_ms = self._context.end_frame()
        self._render_timer.end_frame()
        self.last_render_mass_scale = rp.end_frame_get_scalefactor()
        mean = self._render_timer.running_mean_duration
        self.last_render_fps = 1.0 / mean if mean > 0 else float("inf")
        self.has_rendered = True
        self.last_render_blocks = n_blocks        # tsp_render calls of this frame (EXPORT: one on this backend)
        self._visualizer.particle_buffers.last_renderer = self
        return True

    def needs_refine(self):
        return self._render_progression.needs_refine()

    # -- read-back (reference sph.py:118-143) -------------------------------------------------
    def get_image(self):
        return self._get_image_unscaled() * self.last_render_mass_scale

    def ensure_rendered(self):
        """Trigger an EXPORT-quality render unless this renderer's image is resident (reference sph.py:127-131)."""
        if not self._target_is_mine():
            logger.info("Export-quality render has been triggered, because no valid render is resident.")
            self.render(DrawReason.EXPORT)

    def _get_image_unscaled(self):
        self.ensure_rendered()
        return self._context.read_image()

    def get_output_texture(self):
        return self._render_texture

    # -- depth (reference sph.py:90-116) ------------------------------------------------------
    def _get_depth_renderer(self):
        r = DepthSPH(self._visualizer, self._render_resolution, wrapping=self._wrapping,
                     share_render_progression=copy.copy(self._render_progression))
        r.rotation_matrix = self.rotation_matrix
        r.position_offset = self.position_offset
        r.scale = self.scale
        return r

    def get_depth_image(self, depth_renderer_reason=DrawReason.CHANGE):
        """Density-weighted line-of-sight position of the scene, in simulation units."""
        depth = self._get_depth_renderer()
        depth.render(depth_renderer_reason)
        image = depth.get_image()
        self.has_rendered = False          # the shared render target now holds the depth pass
        with np.errstate(divide="ignore", invalid="ignore"):
            depth_viewport = image[..., 1] / image[..., 0]
        return (depth_viewport - 0.5) * self.scale * 2.0


class BivariateSPH(SPH):
    """Density + mass-weighted mean pair: the same splat (reference sph.py:428-429)."""


class RGBSPH(SPH):
    render_format = "rgba32float"     # reference sph.py:432-439
    _buffer_name = "rgb"
    _nchannels_input = 3
    _nchannels_output = 4
    _mode = _native.MODE_RGB

    def _prepare_buffers(self):
        self._visualizer.particle_buffers.ensure_rgb()


class DepthSPH(SPH):
    """Channel 1 carries clip-space z instead of the quantity (reference sph.py:443-446)."""
    _mode = _native.MODE_DEPTH

    def _prepare_buffers(self):
        pass
