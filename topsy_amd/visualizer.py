"""Headless Visualizer: the orchestrator surface of reference src/topsy/visualizer.py:31-602 for
the accelerated path -- data loader -> resident particle buffers -> SPH renderer -> colormap.

Windowing, overlays (colorbar / scalebar / status text / crosshairs), the recorder and view
synchronisation are out of scope; what the UI layers call on the orchestrator
-- rotate / scale / position_offset / quantity_name / render_mode / invalidate / draw /
colormap_autorange / get_sph_image / get_sph_presentation_image / get_depth_image / save -- is here
with the reference's semantics, so those layers can sit on top unchanged.
"""
import logging

import numpy as np

from . import colormap, config, loader, particle_buffers, periodic_sph, sph
from .drawreason import DrawReason

logger = logging.getLogger(__name__)

_VALID_MODES = {"univariate", "bivariate", "rgb", "rgb-hdr", "surface"}
_UNSUPPORTED_MODES = {
    "surface": "needs the depth-tested occlusion pass and bilateral filter (out of scope, SURVEY.md section 2)",
}


class VisualizerBase:
    device = None      # kept for signature compatibility; the GPU is owned by particle_buffers.context

    def __init__(self, data_loader_class=loader.TestDataLoader, data_loader_args=(), data_loader_kwargs={},
                 *, render_resolution=config.DEFAULT_RESOLUTION, periodic_tiling=False,
                 colormap_name=config.DEFAULT_COLORMAP, canvas_class=None, render_mode="univariate", device_id=0,
                 n_gpus=None, device_ids=None, shard_assignment=None):
        """n_gpus / device_ids: render on several GPUs of this node from this one process -- the particles are sharded
        by index range, every render block runs on all shards at once and the frame ends with one RCCL sum-reduce of the
        image onto the first device (topsy_amd/multigpu.py); everything else (colormap, autorange, exports) is unchanged.
        shard_assignment: "contiguous" | "interleaved" | "auto" (config.MULTI_GPU_SHARD_ASSIGNMENT)."""
        self._render_resolution = render_resolution
        self._sph = None
        self._colormap = None
        if device_ids is None and n_gpus is not None and n_gpus > 1:
            device_ids = list(range(device_id, device_id + n_gpus))
        self._device_id = device_id if not device_ids else device_ids[0]
        self._prevent_sph_rendering = False
        self._validate_render_mode(render_mode)
        self._render_mode = render_mode
        self.canvas_format = self._render_mode_to_canvas_format(render_mode)
        self.data_loader = data_loader_class(self.device, *data_loader_args, **data_loader_kwargs)
        self.particle_buffers = particle_buffers.ParticleBuffers(
            self.data_loader, render_resolution, self._device_id,
            self.data_loader.get_render_progression().get_max_particle_regions_per_block(), device_ids=device_ids,
            shard_assignment=shard_assignment)
        self.periodicity_scale = self.data_loader.get_periodicity_scale()
        self._periodic_tiling = periodic_tiling
        if periodic_tiling and not self.periodicity_scale:
            raise ValueError("periodic_tiling needs a data loader with a finite periodicity scale")
        self._pending_draw = None
        self._initialize_sph_and_colormap(colormap_name)

    # -- mode plumbing (reference visualizer.py:96-120, 170-186, 203-231) ----------------------
    def _get_sph_class_for_render_mode(self, render_mode):
        return sph.RGBSPH if render_mode in ("rgb", "rgb-hdr") else sph.SPH

    def _get_colormap_parameters_for_render_mode(self, render_mode):
        params = {"weighted_average": self.quantity_name is not None}
        if render_mode == "rgb":
            params.update({"type": "rgb", "hdr": False, "log": True})
        elif render_mode == "rgb-hdr":
            params.update({"type": "rgb", "hdr": True, "log": True})
        elif render_mode == "bivariate":
            params.update({"type": "bivariate"})
        else:
            params.update({"type": "density"})
        return params

    def _render_mode_to_canvas_format(self, render_mode):
        if render_mode is None:
            return None
        return "rgba16float" if render_mode.endswith("hdr") else "rgba8unorm"

    def _validate_render_mode(self, new_render_mode):
        if new_render_mode not in _VALID_MODES:
            raise ValueError(f"Invalid render_mode '{new_render_mode}'. Valid modes: {_VALID_MODES}")
        if new_render_mode in _UNSUPPORTED_MODES:
            raise ValueError(f"render_mode '{new_render_mode}' is not provided by the MI355X backend: "
                             f"{_UNSUPPORTED_MODES[new_render_mode]}")

    def _initialize_sph_and_colormap(self, colormap_name=None):
        previous = None if self._sph is None else (self._sph.rotation_matrix, self._sph.position_offset, self._sph.scale)
        if self._periodic_tiling:
            self._sph = periodic_sph.PeriodicSPH(self, self._render_resolution)
        else:
            sph_class = self._get_sph_class_for_render_mode(self._render_mode)
            logger.info(f"Using {sph_class.__name__} renderer for render mode '{self._render_mode}'")
            self._sph = sph_class(self, self._render_resolution)
        self.reset_view(*(previous or (None, None, None)))
        self._sph.invalidate()
        if colormap_name is None:
            colormap_name = self._colormap.get_parameter("colormap_name")
        self.render_texture = self._sph.get_output_texture()
        self._colormap = colormap.ColormapHolder(self.device, self.render_texture, self.canvas_format)
        self._colormap.update_parameters({"colormap_name": colormap_name})
        self._initialize_colormap()

    def _initialize_colormap(self):
        changed_type = self._colormap.update_parameters(self._get_colormap_parameters_for_render_mode(self._render_mode))
        params = self._colormap.get_parameters()
        if changed_type or params["vmin"] is None or params["vmax"] is None:
            logger.info("Autorange colormap parameters")
            self._autorange()

    def _update_render_mode(self, new_render_mode, revert_on_failure=True):
        self._validate_render_mode(new_render_mode)
        old = self._render_mode
        self._render_mode = new_render_mode
        try:
            self.canvas_format = self._render_mode_to_canvas_format(new_render_mode)
            self._initialize_sph_and_colormap()
        except Exception:
            if revert_on_failure:
                logger.error(f"Failed to update render mode to '{new_render_mode}'; reverting to '{old}'")
                self._update_render_mode(old, revert_on_failure=False)
            raise
        self.invalidate(DrawReason.CHANGE)

    # -- view state -------------------------------------------------------------------------------
    def invalidate(self, reason=DrawReason.CHANGE):
        """Mark the SPH image stale.  Without an event loop the redraw happens lazily on the next read."""
        self._sph.invalidate(reason)
        self._pending_draw = reason

    @staticmethod
    def _y_rotation_matrix(angle):     # rotates about x (name as in the reference, visualizer.py:347-351)
        c, s = np.cos(angle), np.sin(angle)
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    @staticmethod
    def _x_rotation_matrix(angle):     # rotates about y (visualizer.py:353-357)
        c, s = np.cos(angle), np.sin(angle)
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])

    def rotate(self, x_angle, y_angle):
        self.rotation_matrix = self._x_rotation_matrix(x_angle) @ self._y_rotation_matrix(y_angle) @ self.rotation_matrix

    def reset_view(self, rotation_matrix=None, position_offset=None, scale=None):
        self._sph.rotation_matrix = np.eye(3) if rotation_matrix is None else rotation_matrix
        self._sph.scale = self.data_loader.get_initial_view_width() if scale is None else scale
        self._sph.position_offset = -self.data_loader.get_initial_center() if position_offset is None else position_offset

    @property
    def colormap(self):
        return self._colormap

    @property
    def rotation_matrix(self):
        return self._sph.rotation_matrix

    @rotation_matrix.setter
    def rotation_matrix(self, value):
        self._sph.rotation_matrix = value
        self.invalidate()

    @property
    def position_offset(self):
        return self._sph.position_offset

    @position_offset.setter
    def position_offset(self, value):
        self._sph.position_offset = value
        self.invalidate()

    @property
    def scale(self):
        """Half-width of the view in simulation units."""
        return self._sph.scale

    @scale.setter
    def scale(self, value):
        self._sph.scale = value
        self.invalidate()

    @property
    def render_mode(self):
        return self._render_mode

    @render_mode.setter
    def render_mode(self, value):
        self._update_render_mode(value)

    @property
    def quantity_name(self):
        return self.particle_buffers.quantity_name

    @quantity_name.setter
    def quantity_name(self, value):
        if value == self.particle_buffers.quantity_name:
            return
        if value is not None:
            try:
                self.data_loader.get_named_quantity(value)
            except Exception as e:
                raise ValueError(f"Unable to get quantity named '{value}'") from e
        self.particle_buffers.quantity_name = value
        self.invalidate(DrawReason.CHANGE)
        self._colormap.update_parameters({"vmin": None, "vmax": None, "log": None})
        self._initialize_colormap()

    @property
    def averaging(self):
        return self.quantity_name is not None

    def _autorange(self):
        """colormap.autorange(sph.get_image()) (reference visualizer.py:313,336) without the read-back."""
        self._sph.ensure_rendered()
        self._colormap.autorange_on_device(self._sph.last_render_mass_scale)

    def colormap_autorange(self):
        self._autorange()
        self.invalidate(DrawReason.PRESENTATION_CHANGE)

    # -- drawing (reference visualizer.py:386-405) ---------------------------------------------
    def render_sph(self, draw_reason=DrawReason.CHANGE):
        self._sph.render(draw_reason)

    def draw(self, reason, target_texture_view=None):
        """One frame: SPH blocks, fold the sampling fraction into the colormap, colour.  Returns the
        (R, R, 4) presentation array (also written into `target_texture_view` when given)."""
        if not self._prevent_sph_rendering:
            self.render_sph(reason)
        res = self._render_resolution
        self._colormap.set_scaling(res, res, self._sph.last_render_mass_scale)
        out = self._colormap.encode_render_pass(None, target_texture_view)
        self._pending_draw = None
        if reason != DrawReason.EXPORT and not self._prevent_sph_rendering and self._sph.needs_refine():
            self.invalidate(DrawReason.REFINE)
        return out

    # -- exports (reference visualizer.py:452-570) ---------------------------------------------
    def get_sph_image(self):
        """Logical content of the SPH image (no colormap): density, weighted mean, or rgb."""
        return self._colormap.sph_raw_output_to_content(self._sph.get_image())

    def get_sph_presentation_image(self):
        """Colormapped export-quality image, (R, R, 4) uint8 RGBA (float16 for rgb-hdr)."""
        self.render_sph(DrawReason.EXPORT)
        res = self._render_resolution
        self._colormap.set_scaling(res, res, self._sph.last_render_mass_scale)
        return self._colormap.encode_render_pass(None, None)

    def get_depth_image(self):
        depth = self._sph.get_depth_image()
        # the depth pass went through the shared render target: the next frame must redraw the scene
        self.invalidate(DrawReason.CHANGE)
        return depth

    def save(self, filename="output.npy"):
        self._sph.render(DrawReason.EXPORT)
        if filename.endswith(".npy"):
            np.save(filename, self.get_sph_image())
            return
        import matplotlib.pyplot as plt
        extent = np.array([-1.0, 1.0, -1.0, 1.0]) * self.scale
        fig = plt.figure()
        plt.imshow(self.get_sph_presentation_image(), extent=extent)
        plt.xlabel("$x$/" + self.data_loader.get_position_units())
        plt.savefig(filename)
        plt.close(fig)

    def close(self):
        self.particle_buffers.context.close()


class Visualizer(VisualizerBase):
    pass
