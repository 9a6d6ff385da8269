"""ColormapHolder: owns the currently active colormap implementation and swaps it when a parameter
update asks for a different kind (mirror of reference src/topsy/colormap/__init__.py:12-159)."""
import numpy as np

from .. import config
from .implementation import ColormapBase, NoColormap, Colormap, RGBColormap, RGBHDRColormap, BivariateColormap


def _all_subclasses(base):
    for sub in base.__subclasses__():
        yield sub
        yield from _all_subclasses(sub)


class ColormapHolder:
    def __init__(self, device, input_texture, output_format):
        self._device = device
        self._input_texture = input_texture
        self._output_format = output_format
        self._impl = self.instance_from_parameters(
            {"colormap_name": config.DEFAULT_COLORMAP, "vmin": None, "vmax": None, "log": False, "type": "none"},
            device, input_texture, output_format)

    def _check_valid(self):
        if self._impl is None or isinstance(self._impl, NoColormap):
            raise ValueError("ColormapHolder is not fully initialized")

    @classmethod
    def _class_from_parameters(cls, parameters):
        for candidate in _all_subclasses(ColormapBase):
            if candidate.accepts_parameters(parameters):
                return candidate
        return None

    @classmethod
    def instance_from_parameters(cls, parameters, device, input_texture, output_format):
        chosen = cls._class_from_parameters(parameters)
        if chosen is None:
            raise ValueError(f"No colormap class found for parameters: {parameters}")
        return chosen(device, input_texture, output_format, parameters)

    def update_parameters(self, parameters):
        """Returns True when a new implementation object had to be created."""
        merged = self.get_parameters() | parameters
        if self._impl is None and self._class_from_parameters(merged) is None:
            return None
        if self._impl is None or not self._impl.accepts_parameters(merged):
            self._impl = self.instance_from_parameters(merged, self._device, self._input_texture, self._output_format)
            return True
        self._impl.update_parameters(parameters)
        return False

    def get_parameter(self, name):
        return self._impl.get_parameter(name)

    def get_parameters(self):
        return self._impl.get_parameters()

    def autorange(self, sph_render_output: np.ndarray):
        self._check_valid()
        self._impl.autorange_vmin_vmax(sph_render_output)

    def autorange_on_device(self, mass_scaling=1.0):
        """Same result as autorange(sph.get_image()) computed from device-side order statistics."""
        self._check_valid()
        self._impl.autorange_on_device(mass_scaling)

    def encode_render_pass(self, command_encoder, target_texture_view):
        self._check_valid()
        return self._impl.encode_render_pass(command_encoder, target_texture_view)

    def set_scaling(self, width, height, mass_scaling):
        self._check_valid()
        self._impl.set_scaling(width, height, mass_scaling)

    def sph_raw_output_to_image(self, sph_raw_output):
        self._check_valid()
        return self._impl.sph_raw_output_to_image(sph_raw_output)

    def sph_raw_output_to_content(self, sph_raw_output):
        self._check_valid()
        return self._impl.sph_raw_output_to_content(sph_raw_output)

    def colormap_kind(self):
        """'bivariate' | 'rgb' | 'scalar': the family of the active implementation (what the reference's
        make_ui_controller dispatches on, src/topsy/colormap/__init__.py:131-147; 'surface' is out of scope here)."""
        self._check_valid()
        if isinstance(self._impl, BivariateColormap):
            return "bivariate"
        if isinstance(self._impl, RGBColormap):
            return "rgb"
        return "scalar"

    def make_ui_controller(self, visualizer, refresh_ui_callback=None, controller_classes=None):
        """The UI controller for the active colormap (reference src/topsy/colormap/__init__.py:131-147).

        The controllers themselves are UI (reference src/topsy/colormap/ui.py: Qt / Jupyter widget descriptions) and out of
        this backend's scope, so they are taken from the caller: `controller_classes` maps 'bivariate' / 'rgb' / 'scalar' to
        a GenericController subclass; when omitted, the reference's own classes are used if topsy is importable (the
        maintainer's integration).  They only need what this holder offers: get_parameter(s),
        update_parameters and [] access."""
        kind = self.colormap_kind()
        if controller_classes is None:
            try:
                from topsy.colormap import ui as ref_ui
            except ImportError as e:
                raise NotImplementedError(
                    "make_ui_controller needs the UI controller classes of topsy's colormap/ui.py (BivariateColorMapController, "
                    "RGBMapController, ColorMapController); pass them as controller_classes={'bivariate': ..., 'rgb': ..., "
                    "'scalar': ...} or make the `topsy` package importable") from e
            controller_classes = {"bivariate": ref_ui.BivariateColorMapController, "rgb": ref_ui.RGBMapController,
                                  "scalar": ref_ui.ColorMapController}
        return controller_classes[kind](visualizer, refresh_ui_callback)

    def __getitem__(self, key):
        return self.get_parameter(key)

    def __setitem__(self, key, value):
        self.update_parameters({key: value})
