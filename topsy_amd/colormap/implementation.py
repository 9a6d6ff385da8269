"""Colormap implementations on the HIP post-pass (kernels B / B').

Host-side mirror of reference src/topsy/colormap/implementation.py: the parameter dictionaries,
class selection by `accepts_parameters`, autorange, the folding of the progressive-render mass
scale into vmin/vmax, and the test entry `sph_raw_output_to_image` keep the reference's
semantics; the per-pixel work (colormap.wgsl) runs in tsp_colormap_* on the GPU.
"""
import logging

import numpy as np

from .. import config

logger = logging.getLogger(__name__)

_UINT8_FORMATS = ("rgba8unorm", "bgra8unorm")
_FLOAT_FORMATS = ("rgba16float", "rgba32float")


def percentile_from_order_statistics(fetch, first, n, q, transform=None):
    """np.percentile(sample, q) (method 'linear') where the float32 `sample` is known only through its
    order statistics: fetch(ranks) returns the values at the given ascending ranks, the sample is the n
    values starting at rank `first`, optionally passed through the monotone `transform` (np.log10).

    Follows numpy's own arithmetic step by step (numpy/lib/_function_base_impl.py): the divisor 100 takes
    the data dtype, so a Python-float q interpolates in float32 and an array q in float64; virtual index
    (n-1)*q; _lerp with its t >= 0.5 branch.  The device autorange therefore equals the host one bit for
    bit (tests/test_host_logic.py, tests/test_gpu_visualizer.py)."""
    scalar = np.ndim(q) == 0
    quantiles = np.true_divide(q, np.float32(100))          # Python float -> float32, float64 array stays
    virtual = np.asanyarray((n - 1) * quantiles)
    prev_f = np.floor(virtual)
    prev = np.atleast_1d(prev_f).astype(np.intp)
    nxt = np.minimum(prev + 1, n - 1)
    gamma = np.asanyarray(virtual - prev_f)
    vals = np.asarray(fetch(np.concatenate([first + prev, first + nxt])), dtype=np.float32)
    if transform is not None:
        vals = transform(vals)
    a, b = vals[:len(prev)], vals[len(prev):]
    if scalar:
        a, b = a[0], b[0]
    diff = np.subtract(b, a)
    out = np.asanyarray(np.add(a, diff * gamma))
    np.subtract(b, diff * (1 - gamma), out=out, where=gamma >= 0.5, casting="unsafe", dtype=type(out.dtype))
    return out[()] if scalar else out


def _lut_from_matplotlib(name, num_points):
    import matplotlib
    return matplotlib.colormaps[name](np.linspace(0.001, 0.999, num_points)).astype(np.float32)


class ColormapBase:
    _default_params = {}

    def __init__(self, device, input_texture, output_format, params):
        self._device = device                    # unused (kept for signature compatibility)
        self._input_texture = input_texture      # sph.RenderTarget
        self._output_format = output_format
        self._params = self._default_params | params

    @classmethod
    def accepts_parameters(cls, parameters):
        return False

    def update_parameters(self, parameters):
        if not self.accepts_parameters(self._params | parameters):
            raise ValueError(f"Colormap {self.__class__.__name__} does not accept parameter update: {parameters}")
        self._params.update(parameters)

    def get_parameter(self, name):
        return self._params.get(name, None)

    def get_parameters(self):
        return self._params.copy()

    def encode_render_pass(self, command_encoder, target_texture_view, bind_group=None):
        raise NotImplementedError

    def set_scaling(self, output_width, output_height, mass_scaling):
        raise NotImplementedError


class NoColormap(ColormapBase):
    """Placeholder before a type has been chosen (reference implementation.py:57-63)."""

    @classmethod
    def accepts_parameters(cls, parameters):
        return parameters.get("type", None) == "none"


class Colormap(ColormapBase):
    """1-D LUT map of density or of a density-weighted mean, log or linear."""
    input_channels = 2
    percentile_scaling = [1.0, 99.9]
    may_produce_weighted_average = True
    _default_params = {"colormap_name": "viridis", "vmin": 0.0, "vmax": 1.0, "log": True, "weighted_average": False}

    def __init__(self, device, input_texture, output_format, params):
        super().__init__(device, input_texture, output_format, params)
        self._lut_for = None
        self._shader_params = None
        self._setup_map_texture()

    @classmethod
    def accepts_parameters(cls, parameters):
        return parameters.get("type", None) == "density"

    def update_parameters(self, parameters):
        super().update_parameters(parameters)
        self._setup_map_texture()

    # -- LUT (reference :205-238) ---------------------------------------------------------------
    def _generate_mapping_rgba_f32(self, num_points):
        return _lut_from_matplotlib(self._params.get("colormap_name", config.DEFAULT_COLORMAP), num_points)

    def _setup_map_texture(self, num_points=config.COLORMAP_NUM_SAMPLES):
        name = self._params.get("colormap_name", config.DEFAULT_COLORMAP)
        if self._lut_for != (name, num_points):
            self._lut = np.ascontiguousarray(self._generate_mapping_rgba_f32(num_points))
            self._lut_for = (name, num_points)

    # -- content / range (reference :119-130, :381-425) ----------------------------------------
    def sph_raw_output_to_content(self, numpy_image):
        if self._params["weighted_average"]:
            with np.errstate(divide="ignore", invalid="ignore"):
                return numpy_image[..., 1] / numpy_image[..., 0]
        return numpy_image[..., 0]

    @classmethod
    def _finite_range(cls, values):
        good = values[np.isfinite(values)]
        return (np.min(good), np.max(good)) if len(good) else (np.nan, np.nan)

    def autorange_vmin_vmax(self, vals):
        self._autorange_using_values(self.sph_raw_output_to_content(vals).ravel())

    def _autorange_using_values(self, vals):
        with np.errstate(divide="ignore", invalid="ignore"):
            logged = np.log10(vals)
        lo_log, hi_log = self._finite_range(logged)
        lo, hi = self._finite_range(vals)
        if hi_log == lo_log:
            hi_log, lo_log = hi_log + 1.0, lo_log - 1.0
        if hi == lo:
            hi, lo = hi + 1.0, lo - 1.0
        use_log = not (vals < 0).any()
        sample = logged if use_log else vals
        sample = sample[np.isfinite(sample)]
        if len(sample) > 200:
            self._params["vmin"], self._params["vmax"] = np.percentile(sample, self.percentile_scaling)
        elif len(sample) > 2:
            self._params["vmin"], self._params["vmax"] = np.min(sample), np.max(sample)
        else:
            logger.warning("Problem setting vmin/vmax, perhaps there are no particles or something is wrong with them?")
            self._params["vmin"], self._params["vmax"] = 0.0, 1.0
        self.update_parameters({"ui_range_linear": (lo, hi), "ui_range_log": (lo_log, hi_log), "log": use_log})
        logger.info(f"Autoscale: log_scale={self._params['log']}, vmin={self._params['vmin']}, vmax={self._params['vmax']}")

    def autorange_on_device(self, mass_scale=1.0):
        """autorange_vmin_vmax(get_image()) without reading the image back: the device sorts the finite
        content values (tsp_content_sort) and the host needs ~8 of them."""
        ctx = self._input_texture.context
        kind = 1 if self._params["weighted_average"] else 0
        n_fin, n_nonpos = ctx.content_sort(kind, mass_scale)
        fetch = ctx.content_values
        n_pos = n_fin - n_nonpos
        with np.errstate(divide="ignore", invalid="ignore"):
            if n_fin:
                lo, hi = fetch([0, n_fin - 1])
                any_negative = bool(lo < 0)
            else:
                lo = hi = np.nan
                any_negative = False
            if n_pos:
                lo_log, hi_log = np.log10(fetch([n_nonpos, n_fin - 1]))
            else:
                lo_log = hi_log = np.nan
        if hi_log == lo_log:
            hi_log, lo_log = hi_log + 1.0, lo_log - 1.0
        if hi == lo:
            hi, lo = hi + 1.0, lo - 1.0
        use_log = not any_negative
        first, n, tf = (n_nonpos, n_pos, np.log10) if use_log else (0, n_fin, None)
        if n > 200:
            self._params["vmin"], self._params["vmax"] = percentile_from_order_statistics(fetch, first, n, self.percentile_scaling, tf)
        elif n > 2:
            ends = fetch([first, first + n - 1])
            self._params["vmin"], self._params["vmax"] = (tf(ends) if tf else ends)
        else:
            logger.warning("Problem setting vmin/vmax, perhaps there are no particles or something is wrong with them?")
            self._params["vmin"], self._params["vmax"] = 0.0, 1.0
        self.update_parameters({"ui_range_linear": (lo, hi), "ui_range_log": (lo_log, hi_log), "log": use_log})

    # -- shader parameters (reference :427-453) -------------------------------------------------
    def _update_parameter_buffer(self, width, height, mass_scale):
        """vmin/vmax as the kernel must see them when the image holds only 1/mass_scale of the mass."""
        p = {}
        d_vmin = self._params.get("density_vmin", 0.0)
        d_vmax = self._params.get("density_vmax", 1.0)
        p["density_vmin"] = np.float32((0.0 if d_vmin is None else d_vmin) - np.log10(mass_scale))
        p["density_vmax"] = np.float32((1.0 if d_vmax is None else d_vmax) - np.log10(mass_scale))
        if self.may_produce_weighted_average and self._params.get("weighted_average", False):
            mass_scale = 1.0       # a ratio of two channels does not depend on the sampling fraction
        vmin, vmax = np.float32(self._params["vmin"]), np.float32(self._params["vmax"])
        if self._params["log"]:
            vmin, vmax = vmin - np.log10(mass_scale), vmax - np.log10(mass_scale)
        else:
            vmin, vmax = vmin / mass_scale, vmax / mass_scale
        p["vmin"], p["vmax"] = np.float32(vmin), np.float32(vmax)
        p["window_aspect_ratio"] = np.float32(float(width) / height)
        p["gamma"] = np.float32(self._params.get("gamma", 1.0))
        self._shader_params = p
        return p

    def set_scaling(self, width, height, scaling):
        self._update_parameter_buffer(width, height, scaling)

    # -- the pass itself --------------------------------------------------------------------------
    def _output_dtype(self):
        if self._output_format in _UINT8_FORMATS:
            return np.uint8
        if self._output_format in _FLOAT_FORMATS:
            return np.float16 if self._output_format == "rgba16float" else np.float32
        raise ValueError(f"Unsupported output format: {self._output_format}")

    def _run_on_target(self, ctx):
        p = self._shader_params
        if self._output_dtype() != np.uint8:
            raise ValueError(f"Unsupported output format for a LUT colormap: {self._output_format}")
        return ctx.colormap_scalar(self._lut, p["vmin"], p["vmax"], self._params["log"],
                                   self._params.get("weighted_average", False))

    def _run_on_host_image(self, ctx, img):
        p = self._shader_params
        if self._output_dtype() != np.uint8:
            raise ValueError(f"Unsupported output format: {self._output_format}")
        return ctx.colormap_scalar_host(img, self._lut, p["vmin"], p["vmax"], self._params["log"],
                                        self._params.get("weighted_average", False))

    def encode_render_pass(self, command_encoder, target_texture_view, bind_group=None):
        """Apply the map to the resident render target.  `target_texture_view` is a host array
        (R, R, 4) to fill, or None to get a fresh one back (there is no wgpu encoder on this path)."""
        if self._shader_params is None:
            self.set_scaling(1, 1, 1.0)
        out = self._run_on_target(self._input_texture.context)
        if target_texture_view is not None:
            target_texture_view[...] = out
            return target_texture_view
        return out

    def sph_raw_output_to_image(self, numpy_image):
        """Arbitrary (H, W, C) float32 image -> colour image with the current parameters (S = 1)."""
        if len(numpy_image.shape) != 3:
            raise ValueError(f"Expected a 3D array, but got shape {numpy_image.shape}")
        if numpy_image.shape[2] != self.input_channels:
            raise ValueError(f"Expected the last dimension to have size {self.input_channels}, but got {numpy_image.shape[2]}")
        if numpy_image.dtype != np.float32:
            raise ValueError(f"Expected dtype to be np.float32, but got {numpy_image.dtype}")
        self._output_dtype()     # raises for unsupported formats
        self.set_scaling(numpy_image.shape[1], numpy_image.shape[0], 1.0)
        return self._run_on_host_image(self._input_texture.context, numpy_image)


class RGBColormap(Colormap):
    """Three log-scaled, gamma-mapped channels (stellar I/V/U bands)."""
    input_channels = 3
    max_percentile = 99.9
    dynamic_range = 3.0
    may_produce_weighted_average = False
    _sterrad_to_arcsec2 = 2.3504430539466191e-11
    _default_params = {"vmin": 0.0, "vmax": 1.0, "log": True, "gamma": 1.0}

    @classmethod
    def accepts_parameters(cls, parameters):
        parameters = cls._default_params | parameters
        return parameters.get("type", None) == "rgb" and (not parameters["hdr"]) and parameters["log"]

    def _setup_map_texture(self, num_points=config.COLORMAP_NUM_SAMPLES):
        self._lut = None     # no LUT in the tri-channel map

    # magnitudes per square arcsecond <-> log10 of the rendered surface brightness
    @classmethod
    def _log_output_to_mag_per_arcsec2(cls, val):
        return None if val is None else -2.5 * (val + np.log10(cls._sterrad_to_arcsec2) - 4)

    @classmethod
    def _mag_per_arcsec2_to_log_output(cls, val):
        return None if val is None else val / -2.5 + 4 - np.log10(cls._sterrad_to_arcsec2)

    def get_parameters(self):
        params = super().get_parameters()
        params["min_mag"] = self._log_output_to_mag_per_arcsec2(params["vmax"])
        params["max_mag"] = self._log_output_to_mag_per_arcsec2(params["vmin"])
        return params

    def get_parameter(self, name):
        if name == "min_mag":
            return self._log_output_to_mag_per_arcsec2(self.get_parameter("vmax"))
        if name == "max_mag":
            return self._log_output_to_mag_per_arcsec2(self.get_parameter("vmin"))
        return super().get_parameter(name)

    def update_parameters(self, parameters):
        if "min_mag" in parameters:
            parameters["vmax"] = self._mag_per_arcsec2_to_log_output(parameters["min_mag"])
        if "max_mag" in parameters:
            parameters["vmin"] = self._mag_per_arcsec2_to_log_output(parameters["max_mag"])
        ColormapBase.update_parameters(self, parameters)

    def autorange_vmin_vmax(self, vals):
        with np.errstate(divide="ignore", invalid="ignore"):
            vals = np.log10(vals.ravel())
        vals = vals[np.isfinite(vals)]
        if len(vals) > 200:
            self._params["vmax"] = np.percentile(vals, self.max_percentile)
        elif len(vals) > 2:
            self._params["vmax"] = np.max(vals)
        else:
            logger.warning("Problem setting vmin/vmax, perhaps there are no particles or something is wrong with them?")
            self._params["vmax"] = 1.0
        self._params["vmin"] = self._params["vmax"] - self.dynamic_range

    def sph_raw_output_to_content(self, numpy_image):
        return numpy_image[..., :3]

    def autorange_on_device(self, mass_scale=1.0):
        ctx = self._input_texture.context
        # kind 3 = every channel of the raw image: the reference passes sph.get_image() (R,R,4) and ravel()s
        # it, so the fragment-count channel takes part in the percentile (implementation.py:512-516)
        n_fin, n_nonpos = ctx.content_sort(3, mass_scale)
        n = n_fin - n_nonpos                     # log10 is finite exactly for the positive finite values
        if n > 200:
            self._params["vmax"] = percentile_from_order_statistics(ctx.content_values, n_nonpos, n, self.max_percentile, np.log10)
        elif n > 2:
            self._params["vmax"] = np.log10(ctx.content_values([n_fin - 1]))[0]
        else:
            logger.warning("Problem setting vmin/vmax, perhaps there are no particles or something is wrong with them?")
            self._params["vmax"] = 1.0
        self._params["vmin"] = self._params["vmax"] - self.dynamic_range

    def _run(self, fn):
        p = self._shader_params
        dt = self._output_dtype()
        if dt == np.uint8:
            return fn(p["vmin"], p["vmax"], p["gamma"], False)
        return fn(p["vmin"], p["vmax"], p["gamma"], True).astype(dt)

    def _run_on_target(self, ctx):
        return self._run(lambda a, b, g, f: ctx.colormap_rgb(a, b, g, as_float=f))

    def _run_on_host_image(self, ctx, img):
        return self._run(lambda a, b, g, f: ctx.colormap_rgb_host(img, a, b, g, as_float=f))


class RGBHDRColormap(RGBColormap):
    """Same map on a float16 canvas: values above 1 are kept (reference implementation.py:543-550)."""
    max_percentile = 99.0
    dynamic_range = 2.5

    @classmethod
    def accepts_parameters(cls, parameters):
        parameters = cls._default_params | parameters
        return parameters.get("type", None) == "rgb" and parameters["hdr"] and parameters["log"]


class BivariateColormap(Colormap):
    """2-D LUT: colour from the (weighted) value, brightness from log10 density (reference
    colormap/implementation.py:553-605, colormap.wgsl BIVARIATE branch :91-111)."""
    default_quantity_name = "rho"
    _default_params = Colormap._default_params | {"density_vmin": 0.0, "density_vmax": 1.0, "ui_range_density": (0.0, 1.0)}

    @classmethod
    def accepts_parameters(cls, parameters):
        return parameters.get("type", None) == "bivariate" and (not parameters.get("hdr", False))

    def _generate_mapping_rgba_f32(self, num_points):
        import matplotlib
        ramp = np.linspace(0.001, 0.999, num_points)
        rgba = np.ones((num_points, num_points, 4), dtype=np.float32)
        rgba[:, :, :] = matplotlib.colormaps[self._params["colormap_name"]](ramp)[:, np.newaxis, :]
        hsv = matplotlib.colors.rgb_to_hsv(rgba[..., :3])
        hsv[..., 2] = ramp[np.newaxis, :]                       # brightness follows the density axis
        fade = np.ones(num_points)
        fade[3 * num_points // 4:] = np.linspace(1.0, 0.0, num_points // 4)
        hsv[..., 1] *= fade[np.newaxis, :]                      # desaturate towards white at the bright end
        rgba[..., :3] = matplotlib.colors.hsv_to_rgb(hsv)
        return rgba

    def _setup_map_texture(self, num_points=config.COLORMAP_NUM_SAMPLES):
        name = self._params.get("colormap_name", config.DEFAULT_COLORMAP)
        if self._lut_for != (name, num_points):
            self._lut = np.ascontiguousarray(self._generate_mapping_rgba_f32(num_points))
            self._lut_for = (name, num_points)
            self._lut_resident = False

    def sph_raw_output_to_content(self, numpy_image):
        out = numpy_image.copy()
        if self._params["weighted_average"]:
            with np.errstate(divide="ignore", invalid="ignore"):
                out[..., 1] /= out[..., 0]
        else:
            out[..., 1] = out[..., 0]
        return out

    def autorange_vmin_vmax(self, vals):
        vals = self.sph_raw_output_to_content(vals)
        with np.errstate(divide="ignore", invalid="ignore"):
            den = np.log10(vals[..., 0].ravel())
        den = den[np.isfinite(den)]
        d_lo, d_hi = np.percentile(den, self.percentile_scaling)
        self.update_parameters({"density_vmin": d_lo, "density_vmax": d_hi, "ui_range_density": self._finite_range(den)})
        self._autorange_using_values(vals[..., 1])

    def autorange_on_device(self, mass_scale=1.0):
        ctx = self._input_texture.context
        n_fin, n_nonpos = ctx.content_sort(0, mass_scale)
        n = n_fin - n_nonpos
        d_lo, d_hi = percentile_from_order_statistics(ctx.content_values, n_nonpos, n, self.percentile_scaling, np.log10)
        ends = np.log10(ctx.content_values([n_nonpos, n_fin - 1]))
        self.update_parameters({"density_vmin": d_lo, "density_vmax": d_hi, "ui_range_density": (ends[0], ends[1])})
        super().autorange_on_device(mass_scale)     # the value axis: ch1/ch0 (weighted) or ch0

    def _ensure_lut_resident(self, ctx):
        if not getattr(self, "_lut_resident", False) or getattr(ctx, "_lut2d_owner", None) is not self:
            ctx.colormap_set_lut2d(self._lut)
            ctx._lut2d_owner = self
            self._lut_resident = True

    def _run_on_target(self, ctx):
        p = self._shader_params
        if self._output_dtype() != np.uint8:
            raise ValueError(f"Unsupported output format for a LUT colormap: {self._output_format}")
        self._ensure_lut_resident(ctx)
        return ctx.colormap_bivariate(p["vmin"], p["vmax"], p["density_vmin"], p["density_vmax"], self._params["log"],
                                      self._params.get("weighted_average", False))

    def _run_on_host_image(self, ctx, img):
        p = self._shader_params
        if self._output_dtype() != np.uint8:
            raise ValueError(f"Unsupported output format: {self._output_format}")
        self._ensure_lut_resident(ctx)
        return ctx.colormap_bivariate_host(img, p["vmin"], p["vmax"], p["density_vmin"], p["density_vmax"], self._params["log"],
                                           self._params.get("weighted_average", False))
