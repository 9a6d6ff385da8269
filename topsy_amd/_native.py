"""ctypes binding of libtopsy_splat.so -- the only route from the Python host layer to the GPU.

There is deliberately NO fallback: if the HIP library is missing or no GPU is present the
product raises `BackendUnavailable` (the CPU oracle under oracle/ is test infrastructure and is
never imported from here).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOPSY_SPLAT_LIB") or os.path.join(_HERE, "libtopsy_splat.so")    # TOPSY_SPLAT_LIB: an alternative build (A/B measurements)

MODE_WEIGHTED, MODE_DEPTH, MODE_RGB = 0, 1, 2
PIPE_DEFAULT, PIPE_GENERIC = 0, 1
SAMPLE_BILINEAR_MIP0, SAMPLE_BILINEAR_MIP = 0x10, 0x20      # diagnostic sampling rules (generic kernel)
UNIQUE_ID_BYTES = 128
ABI_VERSION = 105            # include/topsy_splat.h: tsp_version()


class BackendUnavailable(RuntimeError):
    """libtopsy_splat.so could not be loaded, or no MI355X-class GPU is visible."""


class BackendError(RuntimeError):
    """A C-ABI call returned a negative status."""


class Stats(ctypes.Structure):
    _fields_ = [("n_particles", ctypes.c_int64), ("n_small", ctypes.c_int64), ("n_mid", ctypes.c_int64),
                ("n_huge", ctypes.c_int64), ("n_culled", ctypes.c_int64), ("n_fragments", ctypes.c_int64),
                ("ms_stream", ctypes.c_double), ("ms_mid", ctypes.c_double), ("ms_huge", ctypes.c_double),
                ("ms_total", ctypes.c_double), ("ms_mega", ctypes.c_double), ("n_mega", ctypes.c_int64),
                ("n_fragments_stream", ctypes.c_int64), ("n_fragments_mid", ctypes.c_int64),
                ("n_fragments_huge", ctypes.c_int64), ("n_fragments_mega", ctypes.c_int64), ("n_chunk_culled", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_fp = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_ctx = ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/topsy_splat.h
SIGNATURES = {
    "tsp_last_error": (ctypes.c_char_p, []),
    "tsp_version": (ctypes.c_int, []),
    "tsp_stats_size": (ctypes.c_int, []),
    "tsp_device_count": (ctypes.c_int, []),
    "tsp_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_ctx)]),
    "tsp_destroy": (None, [_ctx]),
    "tsp_set_kernel_mips": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_int]),
    "tsp_upload_particles": (ctypes.c_int, [_ctx, ctypes.c_int64, _fp, _fp, _fp, _fp, _fp]),
    "tsp_upload_quantity": (ctypes.c_int, [_ctx, _fp]),
    "tsp_upload_rgb": (ctypes.c_int, [_ctx, _fp, _fp, _fp]),
    "tsp_upload_band_magnitudes": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "tsp_generate_synthetic": (ctypes.c_int, [_ctx, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                              ctypes.c_float, ctypes.c_int, ctypes.c_int]),
    "tsp_reorder_spatial": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_uint64, _i64p]),
    "tsp_get_strata_offsets": (ctypes.c_int, [_ctx, _i64p, ctypes.c_int]),
    "tsp_get_cell_layout": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _fp, _fp]),
    "tsp_get_cell_offsets": (ctypes.c_int64, [_ctx, _i64p, ctypes.c_int64]),
    "tsp_download_particles": (ctypes.c_int, [_ctx] + [_fp] * 9),
    "tsp_num_particles": (ctypes.c_int64, [_ctx]),
    "tsp_render": (ctypes.c_int, [_ctx, _fp, ctypes.c_float, _i64p, _i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "tsp_read_image": (ctypes.c_int, [_ctx, _fp]),
    "tsp_write_image": (ctypes.c_int, [_ctx, _fp]),
    "tsp_colormap_scalar": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                           ctypes.c_int, _u8p]),
    "tsp_colormap_rgb": (ctypes.c_int, [_ctx, ctypes.c_float, ctypes.c_float, ctypes.c_float, _u8p, _fp]),
    "tsp_colormap_set_lut2d": (ctypes.c_int, [_ctx, _fp, ctypes.c_int]),
    "tsp_colormap_bivariate": (ctypes.c_int, [_ctx, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                              ctypes.c_int, _u8p]),
    "tsp_colormap_bivariate_host": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _u8p]),
    "tsp_colormap_scalar_host": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _u8p]),
    "tsp_colormap_rgb_host": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float, _u8p, _fp]),
    "tsp_tile_periodic": (ctypes.c_int, [_ctx, ctypes.c_int, _fp, _fp]),
    "tsp_content_sort": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_float, _i64p, _i64p]),
    "tsp_content_values": (ctypes.c_int, [_ctx, _i64p, ctypes.c_int, _fp]),
    "tsp_get_stats": (ctypes.c_int, [_ctx, ctypes.POINTER(Stats)]),
    "tsp_set_option": (ctypes.c_int, [_ctx, ctypes.c_char_p, ctypes.c_int64]),
    "tsp_measure_read_bandwidth": (ctypes.c_int, [_ctx, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "tsp_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "tsp_comm_init": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    "tsp_comm_reduce_image": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "tsp_comm_destroy": (ctypes.c_int, [_ctx]),
    "tsp_set_reduced_image": (ctypes.c_int, [_ctx, _fp]),
    # several GPUs behind one handle (C clients; the Python layer's own driver is multigpu.MultiGpuContext)
    "tsp_group_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_ctx)]),
    "tsp_group_destroy": (None, [_ctx]),
    "tsp_group_size": (ctypes.c_int, [_ctx]),
    "tsp_group_context": (_ctx, [_ctx, ctypes.c_int]),
    "tsp_group_uses_rccl": (ctypes.c_int, [_ctx]),
    "tsp_group_set_kernel_mips": (ctypes.c_int, [_ctx, _fp, ctypes.c_int, ctypes.c_int]),
    "tsp_group_upload_particles": (ctypes.c_int, [_ctx, ctypes.c_int64, _fp, _fp, _fp, _fp, _fp]),
    "tsp_group_upload_quantity": (ctypes.c_int, [_ctx, _fp]),
    "tsp_group_upload_rgb": (ctypes.c_int, [_ctx, _fp, _fp, _fp]),
    "tsp_group_generate_synthetic": (ctypes.c_int, [_ctx, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                                    ctypes.c_float, ctypes.c_int, ctypes.c_int]),
    "tsp_group_reorder_spatial": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_uint64]),
    "tsp_group_num_particles": (ctypes.c_int64, [_ctx]),
    "tsp_group_set_option": (ctypes.c_int, [_ctx, ctypes.c_char_p, ctypes.c_int64]),
    "tsp_group_render": (ctypes.c_int, [_ctx, _fp, ctypes.c_float, _i64p, _i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "tsp_group_end_frame": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_double)]),
    "tsp_group_get_stats": (ctypes.c_int, [_ctx, ctypes.POINTER(Stats)]),
    "tsp_group_shard_range": (ctypes.c_int, [_ctx, ctypes.c_int, _i64p, _i64p]),
    "tsp_group_upload_band_magnitudes": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def load_library():
    """dlopen libtopsy_splat.so and declare every prototype. Raises BackendUnavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C topsy_amd/csrc` (needs hipcc, --offload-arch=gfx950)")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise BackendUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BackendUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.tsp_version() < ABI_VERSION or lib.tsp_stats_size() != ctypes.sizeof(Stats):
        raise BackendUnavailable(f"{LIB_PATH} has ABI version {lib.tsp_version()} / tsp_stats of {lib.tsp_stats_size()} bytes; "
                                 f"this binding needs version >= {ABI_VERSION} and {ctypes.sizeof(Stats)} bytes: rebuild the library")
    _lib = lib
    return lib


def _check(rc):
    if rc < 0:
        raise BackendError(f"libtopsy_splat error {rc}: {load_library().tsp_last_error().decode(errors='replace')}")
    return rc


def _f32(a, n=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"{name} has {a.size} elements, expected {n}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_fp)


def device_count():
    return load_library().tsp_device_count()


class Context:
    """One GPU renderer: R x R x C float32 target + resident SoA particles (include/topsy_splat.h)."""

    def __init__(self, resolution, n_channels, device_id=0):
        lib = load_library()
        ndev = lib.tsp_device_count()
        if ndev <= 0:
            raise BackendUnavailable("no HIP device visible: the topsy_amd render path needs an AMD GPU "
                                     "(there is no CPU fallback)")
        h = _ctx()
        _check(lib.tsp_create(int(device_id), int(resolution), int(n_channels), ctypes.byref(h)))
        self._h = h
        self._lib = lib
        self.resolution = int(resolution)
        self.n_channels = int(n_channels)          # capacity of the render target
        self.active_channels = int(n_channels)     # layout of the image currently held (2 or 4)
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tsp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data -----------------------------------------------------------------------------
    def set_kernel_mips(self, mips, n0=64, n_levels=4):
        mips = _f32(mips, name="kernel mips")
        _check(self._lib.tsp_set_kernel_mips(self._h, _ptr(mips), n0, n_levels))

    def upload_particles(self, x, y, z, h, mass=None):
        n = len(x)
        arrs = [_f32(a, n, nm) for a, nm in ((x, "x"), (y, "y"), (z, "z"), (h, "h"))]
        m = None if mass is None else _f32(mass, n, "mass")
        _check(self._lib.tsp_upload_particles(self._h, n, *[_ptr(a) for a in arrs], _ptr(m)))

    def upload_quantity(self, q):
        q = None if q is None else _f32(q, self.num_particles, "quantity")
        _check(self._lib.tsp_upload_quantity(self._h, _ptr(q)))

    def upload_rgb(self, r, g, b):
        n = self.num_particles
        r, g, b = _f32(r, n, "r"), _f32(g, n, "g"), _f32(b, n, "b")
        _check(self._lib.tsp_upload_rgb(self._h, _ptr(r), _ptr(g), _ptr(b)))

    def upload_band_magnitudes(self, mags, weights):
        """rgb channels from SSP band magnitudes on the device: channel_c = sum_b weights[c, b] * 10^(-0.4 * mags[b]).
        mags: (n_bands, n) float64; weights: (3, n_bands) float64 (reference loader.py:112-121: diag(0.5, 1, 1) over I, V, U)."""
        mags = np.ascontiguousarray(mags, dtype=np.float64)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if mags.ndim != 2 or mags.shape[1] != self.num_particles or weights.shape != (3, mags.shape[0]):
            raise ValueError("mags must be (n_bands, n_particles) and weights (3, n_bands)")
        dp = ctypes.POINTER(ctypes.c_double)
        _check(self._lib.tsp_upload_band_magnitudes(self._h, mags.shape[0], mags.ctypes.data_as(dp), weights.ctypes.data_as(dp)))

    def generate_synthetic(self, n_total, first=0, count=None, seed=1337, h_cap=0.0, with_quantity=False,
                           with_rgb=False):
        count = n_total - first if count is None else count
        _check(self._lib.tsp_generate_synthetic(self._h, n_total, first, count, seed, h_cap, int(with_quantity),
                                                int(with_rgb)))

    def reorder_spatial(self, n_strata=1, seed=1337, want_permutation=False):
        perm = np.empty(self.num_particles, dtype=np.int64) if want_permutation else None
        _check(self._lib.tsp_reorder_spatial(self._h, n_strata, seed,
                                             None if perm is None else perm.ctypes.data_as(_i64p)))
        return perm

    def strata_offsets(self):
        """First index of every stratum of the last reorder_spatial call, then n (empty if never reordered)."""
        buf = np.empty(4098, dtype=np.int64)
        k = int(self._lib.tsp_get_strata_offsets(self._h, buf.ctypes.data_as(_i64p), len(buf)))
        return buf[:k].copy()

    def cell_layout(self):
        """Cells of the library's load-time ordering (tsp_get_cell_layout / tsp_get_cell_offsets) or None when the
        particles were never reordered: dict(n_strata, cells_per_axis, box_lo (3,), cell_width (3,), offsets (n_strata *
        cells_per_axis^3 + 1,))."""
        ns, ca = ctypes.c_int(0), ctypes.c_int(0)
        lo, width = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        if self._lib.tsp_get_cell_layout(self._h, ctypes.byref(ns), ctypes.byref(ca), _ptr(lo), _ptr(width)) < 0:
            return None
        buf = np.empty(ns.value * ca.value ** 3 + 1, dtype=np.int64)
        k = int(self._lib.tsp_get_cell_offsets(self._h, buf.ctypes.data_as(_i64p), len(buf)))
        if k != len(buf):
            raise BackendError("tsp_get_cell_offsets returned an unexpected number of values")
        return {"n_strata": ns.value, "cells_per_axis": ca.value, "box_lo": lo.astype(np.float64),
                "cell_width": width.astype(np.float64), "offsets": buf}

    def cell_layouts(self):
        """[cell_layout()] or [] -- the list form that a multi-GPU context fills with one entry per shard."""
        lay = self.cell_layout()
        return [] if lay is None else [lay]

    def download_particles(self, names=("x", "y", "z", "h", "mass")):
        order = ("x", "y", "z", "h", "mass", "q", "r", "g", "b")
        n = self.num_particles
        out = {k: np.empty(n, dtype=np.float32) for k in names}
        _check(self._lib.tsp_download_particles(self._h, *[_ptr(out.get(k)) for k in order]))
        return out

    @property
    def num_particles(self):
        return int(self._lib.tsp_num_particles(self._h))

    # ---- render ---------------------------------------------------------------------------
    def render(self, matrix, scale_factor, starts=None, lens=None, clear=True, mode=MODE_WEIGHTED, flags=PIPE_DEFAULT):
        """One synchronous render block; returns GPU milliseconds (hipEvent pair)."""
        M = _f32(np.asarray(matrix, dtype=np.float32).reshape(16), 16, "matrix")
        ms = ctypes.c_double(0.0)
        if starts is None:
            sp = lp = None
            nr = 0
        else:
            s = np.ascontiguousarray(starts, dtype=np.int64)
            l = np.ascontiguousarray(lens, dtype=np.int64)
            if s.shape != l.shape or s.ndim != 1:
                raise ValueError("starts and lens must be 1-D arrays of equal length")
            if len(s) == 0:
                # an empty selection (e.g. view culling picked no cell of this block) draws nothing: hand the library one
                # explicit zero-length range so that it can never be read as "no ranges given = all particles"
                s = np.zeros(1, dtype=np.int64)
                l = np.zeros(1, dtype=np.int64)
            sp, lp, nr = s.ctypes.data_as(_i64p), l.ctypes.data_as(_i64p), len(s)
        _check(self._lib.tsp_render(self._h, _ptr(M), float(scale_factor), sp, lp, nr, int(bool(clear)), int(mode),
                                    int(flags), ctypes.byref(ms)))
        self.active_channels = 4 if mode == MODE_RGB else 2
        return ms.value

    def read_image(self):
        out = np.empty((self.resolution, self.resolution, self.active_channels), dtype=np.float32)
        _check(self._lib.tsp_read_image(self._h, _ptr(out)))
        return out

    def write_image(self, img):
        """Overwrite the render target; the last axis (2 or 4) selects the active layout."""
        img = np.ascontiguousarray(img, dtype=np.float32)
        if img.shape[:2] != (self.resolution, self.resolution) or img.shape[2] not in (2, 4) or img.shape[2] > self.n_channels:
            raise ValueError(f"image shape {img.shape} does not fit the render target")
        self.set_option("active_channels", img.shape[2])
        self.active_channels = img.shape[2]
        _check(self._lib.tsp_write_image(self._h, _ptr(img)))

    def tile_periodic(self, offsets_xy, weights):
        """Replace the render target by the weighted sum of shifted copies of itself (clip-space offsets)."""
        off = _f32(offsets_xy, name="offsets")
        w = _f32(weights, name="weights")
        if off.size != 2 * w.size:
            raise ValueError("offsets must have shape (n, 2) for n weights")
        _check(self._lib.tsp_tile_periodic(self._h, w.size, _ptr(off), _ptr(w)))

    def content_sort(self, kind, scale=1.0):
        """Sort the finite content values on the device; returns (n_finite, n_nonpositive)."""
        nf, nnp = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.tsp_content_sort(self._h, int(kind), float(np.float32(scale)), ctypes.byref(nf), ctypes.byref(nnp)))
        return nf.value, nnp.value

    def content_values(self, ranks):
        r = np.ascontiguousarray(ranks, dtype=np.int64)
        out = np.empty(len(r), dtype=np.float32)
        _check(self._lib.tsp_content_values(self._h, r.ctypes.data_as(_i64p), len(r), _ptr(out)))
        return out

    def stats(self):
        s = Stats()
        _check(self._lib.tsp_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def set_option(self, name, value):
        _check(self._lib.tsp_set_option(self._h, name.encode(), int(value)))

    def measure_read_bandwidth(self, nbytes=1 << 30, iters=10):
        g = ctypes.c_double(0.0)
        _check(self._lib.tsp_measure_read_bandwidth(self._h, nbytes, iters, ctypes.byref(g)))
        return g.value

    n_gpus = 1

    def end_frame(self, root=0):
        """Frame boundary of the render loop: nothing to exchange on one GPU (multigpu.MultiGpuContext sums the shards here)."""
        return 0.0

    # ---- colormap -------------------------------------------------------------------------
    def colormap_scalar(self, lut_rgba, vmin, vmax, log, weighted, out=None):
        """-> (R, R, 4) uint8.  `out`: a caller-owned array of that shape to write into (a display loop reuses one: a fresh
        4 MiB numpy array per frame costs its page faults inside the device-to-host copy, ~0.15 ms)."""
        lut = _f32(lut_rgba, name="lut")
        if out is None:
            out = np.empty((self.resolution, self.resolution, 4), dtype=np.uint8)
        elif out.shape != (self.resolution, self.resolution, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (R, R, 4) uint8 array")
        _check(self._lib.tsp_colormap_scalar(self._h, _ptr(lut), lut.size // 4, float(vmin), float(vmax), int(bool(log)),
                                             int(bool(weighted)), out.ctypes.data_as(_u8p)))
        return out

    def colormap_rgb(self, vmin, vmax, gamma, as_float=False):
        shape = (self.resolution, self.resolution, 4)
        if as_float:
            out = np.empty(shape, dtype=np.float32)
            _check(self._lib.tsp_colormap_rgb(self._h, float(vmin), float(vmax), float(gamma), None, _ptr(out)))
        else:
            out = np.empty(shape, dtype=np.uint8)
            _check(self._lib.tsp_colormap_rgb(self._h, float(vmin), float(vmax), float(gamma),
                                              out.ctypes.data_as(_u8p), None))
        return out

    def colormap_set_lut2d(self, lut_rgba):
        lut = np.ascontiguousarray(lut_rgba, dtype=np.float32)
        if lut.ndim != 3 or lut.shape[0] != lut.shape[1] or lut.shape[2] != 4:
            raise ValueError("2-D LUT must have shape (n, n, 4)")
        _check(self._lib.tsp_colormap_set_lut2d(self._h, _ptr(lut), lut.shape[0]))

    def colormap_bivariate(self, vmin, vmax, dvmin, dvmax, log, weighted):
        out = np.empty((self.resolution, self.resolution, 4), dtype=np.uint8)
        _check(self._lib.tsp_colormap_bivariate(self._h, float(vmin), float(vmax), float(dvmin), float(dvmax), int(bool(log)),
                                                int(bool(weighted)), out.ctypes.data_as(_u8p)))
        return out

    def colormap_bivariate_host(self, img, vmin, vmax, dvmin, dvmax, log, weighted):
        img = np.ascontiguousarray(img, dtype=np.float32)
        H, W, C = img.shape
        out = np.empty((H, W, 4), dtype=np.uint8)
        _check(self._lib.tsp_colormap_bivariate_host(self._h, _ptr(img), H, W, C, float(vmin), float(vmax), float(dvmin),
                                                     float(dvmax), int(bool(log)), int(bool(weighted)), out.ctypes.data_as(_u8p)))
        return out

    def colormap_scalar_host(self, img, lut_rgba, vmin, vmax, log, weighted):
        img = np.ascontiguousarray(img, dtype=np.float32)
        H, W, C = img.shape
        lut = _f32(lut_rgba, name="lut")
        out = np.empty((H, W, 4), dtype=np.uint8)
        _check(self._lib.tsp_colormap_scalar_host(self._h, _ptr(img), H, W, C, _ptr(lut), lut.size // 4, float(vmin),
                                                  float(vmax), int(bool(log)), int(bool(weighted)),
                                                  out.ctypes.data_as(_u8p)))
        return out

    def colormap_rgb_host(self, img, vmin, vmax, gamma, as_float=False):
        img = np.ascontiguousarray(img, dtype=np.float32)
        H, W, C = img.shape
        if as_float:
            out = np.empty((H, W, 4), dtype=np.float32)
            _check(self._lib.tsp_colormap_rgb_host(self._h, _ptr(img), H, W, C, float(vmin), float(vmax), float(gamma),
                                                   None, _ptr(out)))
        else:
            out = np.empty((H, W, 4), dtype=np.uint8)
            _check(self._lib.tsp_colormap_rgb_host(self._h, _ptr(img), H, W, C, float(vmin), float(vmax), float(gamma),
                                                   out.ctypes.data_as(_u8p), None))
        return out

    # ---- multi-GPU ------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _check(load_library().tsp_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, n_ranks, rank, unique_id):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        _check(self._lib.tsp_comm_init(self._h, n_ranks, rank, unique_id))

    def comm_destroy(self):
        if getattr(self, "_h", None):
            _check(self._lib.tsp_comm_destroy(self._h))

    def set_reduced_image(self, total):
        """Hand over a sum the caller formed (host collective): presentation image only, the accumulator stays this shard's."""
        total = np.ascontiguousarray(total, dtype=np.float32)
        if total.shape != (self.resolution, self.resolution, self.active_channels):
            raise ValueError(f"image shape {total.shape} does not fit the render target")
        _check(self._lib.tsp_set_reduced_image(self._h, _ptr(total)))

    def comm_reduce_image(self, root=0):
        ms = ctypes.c_double(0.0)
        _check(self._lib.tsp_comm_reduce_image(self._h, root, ctypes.byref(ms)))
        return ms.value


class Group:
    """ctypes view of the C-level device group (tsp_group_*, include/topsy_splat.h): what a C client uses to put several
    GPUs behind one handle.  The product's own multi-GPU driver is multigpu.MultiGpuContext; this class exists so that the
    tests exercise the C entry points.  `root` is a borrowed Context over tsp_group_context(group, 0)."""

    def __init__(self, resolution, n_channels, device_ids):
        self._lib = load_library()
        ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = _ctx()
        _check(self._lib.tsp_group_create(len(device_ids), ids, int(resolution), int(n_channels), ctypes.byref(h)))
        self._g = h
        self.resolution, self.n_channels = int(resolution), int(n_channels)
        self.root = self.member(0)

    def member(self, index):
        h = self._lib.tsp_group_context(self._g, int(index))
        if not h:
            raise IndexError(index)
        c = Context.__new__(Context)
        c._lib = self._lib
        c._h = _ctx(h)
        c.close = lambda: None           # borrowed: the group destroys its contexts
        c.resolution, c.n_channels, c.device_id, c.active_channels = self.resolution, self.n_channels, -1, 2
        return c

    def close(self):
        if getattr(self, "_g", None):
            self.root._h = None
            self._lib.tsp_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self._lib.tsp_group_size(self._g)

    @property
    def uses_rccl(self):
        return bool(self._lib.tsp_group_uses_rccl(self._g))

    @property
    def num_particles(self):
        return self._lib.tsp_group_num_particles(self._g)

    def set_kernel_mips(self, mips, n0=64, n_levels=4):
        m = _f32(mips, None, "mips")
        _check(self._lib.tsp_group_set_kernel_mips(self._g, _ptr(m), n0, n_levels))

    def upload_particles(self, x, y, z, h, mass=None):
        n = len(x)
        arrs = [_f32(a, n, nm) for a, nm in ((x, "x"), (y, "y"), (z, "z"), (h, "h"))]
        m = None if mass is None else _f32(mass, n, "mass")
        _check(self._lib.tsp_group_upload_particles(self._g, n, *[_ptr(a) for a in arrs], None if m is None else _ptr(m)))

    def upload_quantity(self, q):
        qa = None if q is None else _f32(q, self.num_particles, "q")
        _check(self._lib.tsp_group_upload_quantity(self._g, None if qa is None else _ptr(qa)))

    def upload_rgb(self, r, g, b):
        n = self.num_particles
        arrs = [_f32(a, n, nm) for a, nm in ((r, "r"), (g, "g"), (b, "b"))]
        _check(self._lib.tsp_group_upload_rgb(self._g, *[_ptr(a) for a in arrs]))

    def upload_band_magnitudes(self, mags, weights):
        mags = np.ascontiguousarray(mags, dtype=np.float64)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if mags.ndim != 2 or mags.shape[1] != self.num_particles or weights.shape != (3, mags.shape[0]):
            raise ValueError("mags must be (n_bands, N) and weights (3, n_bands)")
        dp = ctypes.POINTER(ctypes.c_double)
        _check(self._lib.tsp_group_upload_band_magnitudes(self._g, mags.shape[0], mags.ctypes.data_as(dp), weights.ctypes.data_as(dp)))

    def shard_range(self, index):
        first, count = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.tsp_group_shard_range(self._g, int(index), ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    def generate_synthetic(self, n_total, first=0, count=None, seed=1337, h_cap=0.0, with_quantity=False, with_rgb=False):
        count = n_total - first if count is None else count
        _check(self._lib.tsp_group_generate_synthetic(self._g, int(n_total), int(first), int(count), int(seed), float(h_cap),
                                                      int(with_quantity), int(with_rgb)))

    def reorder_spatial(self, n_strata=1, seed=1337):
        _check(self._lib.tsp_group_reorder_spatial(self._g, int(n_strata), int(seed)))

    def set_option(self, name, value):
        _check(self._lib.tsp_group_set_option(self._g, name.encode(), int(value)))

    def render(self, matrix, scale_factor, starts=None, lens=None, clear=True, mode=MODE_WEIGHTED, flags=PIPE_DEFAULT):
        M = _f32(np.asarray(matrix, dtype=np.float32).reshape(16), 16, "matrix")
        ms = ctypes.c_double(0.0)
        if starts is None:
            sp = lp = None
            nr = 0
        else:
            s = np.ascontiguousarray(starts, dtype=np.int64)
            l = np.ascontiguousarray(lens, dtype=np.int64)
            if len(s) == 0:
                s, l = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
            sp, lp, nr = s.ctypes.data_as(_i64p), l.ctypes.data_as(_i64p), len(s)
        _check(self._lib.tsp_group_render(self._g, _ptr(M), float(scale_factor), sp, lp, nr, int(bool(clear)), int(mode), int(flags),
                                          ctypes.byref(ms)))
        self.root.active_channels = 4 if mode == MODE_RGB else 2
        return ms.value

    def end_frame(self):
        ms = ctypes.c_double(0.0)
        _check(self._lib.tsp_group_end_frame(self._g, ctypes.byref(ms)))
        return ms.value

    def stats(self):
        st = Stats()
        _check(self._lib.tsp_group_get_stats(self._g, ctypes.byref(st)))
        return st.as_dict()
