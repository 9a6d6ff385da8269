"""Data loaders: what the renderer asks of a particle source (host side).

The abstract surface and the synthetic TestDataLoader mirror reference src/topsy/loader.py
(:16-77 and :241-332).  pynbody I/O itself is out of scope for this backend: a pynbody snapshot is
fed through `ArrayDataLoader` (positions / smoothing / mass / named quantities as numpy arrays),
which is what the reference's PynbodyDataInMemory hands to its GPU buffers anyway.
"""
from abc import ABC, abstractmethod

import numpy as np

from . import cell_layout, config


class AbstractDataLoader(ABC):
    def __init__(self, device=None):
        self._device = device

    @abstractmethod
    def __len__(self): ...

    @abstractmethod
    def get_positions(self): ...

    @abstractmethod
    def get_smooth(self): ...

    @abstractmethod
    def get_mass(self): ...

    @abstractmethod
    def get_named_quantity(self, name): ...

    @abstractmethod
    def get_quantity_label(self, quantity_name): ...

    @abstractmethod
    def get_rgb_masses(self): ...

    @abstractmethod
    def get_position_units(self): ...

    def get_pos_smooth(self):
        """(N,4) float32: x, y, z, h -- the reference's vertex layout (loader.py:52-56)."""
        out = np.empty((len(self), 4), dtype=np.float32)
        out[:, :3] = self.get_positions()
        out[:, 3] = self.get_smooth()
        return out

    def get_periodicity_scale(self):
        return np.inf

    def get_render_progression(self):
        from . import progressive_render
        if hasattr(self, "_cell_layout"):
            return progressive_render.RenderProgressionWithCells(self._cell_layout, len(self))
        return progressive_render.RenderProgression(len(self))

    def get_initial_center(self):
        return np.zeros(3, dtype=np.float32)

    def get_initial_view_width(self):
        period = self.get_periodicity_scale()
        return period / 2 if period is not None else config.DEFAULT_SCALE

    def get_quantity_names(self):
        return []

    def get_filename(self):
        return "in-memory data"


class TestDataLoader(AbstractDataLoader):
    """Seeded 3-component Gaussian mixture; bit-for-bit the arrays of the reference's
    TestDataLoader (loader.py:241-332) -- pinned by tests/golden/testdata_n*.npz."""
    __test__ = False   # not a pytest class

    _WEIGHTS = (0.5, 0.4, 0.1)
    _MEANS = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [6.0, 10.0, 0.0]])
    _STDS = np.array([[20.0, 20.0, 20.0], [4.0, 0.2, 4.0], [2.0, 2.0, 3.0]])

    def __init__(self, device=None, n_particles=config.TEST_DATA_NUM_PARTICLES_DEFAULT, n_cells=10, seed=1337,
                 with_cells=False, periodic=False):
        super().__init__(device)
        self._n_particles = n_particles
        self._periodic = periodic
        self._gmm_pos = self._draw_positions(seed)
        self._gmm_den = self._number_density(self._gmm_pos)
        if with_cells:
            self._cell_layout, order = cell_layout.CellLayout.from_positions(
                self._gmm_pos, self._gmm_pos.min() - 1e-3, self._gmm_pos.max() + 1, n_cells)
            self._gmm_pos = self._gmm_pos[order]
            self._gmm_den = self._gmm_den[order]

    def __len__(self):
        return self._n_particles

    def _draw_positions(self, seed):
        np.random.seed(seed)
        n = self._n_particles
        pos = np.empty((n, 3), dtype=np.float32)
        if n == 1:
            pos[0] = self._MEANS[0]
        else:
            filled = 0
            for w, mu, sd in zip(self._WEIGHTS, self._MEANS, self._STDS):
                k = int(n * w)
                pos[filled:filled + k] = np.random.normal(size=(k, 3), scale=1.0).astype(np.float32) * sd[np.newaxis, :] + mu
                filled += k
            assert filled == n, "component sizes int(n*w) must add up to n (as in the reference)"
        return np.random.permutation(pos)

    def _number_density(self, pos):
        # note: exp(-r^2/sigma^2) without the 1/2, as in the reference (loader.py:269-271)
        den = np.zeros(len(pos))
        for w, mu, sd in zip(self._WEIGHTS, self._MEANS, self._STDS):
            den += w * np.exp(-np.sum((pos - mu) ** 2 / sd ** 2, axis=1)) / ((2 * np.pi) ** 1.5 * np.prod(sd))
        return den * self._n_particles

    def get_positions(self):
        return self._gmm_pos

    def get_smooth(self):
        return 2.0 / self._gmm_den ** 0.333333

    def get_mass(self):
        return np.repeat(np.float32(1e-8), self._n_particles)

    def get_named_quantity(self, name):
        if name != "test-quantity":
            raise KeyError("Unknown quantity name")
        p = self._gmm_pos
        return np.sin(p[:, 0]) * np.cos(p[:, 1]) * np.cos(p[:, 2]) * 1e-4

    def get_rgb_masses(self):
        p = self._gmm_pos
        rgb = np.empty((len(p), 3), dtype=np.float32)
        rgb[:, 0] = abs(np.sin(p[:, 0] / 10.0))
        rgb[:, 1] = abs(np.cos(p[:, 1] / 10.0))
        rgb[:, 2] = abs(np.cos(p[:, 2] / 10.0))
        return rgb

    def get_position_units(self):
        return "kpc"

    def get_quantity_names(self):
        return ["test-quantity"]

    def get_quantity_label(self, quantity_name):
        if quantity_name is None:
            return r"test density / $M_{\odot} / \mathrm{kpc}^2$"
        return "test quantity" if quantity_name == "test-quantity" else "unknown"

    def get_filename(self):
        return "test data"

    def get_periodicity_scale(self):
        return 100.0 if self._periodic else None


class ArrayDataLoader(AbstractDataLoader):
    """Particles given as numpy arrays (e.g. pulled from a pynbody snapshot by the caller:
    snap['pos'], snap['smooth'], snap['mass'], ...; reference PynbodyDataInMemory, loader.py:79-154)."""

    # reference PynbodyDataInMemory.get_rgb_masses (loader.py:115-121): (band, weight) per rgb channel
    RGB_BANDS = (("I", 0.5), ("V", 1.0), ("U", 1.0))

    def __init__(self, device=None, pos=None, smooth=None, mass=None, quantities=None, rgb=None,
                 units="kpc", periodicity_scale=None, with_cells=False, band_magnitudes=None):
        super().__init__(device)
        self._pos = np.asarray(pos, dtype=np.float32)
        self._smooth = np.asarray(smooth, dtype=np.float32)
        self._mass = np.asarray(mass, dtype=np.float32)
        self._quantities = {k: np.asarray(v, dtype=np.float32) for k, v in (quantities or {}).items()}
        self._rgb = None if rgb is None else np.asarray(rgb, dtype=np.float32)
        # SSP magnitudes per band (snap['I_mag'], ...): the rgb masses derive from them, on the device when rendered
        self._mags = None if band_magnitudes is None else {k: np.asarray(v, dtype=np.float64) for k, v in band_magnitudes.items()}
        self._units = units
        self._period = periodicity_scale
        if not (len(self._pos) == len(self._smooth) == len(self._mass)):
            raise ValueError("pos, smooth and mass must have the same length")
        if with_cells:
            # cell sort + shuffle inside cells, as PynbodyDataInMemory.__init__ (loader.py:88-97)
            lo, hi = self._pos.min(), self._pos.max()
            pad = config.CELL_LAYOUT_FRACTIONAL_PADDING * (hi - lo)
            self._cell_layout, order = cell_layout.CellLayout.from_positions(self._pos, lo - pad, hi + pad,
                                                                              config.DEFAULT_CELLS_NSIDE)
            order = order[self._cell_layout.randomize_within_cells()]
            self._pos, self._smooth, self._mass = self._pos[order], self._smooth[order], self._mass[order]
            self._quantities = {k: v[order] for k, v in self._quantities.items()}
            if self._rgb is not None:
                self._rgb = self._rgb[order]
            if self._mags is not None:
                self._mags = {k: v[order] for k, v in self._mags.items()}

    def __len__(self):
        return len(self._pos)

    def get_positions(self):
        return self._pos

    def get_smooth(self):
        return self._smooth

    def get_mass(self):
        return self._mass

    def get_named_quantity(self, name):
        return self._quantities[name]

    def get_quantity_names(self):
        return list(self._quantities)

    def get_quantity_label(self, quantity_name):
        return "density" if quantity_name is None else quantity_name

    def get_rgb_masses(self):
        if self._rgb is not None:
            return self._rgb
        if self._mags is None:
            raise KeyError("no rgb band masses were supplied")
        # _effective_mass_for_band / get_rgb_masses of the reference (loader.py:112-121), on the host
        rgb = np.empty((len(self), 3), dtype=np.float32)
        for c, (band, weight) in enumerate(self.RGB_BANDS):
            rgb[:, c] = (10 ** (-0.4 * self._mags[band])) * weight
        rgb[np.isnan(rgb)] = 0.0
        return rgb

    def get_band_magnitudes(self):
        """(mags (3, n) float64, weights (3, 3) float64) for the device-side contraction (tsp_upload_band_magnitudes), or None."""
        if self._rgb is not None or self._mags is None:
            return None
        mags = np.stack([self._mags[band] for band, _ in self.RGB_BANDS])
        return mags, np.diag([w for _, w in self.RGB_BANDS]).astype(np.float64)

    def get_position_units(self):
        return self._units

    def get_periodicity_scale(self):
        return self._period

    def get_initial_view_width(self):
        return float(np.ptp(self._pos)) if self._period is None else self._period / 2


class DeviceSyntheticLoader(AbstractDataLoader):
    """TestDataLoader's distribution generated ON the GPU by a counter-based generator
    (tsp_generate_synthetic), for sizes that must not be materialised in numpy (1e8-1e9):
    shard [first, first+count) of an n_total-particle snapshot.  The arrays never visit the host;
    ParticleBuffers recognises this loader and skips the upload."""
    on_device = True

    def __init__(self, device=None, n_total=config.TEST_DATA_NUM_PARTICLES_DEFAULT, first=0, count=None, seed=1337,
                 h_cap=0.0, spatial_order=True):
        super().__init__(device)
        self.n_total = int(n_total)
        self.first = int(first)
        self.count = self.n_total - self.first if count is None else int(count)
        self.seed = seed
        self.h_cap = float(h_cap)
        self.spatial_order = spatial_order

    def __len__(self):
        return self.count

    def _not_on_host(self, *a):
        raise RuntimeError("DeviceSyntheticLoader keeps its arrays on the GPU; use Context.download_particles")

    get_positions = get_smooth = get_mass = get_rgb_masses = _not_on_host

    def get_named_quantity(self, name):
        if name != "test-quantity":
            raise KeyError("Unknown quantity name")
        return None     # generated on device

    def get_quantity_names(self):
        return ["test-quantity"]

    def get_quantity_label(self, quantity_name):
        return "test density" if quantity_name is None else "test quantity"

    def get_position_units(self):
        return "kpc"

    def get_periodicity_scale(self):
        return None
