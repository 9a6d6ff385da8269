"""Spatial cells over the particle set (host-side, numpy only).

Behavioural mirror of reference src/topsy/cell_layout.py: particles are bucketed on an nside^3
grid, stored cell after cell, and a cell subset can be selected by a sphere for view culling.
"""
import numpy as np


class CellLayout:
    def __init__(self, centres, offsets, lengths):
        self._centres = np.ascontiguousarray(centres)
        self._offsets = offsets
        self._lengths = lengths
        self._num_particles = lengths.sum()
        # grid spacing = distance between the first two centres (they differ along one axis only)
        self._cell_size = np.linalg.norm(self._centres[1] - self._centres[0])

    # -- construction -----------------------------------------------------------------------
    @classmethod
    def from_positions(cls, particle_positions, box_min, box_max, nside):
        """Bucket positions (N,3) lying in [box_min, box_max)^3 into nside^3 cells.

        Returns (layout, ordering) with `ordering` the argsort that groups particles by cell
        (reference cell_layout.py:63-113)."""
        lo, hi = particle_positions.min(), particle_positions.max()
        if lo < box_min or hi >= box_max:
            raise ValueError("Particle positions are outside the box")
        width = (box_max - box_min) / nside
        first = box_min + width / 2
        centres = np.mgrid[first:box_max:width, first:box_max:width, first:box_max:width].reshape(3, -1).T
        ijk = np.floor((particle_positions - box_min) / width).astype(np.intp)
        if ijk.min() < 0 or ijk.max() >= nside:
            raise ValueError("Particle positions are too close to edge of box; expand box size")
        cell_of = ijk[:, 2] + nside * (ijk[:, 1] + nside * ijk[:, 0])
        ordering = np.argsort(cell_of)
        lengths = np.bincount(cell_of, minlength=nside ** 3)
        assert len(lengths) == len(centres)
        return cls(centres, np.cumsum(lengths) - lengths, lengths), ordering

    # -- queries ----------------------------------------------------------------------------
    def randomize_within_cells(self):
        """A permutation that shuffles particles inside every cell and keeps the cells in place."""
        out = np.empty(self._lengths.sum(), dtype=np.uintp)
        for start, count in zip(self._offsets, self._lengths):
            out[start:start + count] = start + np.random.permutation(count)
        return out

    def cells_in_sphere(self, centre, radius):
        """Indices of cells whose centre lies within radius + one cell diagonal of `centre`."""
        reach = radius + self._cell_size * np.sqrt(3.0)
        return np.where(np.linalg.norm(self._centres - centre, axis=1) < reach)[0]

    def cell_index_from_offset(self, offset):
        idx = np.searchsorted(self._offsets, offset, side="right") - 1
        if idx < 0 or idx >= len(self._lengths):
            raise ValueError("Offset is out of bounds")
        return idx

    def cell_slice(self, cell_index):
        begin = self._offsets[cell_index]
        return slice(begin, begin + self._lengths[cell_index])

    def get_num_cells(self):
        return len(self._lengths)

    def get_num_particles(self):
        return self._num_particles

    def get_cell_length(self, cell_index):
        return self._lengths[cell_index]

    def get_cell_offset(self, cell_index):
        return self._offsets[cell_index]


class StratifiedCells:
    """Cells of the LIBRARY's load-time ordering (tsp_reorder_spatial): the particles are stored as uniform random
    strata, each Morton-sorted, so inside a stratum every cell of a (2^k)^3 grid over the bounding box is one contiguous
    index run (tsp_get_cell_layout / tsp_get_cell_offsets).  This class plays the part of the reference's CellLayout +
    RenderProgressionWithCells._map_logical_range_to_actual_ranges (src/topsy/cell_layout.py:26-31,
    src/topsy/progressive_render.py:152-187) for that ordering: pick the cells that meet the view sphere, and turn a block
    of whole strata into the (start, len) runs of the picked cells.  With several GPUs every shard brings its own grid
    (one `group` per shard, offsets shifted to global indices)."""

    def __init__(self, layouts):
        self.groups = []
        for lay in layouts:
            ca = int(lay["cells_per_axis"])
            ncell = ca ** 3
            code = np.arange(ncell)
            cxyz = np.zeros((ncell, 3), dtype=np.int64)
            for j in range(max(ca.bit_length() - 1, 0)):            # de-interleave the Morton cell code
                for a in range(3):
                    cxyz[:, a] |= ((code >> (3 * j + a)) & 1) << j
            width = np.asarray(lay["cell_width"], dtype=np.float64)
            centres = np.asarray(lay["box_lo"], dtype=np.float64) + (cxyz + 0.5) * width
            self.groups.append({"ncell": ncell, "n_strata": int(lay["n_strata"]), "centres": centres,
                                "reach": float(np.linalg.norm(width)),      # one cell diagonal, as cells_in_sphere adds
                                "offsets": np.asarray(lay["offsets"], dtype=np.int64)})
        self.select_all()

    @classmethod
    def from_context(cls, context):
        layouts = context.cell_layouts()
        return cls(layouts) if layouts else None

    def get_num_cells(self):
        return sum(g["ncell"] for g in self.groups)

    def _set_selection(self, masks):
        self._runs = []
        picked = 0
        for g, m in zip(self.groups, masks):
            picked += int(m.sum())
            edge = np.diff(np.concatenate(([0], m.astype(np.int8), [0])))
            self._runs.append((np.flatnonzero(edge == 1), np.flatnonzero(edge == -1)))      # runs [a, b) of picked cell codes
        self._all = picked == self.get_num_cells()
        self._fraction = max(1, picked) / max(1, self.get_num_cells())

    def select_all(self):
        self._set_selection([np.ones(g["ncell"], dtype=bool) for g in self.groups])

    def select_sphere(self, centre, radius):
        """Cells whose centre lies within radius + one cell diagonal of `centre` (reference cell_layout.py:26-31)."""
        centre = np.asarray(centre, dtype=np.float64)
        self._set_selection([np.linalg.norm(g["centres"] - centre, axis=1) < radius + g["reach"] for g in self.groups])

    def all_selected(self):
        return self._all

    def get_fraction_selected(self):
        return self._fraction

    # runs of picked cells closer than this many particles are drawn as one range: the particles in between belong to
    # cells just outside the (already conservative) selection, and tsp_render streams 512-particle chunks per range
    MERGE_GAP = 1024

    def ranges(self, start, end):
        """(starts, lens) of the picked cells' runs inside the index range [start, end) (normally a union of whole strata),
        ascending; runs separated by fewer than MERGE_GAP particles are merged."""
        out_s, out_l = [], []
        for g, (ra, rb) in zip(self.groups, self._runs):
            off, ncell = g["offsets"], g["ncell"]
            strata = off[::ncell]
            s0 = max(int(np.searchsorted(strata, start, side="right")) - 1, 0)
            s1 = min(int(np.searchsorted(strata, end, side="left")), g["n_strata"])
            if s1 <= s0 or len(ra) == 0:
                continue
            base = np.arange(s0, s1, dtype=np.int64)[:, None] * ncell
            st = np.maximum(off[base + ra[None, :]], start).ravel()
            en = np.minimum(off[base + rb[None, :]], end).ravel()
            keep = en > st
            out_s.append(st[keep])
            out_l.append((en - st)[keep])
        if not out_s:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        st, ln = np.concatenate(out_s), np.concatenate(out_l)
        if len(st) > 1:
            en = st + ln
            first = np.concatenate(([True], st[1:] - en[:-1] > self.MERGE_GAP))      # (groups are index-disjoint and ascending)
            idx = np.flatnonzero(first)
            st = st[idx]
            ln = np.concatenate((en[idx[1:] - 1], en[-1:])) - st
        return st, ln
