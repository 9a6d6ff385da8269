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
