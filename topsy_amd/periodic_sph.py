"""Periodic tiling of the rendered box: host-side mirror of reference src/topsy/periodic_sph.py.

The box is splatted once; the image is then replaced by the weighted sum of its copies displaced by the
projected lattice vectors (5^3 candidates, those within one box depth of the view plane), a pure
image-space post-pass (tsp_tile_periodic)."""
import numpy as np

from . import sph
from .drawreason import DrawReason


def instance_offsets_and_weights(rotation_matrix, panel_scale, num_repetitions=2):
    """Clip-space xy shifts and weights of the periodic images (reference periodic_sph.py:36-54): images
    whose rotated lattice vector has |z| < 1 box; weight 1 up to half a box, fading linearly to 0 at one."""
    offsets, weights = [], []
    span = range(-num_repetitions, num_repetitions + 1)
    for xoff in span:
        for yoff in span:
            for zoff in span:
                shift = np.asarray(rotation_matrix) @ np.array([xoff, yoff, zoff], dtype=np.float32)
                depth = abs(shift[2])
                if depth < 1.0:
                    offsets.append(shift[:2])
                    weights.append(1.0 - 2.0 * (depth - 0.5) if depth > 0.5 else 1.0)
    return np.array(offsets, dtype=np.float32) * panel_scale, np.array(weights, dtype=np.float32)


class PeriodicSPH(sph.SPH):
    def __init__(self, visualizer, render_size):
        super().__init__(visualizer, render_size, wrapping=True)
        self.num_repetitions = 2

    def render(self, draw_reason=DrawReason.CHANGE):
        # the base class decides whether anything is drawn: a PRESENTATION_CHANGE re-presents the resident tiled frame,
        # unless another renderer (e.g. the depth pass) has used the shared target since -- then the frame is redrawn
        if not super().render(draw_reason):
            return False
        panel_scale = self._visualizer.periodicity_scale / self._visualizer.scale
        offsets, weights = instance_offsets_and_weights(self.rotation_matrix, panel_scale, self.num_repetitions)
        self._context.tile_periodic(offsets, weights)
        return True
