"""Frame timing for the render loop (the role of reference src/topsy/util.py:76-115).

The reference brackets `queue.submit` with two blocking `on_submitted_work_done_sync()` calls and
wall-clocks the gap.  Here every `tsp_render` call is synchronous and reports its own GPU time
from a hipEvent pair, so the timer simply accumulates those durations."""
import numpy as np


class GpuFrameTimer:
    def __init__(self, n_frames_smooth=10):
        self.n_frames_smooth = n_frames_smooth
        self._recent = []
        self._in_frame = 0.0
        self.last_duration = 0.0

    def add_block(self, gpu_milliseconds):
        self._in_frame += gpu_milliseconds * 1e-3

    def total_time_in_frame(self):
        """Seconds of GPU work since the last end_frame()."""
        return self._in_frame

    def end_frame(self):
        self.last_duration = self._in_frame
        self._in_frame = 0.0
        self._recent.append(self.last_duration)
        if len(self._recent) > self.n_frames_smooth:
            self._recent.pop(0)

    @property
    def running_mean_duration(self):
        return float(np.mean(self._recent)) if self._recent else 0.0
