"""Frame timing for the render loop (the role of reference src/topsy/util.py:76-115).

The reference brackets `queue.submit` with two blocking `on_submitted_work_done_sync()` calls and wall-clocks the gap,
so the time a block is charged includes the host-side work of issuing it.  Every `tsp_render` call is synchronous
(it returns when the GPU is idle again), so the same quantity is the wall-clock time around the call: `block()` is a
context manager that measures it.  The GPU-only milliseconds that tsp_render reports from its hipEvent pair are kept
next to it (`last_gpu_seconds`) for diagnostics; the 30-fps budget of RenderProgression sees the wall-clock time, as in
the reference."""
import time

import numpy as np


class GpuFrameTimer:
    def __init__(self, n_frames_smooth=10):
        self.n_frames_smooth = n_frames_smooth
        self._recent = []
        self._in_frame = 0.0
        self._gpu_in_frame = 0.0
        self.last_duration = 0.0
        self.last_gpu_seconds = 0.0

    # -- one block: `with timer.block() as b: b.gpu_ms = context.render(...)` ------------------------------------------
    class _Block:
        def __init__(self, timer):
            self._timer = timer
            self.gpu_ms = 0.0

        def __enter__(self):
            self._t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            self._timer.add_block(self.gpu_ms, wall_seconds=time.perf_counter() - self._t0)
            return False

    def block(self):
        return GpuFrameTimer._Block(self)

    def add_block(self, gpu_milliseconds, wall_seconds=None):
        """Charge one block to the frame: its wall-clock seconds (never less than the GPU time it reports)."""
        gpu_s = gpu_milliseconds * 1e-3
        self._gpu_in_frame += gpu_s
        self._in_frame += gpu_s if wall_seconds is None else max(wall_seconds, gpu_s)

    def total_time_in_frame(self):
        """Seconds charged to the blocks since the last end_frame()."""
        return self._in_frame

    def end_frame(self):
        self.last_duration = self._in_frame
        self.last_gpu_seconds = self._gpu_in_frame
        self._in_frame = 0.0
        self._gpu_in_frame = 0.0
        self._recent.append(self.last_duration)
        if len(self._recent) > self.n_frames_smooth:
            self._recent.pop(0)

    @property
    def running_mean_duration(self):
        return float(np.mean(self._recent)) if self._recent else 0.0
