"""Datapoint, not a test: the at-scale and Visualizer GPU tests with the option integrated_px = 256 forced on every context
(their assertions are the per-pixel ones of the exact kernels; what fails here is where kernel I's weaker contract shows)."""
import sys
sys.path.insert(0, ".")
import pytest
from topsy_amd import _native
_orig = _native.Context.__init__


def _init(self, *a, **k):
    _orig(self, *a, **k)
    self.set_option("integrated_px", int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256)


_native.Context.__init__ = _init
sys.exit(pytest.main(["tests/test_gpu_scale.py", "tests/test_gpu_visualizer.py", "tests/test_gpu_multirank.py", "-q", "-m", "gpu", "-p", "no:cacheprovider"]))
