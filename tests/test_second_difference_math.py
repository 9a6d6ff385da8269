"""The mathematics of kernel I (csrc/tsp_integrated.hip) on the CPU: the mixed second difference of a bilinear footprint is
sparse -- 4 entries per pair of texel breakpoints, strengths from one 66 x 66 table -- and two prefix sums along either axis give
the footprint back.  numpy float64 model (tests/second_difference_model.py) against the oracle's direct evaluation."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

f32 = np.float32


@pytest.fixture(scope="module")
def model():
    import second_difference_model as sd
    from oracle import oracle_np
    mips = oracle_np.kernel_mips()
    return sd, mips, sd.second_difference_table(mips[:4096].reshape(64, 64))


def test_table_closes_exactly(model):
    """Every row and column of S0 sums to zero, and so does its first moment over the 64 slope changes: beyond the last
    breakpoint the integrated footprint is exactly zero in exact arithmetic (sums of <= 9 float32 values: exact in float64)."""
    _, _, S0 = model
    assert S0.shape == (66, 66)
    assert np.abs(S0.sum(axis=0)).max() == 0.0 and np.abs(S0.sum(axis=1)).max() == 0.0
    assert np.abs((S0[:, 1:65] * np.arange(64)).sum(axis=1)).max() < 1e-12


@pytest.mark.parametrize("case", [(130.3, 100.7, 130.0), (128.0, 128.0, 256.0), (60.2, 200.9, 313.7), (20.0, 10.0, 600.0),
                                  (-50.0, 300.0, 1500.0), (250.5, 5.25, 400.0), (128.5, 128.5, 4000.0)])
def test_scatter_and_integrate_reproduces_the_footprint(model, case):
    sd, mips, S0 = model
    R = 256
    cx, cy, P = case
    P = f32(P); half = f32(0.5) * P; invP = f32(1.0) / P
    D2 = np.zeros((R, R))
    entries = sd.scatter(D2, S0, f32(cx), f32(cy), half, P, 1.0, R)
    assert entries <= 4 * 66 * 66
    V = sd.integrate(D2)
    ref = np.zeros((R, R))
    sd.direct(ref, mips, cx, cy, half, invP, P, 1.0, R)
    peak = float(mips[:4096].max())
    assert np.abs(V - ref).max() <= 1e-6 * peak           # float32 texel coordinates of the oracle against exact ones
    outside = ref == 0.0
    if outside.any():
        assert np.abs(V[outside]).max() <= 1e-9 * peak    # cancellation leak (float64)
