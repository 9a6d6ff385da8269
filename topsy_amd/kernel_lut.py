"""Kernel texture of the splat path: the 64^2 image of the projected SPH kernel plus 3 mips.

Mirrors SPH._setup_kernel / _get_kernel_at_resolution / _get_kernel_image_normalization /
_setup_kernel_texture (reference src/topsy/sph.py:364-426).  The reference asks
`pynbody.sph.kernels.Kernel2D().get_value(d)` for every texel; pynbody is not a dependency of
this backend, so Kernel2D is restated here from its published definition: the line-of-sight
integral of the M4 cubic spline, 2 * int_0^sqrt(4-d^2) W3(sqrt(z^2+d^2)) dz (support 2h).  The
integrand is piecewise smooth (break at r = 1), so each piece is integrated with 48-point
Gauss-Legendre -- accurate to ~1e-15, no scipy needed.  The overall constant cancels in the
per-level renormalisation.
"""
import numpy as np

MIP_SIZES = (64, 32, 16, 8)

_GL_X, _GL_W = np.polynomial.legendre.leggauss(48)


def _w3(r):
    """M4 cubic spline profile, h = 1 (1/pi dropped)."""
    r = np.asarray(r, dtype=np.float64)
    inner = 1.0 - 1.5 * r ** 2 + 0.75 * r ** 3
    outer = 0.25 * np.clip(2.0 - r, 0.0, None) ** 3
    return np.where(r < 1.0, inner, outer)


def _integrate(d, a, b):
    """int_a^b W3(sqrt(z^2 + d^2)) dz for arrays d, a, b (Gauss-Legendre on each interval)."""
    mid = 0.5 * (a + b)[..., None]
    half = 0.5 * (b - a)[..., None]
    z = mid + half * _GL_X
    return (half * _GL_W * _w3(np.sqrt(z * z + d[..., None] ** 2))).sum(axis=-1)


def kernel2d(d):
    """Projected (column-integrated) cubic-spline kernel at impact parameter d; 0 beyond 2."""
    d = np.asarray(d, dtype=np.float64)
    inside = d < 2.0
    dd = np.where(inside, d, 0.0)
    zmax = np.sqrt(np.clip(4.0 - dd * dd, 0.0, None))
    zbreak = np.sqrt(np.clip(1.0 - dd * dd, 0.0, None))      # r = 1 crossing (0 when d >= 1)
    val = 2.0 * (_integrate(dd, np.zeros_like(dd), zbreak) + _integrate(dd, zbreak, zmax))
    return np.where(inside, val, 0.0)


def kernel_image(n_samples):
    """n x n texel-centre samples of the projected kernel over [-2, 2]^2, rescaled so the level sums
    to (n/4)^2, i.e. integrates to 1 at h = 1 (reference sph.py:372-394)."""
    centres = np.linspace(-2 + 2.0 / n_samples, 2 - 2.0 / n_samples, n_samples)
    gx, gy = np.meshgrid(centres, centres)
    im = kernel2d(np.sqrt(gx ** 2 + gy ** 2))
    return im * ((n_samples / 4) ** 2 / im.sum())


_cache = {}


def kernel_mips(n0=64, n_levels=4):
    """All mip levels, concatenated float32: what _setup_kernel_texture uploads (sph.py:396-426)."""
    key = (n0, n_levels)
    if key not in _cache:
        _cache[key] = np.concatenate(
            [kernel_image(n0 // 2 ** i).astype(np.float32).ravel() for i in range(n_levels)])
    return _cache[key]
