"""ctypes binding of oracle/liboracle.so (C restatement) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int64)
_bp = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_splat.restype = ctypes.c_long
        L.orc_splat.argtypes = [ctypes.c_long, _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                _fp, ctypes.c_float, ctypes.c_int, _fp, _ip, _ip, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, _fp]
        L.orc_splat_rule.restype = ctypes.c_long
        L.orc_splat_rule.argtypes = L.orc_splat.argtypes + [ctypes.c_int]
        L.orc_colormap_scalar.restype = None
        L.orc_colormap_scalar.argtypes = [_fp, ctypes.c_long, ctypes.c_int, _fp, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _bp]
        L.orc_colormap_rgb.restype = None
        L.orc_colormap_rgb.argtypes = [_fp, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, _bp, _fp]
        L.orc_colormap_bivariate.restype = None
        L.orc_colormap_bivariate.argtypes = [_fp, ctypes.c_long, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _bp]
        L.orc_logf.restype = ctypes.c_float
        L.orc_logf.argtypes = [ctypes.c_float]
        L.orc_expf.restype = ctypes.c_float
        L.orc_expf.argtypes = [ctypes.c_float]
        L.orc_powf.restype = ctypes.c_float
        L.orc_powf.argtypes = [ctypes.c_float, ctypes.c_float]
        L.orc_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def splat(x, y, z, h, a, b=None, c=None, *, mode=0, M, sf, R, mips, ranges=None, out=None,
          nthreads=0, sampling=0):
    """mode 0: a=mass, b=qty|None; mode 1 (depth): a=mass; mode 2 (rgb): a,b,c=r,g,b.
    Returns ((R,R,C) float32 image, fragment count)."""
    L = lib()
    n = len(x)
    keep = [_f(v) for v in (x, y, z, h, a, b, c)]
    Mk, Mp = _f(np.asarray(M, dtype=np.float32).reshape(16))
    mk, mp = _f(mips)
    C = 4 if mode == 2 else 2
    accumulate = 0
    if out is None:
        out = np.zeros((R, R, C), dtype=np.float32)
    else:
        accumulate = 1
        assert out.dtype == np.float32 and out.shape == (R, R, C) and out.flags.c_contiguous
    if ranges is None:
        sp = lp = None
        nr = 0
    else:
        s = np.ascontiguousarray(ranges[0], dtype=np.int64)
        l = np.ascontiguousarray(ranges[1], dtype=np.int64)
        sp, lp, nr = s.ctypes.data_as(_ip), l.ctypes.data_as(_ip), len(s)
    nfrag = L.orc_splat_rule(n, *[k[1] for k in keep], mode, Mp, ctypes.c_float(sf), R, mp, sp, lp, nr,
                             accumulate, nthreads, out.ctypes.data_as(_fp), int(sampling))
    if nfrag < 0:
        raise MemoryError("oracle.c: an allocation failed (orc_splat_rule returned -1)")
    return out, nfrag


def colormap_scalar(img, lut, vmin, vmax, log, weighted):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.float32)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    H, W, C = img.shape
    out = np.empty((H, W, 4), dtype=np.uint8)
    L.orc_colormap_scalar(img.ctypes.data_as(_fp), H * W, C, lut.ctypes.data_as(_fp), lut.shape[0],
                          vmin, vmax, int(bool(log)), int(bool(weighted)), out.ctypes.data_as(_bp))
    return out


def colormap_bivariate(img, lut2d, vmin, vmax, dvmin, dvmax, log, weighted):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.float32)
    lut2d = np.ascontiguousarray(lut2d, dtype=np.float32)
    H, W, C = img.shape
    out = np.empty((H, W, 4), dtype=np.uint8)
    L.orc_colormap_bivariate(img.ctypes.data_as(_fp), H * W, C, lut2d.ctypes.data_as(_fp), lut2d.shape[0], vmin, vmax,
                             dvmin, dvmax, int(bool(log)), int(bool(weighted)), out.ctypes.data_as(_bp))
    return out


def colormap_rgb(img, vmin, vmax, gamma, as_float=False):
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W, C = img.shape
    if as_float:
        out = np.empty((H, W, 4), dtype=np.float32)
        L.orc_colormap_rgb(img.ctypes.data_as(_fp), H * W, C, vmin, vmax, gamma, None, out.ctypes.data_as(_fp))
    else:
        out = np.empty((H, W, 4), dtype=np.uint8)
        L.orc_colormap_rgb(img.ctypes.data_as(_fp), H * W, C, vmin, vmax, gamma, out.ctypes.data_as(_bp), None)
    return out


def max_threads():
    return lib().orc_max_threads()
