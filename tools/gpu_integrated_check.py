"""Kernel I (option integrated_px) against the exact kernels on the GPU: a few isolated wide footprints, then the synthetic
snapshot at several class boundaries (not a test: prints differences and times)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native

R = 1024
n_big = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
scale = 200.0
M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]

ctx = _native.Context(R, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
# ---- isolated footprints: P = 2 h R / scale
cases = [(0.0, 0.0, 130.0), (50.3, -20.7, 256.0), (-80.2, 77.9, 513.7), (-160.0, 180.0, 900.0), (-280.0, -230.0, 3000.0),
         (195.5, 190.25, 700.0), (0.1, 0.1, 4000.0), (-198.0, 198.0, 128.0), (10.0, 10.0, 200.0)]
for k in range(len(cases) + 1):
    sel = cases if k == len(cases) else [cases[k]]
    xs = np.array([c[0] for c in sel], dtype=np.float32); ys = np.array([c[1] for c in sel], dtype=np.float32)
    hs = np.array([c[2] * scale / (2 * R) for c in sel], dtype=np.float32)
    ctx.upload_particles(xs, ys, np.zeros_like(xs), hs, np.ones_like(xs))
    ctx.set_option("count_fragments", 1)
    ctx.set_option("integrated_px", 0); ctx.render(M, 1.0 / scale, mode=_native.MODE_WEIGHTED); ref = ctx.read_image()[..., 0].astype(np.float64); f0 = ctx.stats()["n_fragments"]
    ctx.set_option("integrated_px", 128); ctx.render(M, 1.0 / scale, mode=_native.MODE_WEIGHTED); got = ctx.read_image()[..., 0].astype(np.float64); f1 = ctx.stats()["n_fragments"]
    ctx.set_option("count_fragments", 0)
    err = np.abs(got - ref); nz = ref > 0
    print(f"case {k} {sel if len(sel)==1 else 'all'}: peak {ref.max():.3e}  max|err| {err.max():.3e} ({err.max()/ref.max():.2e} of peak)  "
          f"leak where ref=0 {np.abs(got[~nz]).max() if (~nz).any() else 0:.3e}  min {got.min():.3e}  fragments {f0} / {f1}")
# ---- the synthetic snapshot
ctx.generate_synthetic(n_big, 0, n_big, 1337, 0.0)
ctx.reorder_spatial(32, 1337)
ctx.set_option("integrated_px", 0)
for f in range(3): ms = ctx.render(M, 1.0 / scale, mode=_native.MODE_WEIGHTED)
st = ctx.stats(); ref = ctx.read_image()[..., 0].astype(np.float64)
print(f"exact: {ms:.3f} ms  S {st['ms_stream']:.3f} M {st['ms_mid']:.3f} H2 {st['ms_huge']:.3f} H3 {st['ms_mega']:.3f}  huge(mega) {st['n_huge']}({st['n_mega']})")
for px in (1024, 512, 384, 256, 192, 128):
    ctx.set_option("integrated_px", px)
    for f in range(3): ms = ctx.render(M, 1.0 / scale, mode=_native.MODE_WEIGHTED)
    st = ctx.stats(); got = ctx.read_image()[..., 0].astype(np.float64)
    rel = np.abs(got - ref) / np.maximum(ref, 1e-300)
    print(f"integrated_px {px}: {ms:.3f} ms  S {st['ms_stream']:.3f} M {st['ms_mid']:.3f} H2 {st['ms_huge']:.3f} I {st['ms_mega']:.3f}  huge(mega) {st['n_huge']}({st['n_mega']})  "
          f"max rel diff {rel.max():.3e}  99.9% {np.quantile(rel, 0.999):.2e}  median {np.median(rel):.2e}  min pixel {got.min():.3e} (exact {ref.min():.3e})")
