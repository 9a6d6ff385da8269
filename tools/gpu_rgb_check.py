"""rgb: kernel H3 (three accumulator sets) against kernel H on the same snapshot -- values within 1e-5, count channel exact (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5000000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ctx = _native.Context(R, 4); ctx.set_kernel_mips(kernel_lut.kernel_mips())
ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_rgb=True); ctx.reorder_spatial(32, 1337)
M = np.eye(4, dtype=np.float32); M[:3, :3] /= 200.0; M[2, :] = [0, 0, 0.5 / 200.0, 0.5]
ctx.render(M, 1 / 200.0, mode=_native.MODE_RGB); a = ctx.read_image().astype(np.float64); st0 = ctx.stats()
for var, pm in ((1, 512), (1, 256), (2, 512), (3, 512)):
    ctx.set_option("rgb_mega_variant", var); ctx.set_option("p_mega_px", pm)
    ms = ctx.render(M, 1 / 200.0, mode=_native.MODE_RGB); b = ctx.read_image().astype(np.float64); st = ctx.stats()
    rel = np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(a[..., :3]), 1e-300)
    print(f"variant {var} p_mega {pm}: {ms:.2f} ms (H {st['ms_huge']:.2f} + H3 {st['ms_mega']:.2f}; variant 0: {st0['ms_total']:.2f}, H {st0['ms_huge']:.2f})  n_mega {st['n_mega']}  max rel diff {rel.max():.2e}  count exact {np.array_equal(a[..., 3], b[..., 3])}")
