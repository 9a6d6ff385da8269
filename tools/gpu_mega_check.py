"""Kernel H3 variants (mega_variant 0/1/2) on the same snapshot: image within 1e-5, fragment counts equal, timings (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
R = 1024
ctx = _native.Context(R, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
ctx.generate_synthetic(n, 0, n, 1337, 0.0); ctx.reorder_spatial(32, 1337)
M = np.eye(4, dtype=np.float32); M[:3, :3] /= 200.0; M[2, :] = [0, 0, 0.5 / 200.0, 0.5]
def run(opts, count=False):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_fragments", 1 if count else 0)
    for _ in range(4): ms = ctx.render(M, 1 / 200.0)
    return ctx.read_image().astype(np.float64), ctx.stats(), ms
a, st0, ms0 = run({"mega_variant": 1}); _, stc, _ = run({"mega_variant": 1}, count=True)
print(f"variant 0: total {ms0:.2f} ms  H3 {st0['ms_mega']:.2f}")
for var in (2, 3):
    for sp in (0, 128, 256):
        b, st, ms = run({"mega_variant": var, "mega_split": sp}); _, stf, _ = run({"mega_variant": var, "mega_split": sp}, count=True)
        rel = np.abs(a[..., 0] - b[..., 0]) / np.maximum(np.abs(a[..., 0]), 1e-300)
        print(f"variant {var} split {sp}: total {ms:.2f} ms  H3 {st['ms_mega']:.2f}  max rel diff {rel.max():.2e}  frags equal {stf['n_fragments'] == stc['n_fragments']}")
