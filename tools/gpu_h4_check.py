"""Kernel H4 (huge_variant 3) against H2 / H3 on the same snapshot: image within 1e-5 / exact fragment counts, timings (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
mode_name = sys.argv[2] if len(sys.argv) > 2 else "density"
R = 1024
ctx = _native.Context(R, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
ctx.generate_synthetic(n, 0, n, 1337, 0.0, with_quantity=(mode_name == "weighted")); ctx.reorder_spatial(32, 1337)
M = np.eye(4, dtype=np.float32); M[:3, :3] /= 200.0; M[2, :] = [0, 0, 0.5 / 200.0, 0.5]
def run(opts, count=False):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("count_fragments", 1 if count else 0)
    for _ in range(3): ms = ctx.render(M, 1 / 200.0)
    return ctx.read_image().astype(np.float64), ctx.stats(), ms
a, st0, ms0 = run({"huge_variant": 1})
_, stc, _ = run({"huge_variant": 1}, count=True)
print(f"H2/H3: total {ms0:.2f} ms  H2 {st0['ms_huge']:.2f}  H3 {st0['ms_mega']:.2f}  frags {stc['n_fragments']}")
for pm in (512, 1024, 2048, 0):
    b, st, ms = run({"huge_variant": 3, "p_mega_px": pm, "p_mega2_px": pm})
    _, stf, _ = run({"huge_variant": 3, "p_mega_px": pm, "p_mega2_px": pm}, count=True)
    rel = np.abs(a[..., 0] - b[..., 0]) / np.maximum(np.abs(a[..., 0]), 1e-300)
    d1 = np.abs(a[..., 1] - b[..., 1]).max() / max(np.abs(a[..., 1]).max(), 1e-300)
    print(f"H4 p_mega {pm}: total {ms:.2f} ms  H4 {st['ms_huge']:.2f}  H3 {st['ms_mega']:.2f}  n_mega {st['n_mega']}  max rel diff {rel.max():.2e}  ch1 {d1:.2e}  frags equal {stf['n_fragments'] == stc['n_fragments']}")
