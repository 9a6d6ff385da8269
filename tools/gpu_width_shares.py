"""Fragment share of the 1e9 snapshot by footprint-width band (not a test): shard 0 of 8 of the headline snapshot is generated on
the device and downloaded; widths and covered pixel counts are formed on the host as tsp_math.h's cover_range does (pixel centres
inside the square, clipped to the image).  usage: gpu_width_shares.py [scale=200] [n_total=1e9] [shards=8]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
opts = dict(kv.split("=") for kv in sys.argv[1:])
R = int(opts.get("R", 1024)); scale = float(opts.get("scale", 200.0)); n_total = int(float(opts.get("n_total", 1e9))); G = int(opts.get("shards", 8))
ctx = _native.Context(R, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
ctx.generate_synthetic(n_total, first=0, count=n_total // G, seed=1337)
d = ctx.download_particles(("x", "y", "z", "h"))
x, y, z, h = (d[k].astype(np.float64) for k in ("x", "y", "z", "h"))
# s = (sf h) 2; P = s R  ->  P = 2 h R / scale
P = 2.0 * h * R / scale
pcx = (x / scale + 1.0) * R / 2.0; pcy = (1.0 - y / scale) * R / 2.0
cz = 0.5 * z / scale + 0.5
ok = (cz >= 0) & (cz <= 1)
def span(c, half):
    lo = np.clip(np.ceil(c - half - 0.5), 0, R); hi = np.clip(np.floor(c + half - 0.5) + 1, 0, R)      # pixel centres i + 0.5 within |d| < half (edge ties ignored)
    return np.maximum(hi - lo, 0)
frags = span(pcx, P / 2) * span(pcy, P / 2) * ok
edges = [0, 1, 2, 4, 8, 16, 24, 32, 48, 64, 100, 128, 180, 256, 360, 512, 720, 1024, 1500, 1e30]
tot = frags.sum(); n = len(P)
print(f"scale {scale:g}: {n} particles (shard 0 of {G} of {n_total:.3g}), {tot:.4g} fragments, {tot / n:.1f} per particle")
print(f"{'band px':>14} {'particles %':>12} {'fragments %':>12} {'frags/particle':>15}")
for a, b in zip(edges[:-1], edges[1:]):
    sel = (P >= a) & (P < b) & ok
    c = sel.sum()
    print(f"{a:6g}-{b:<7g} {100.0 * c / n:12.4f} {100.0 * frags[sel].sum() / tot:12.3f} {frags[sel].sum() / max(c, 1):15.1f}")
