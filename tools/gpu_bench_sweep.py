"""Timing breakdown of the pipeline on the synthetic snapshot (not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
opts = dict(kv.split("=") for kv in sys.argv[2:])
R = int(opts.pop("R", 1024)); scale = float(opts.pop("scale", 200.0)); hcap_px = float(opts.pop("hcap", 0))
ntotal = int(float(opts.pop("ntotal", n))); first = int(float(opts.pop("first", 0)))
reorder = int(opts.pop("reorder", 32)); frames = int(opts.pop("frames", 5)); mode = opts.pop("mode", "density")
ctx = _native.Context(R, 4 if mode == "rgb" else 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
for k, v in opts.items(): ctx.set_option(k, int(v))     # (load-time options such as reorder_interleave are read by reorder_spatial)
t = time.time()
ctx.generate_synthetic(ntotal, first, n, 1337, hcap_px * scale / (2.0 * R), with_quantity=(mode == "weighted"), with_rgb=(mode == "rgb"))
tg = time.time() - t; t = time.time()
if reorder: ctx.reorder_spatial(reorder, 1337)
print(f"n={n:.3g} generate {tg:.2f}s reorder {time.time()-t:.2f}s")
for k, v in opts.items(): ctx.set_option(k, int(v))
M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
md = _native.MODE_RGB if mode == "rgb" else (_native.MODE_DEPTH if mode == "depth" else _native.MODE_WEIGHTED)
for f in range(frames):
    ms = ctx.render(M, 1.0 / scale, mode=md); st = ctx.stats()
    print(f"frame {f}: total {ms:.3f} ms  S {st['ms_stream']:.3f}  M {st['ms_mid']:.3f}  H2 {st['ms_huge']:.3f}  small/mid/huge/cull {st['n_small']}/{st['n_mid']}/{st['n_huge']}/{st['n_culled']}")
ctx.set_option("count_fragments", 1); ctx.render(M, 1.0 / scale, mode=md); st = ctx.stats()
print(f"fragments {st['n_fragments']:.4g} ({st['n_fragments']/n:.1f}/particle)  -> {n/ms*1e3:.3g} particles/s, {st['n_fragments']/ms*1e3:.3g} frags/s; stream GB/s {20*n/st['ms_stream']/1e6:.0f}")
print("fragments by kernel S / M / H2:", st["n_fragments_stream"], st["n_fragments_mid"], st["n_fragments_huge"])
img = ctx.read_image(); print("mass sum", img[..., 0].sum() * (2*scale/R)**2, "expected ~", n * 1e-8)
