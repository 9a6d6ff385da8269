"""Max relative error of the HIP pipeline against the CPU oracle on an n-particle sample of the benchmark snapshot
(not a test; the oracle needs ~n / 3e6 s on 128 cores)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
from oracle import oracle_c
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
ntotal = int(float(sys.argv[2])) if len(sys.argv) > 2 else 125000000
R, scale = 1024, 200.0
ctx = _native.Context(R, 2); mips = kernel_lut.kernel_mips(); ctx.set_kernel_mips(mips)
ctx.generate_synthetic(ntotal, 0, n, 1337, 0.0)
d = ctx.download_particles(("x", "y", "z", "h", "mass"))
ctx.reorder_spatial(32, 1337)
M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
for opts in ({}, {"huge_variant": 7}, {"huge_variant": 5}):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.render(M, 1.0 / scale); got = ctx.read_image()[..., 0].astype(np.float64); st = ctx.stats()
    if "want" not in globals():
        want, nfrag = oracle_c.splat(d["x"], d["y"], d["z"], d["h"], d["mass"], None, None, mode=0, M=M, sf=np.float32(1.0 / scale), R=R, mips=mips)
        want = want[..., 0].astype(np.float64)
    rel = np.abs(got - want) / np.maximum(want, 1e-300)
    print(f"{opts or 'default'}: n={n} huge {st['n_huge']}  max rel err {rel[want > 0].max():.3e}  median {np.median(rel[want > 0]):.2e}")
