"""Cost per fragment of kernels G (16-64 px) and H2 (>= 64 px) by footprint width (not a test): uniform random centres, one width per run."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from topsy_amd import kernel_lut, _native
opts = dict(kv.split("=") for kv in sys.argv[1:])
R = int(opts.pop("R", 1024)); scale = 200.0
widths = [float(x) for x in opts.pop("P", "16,24,32,48,66,100,180,360,720,1500").split(",")]
target = float(opts.pop("frags", 2e10))
ctx = _native.Context(R, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
for k, v in opts.items(): ctx.set_option(k, int(v))
M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
rs = np.random.RandomState(5)
print(f"{'P':>7} {'kernel':>6} {'records':>9} {'frags':>10} {'ms':>8} {'SIMD-clk/frag':>14} {'Gfrag/s':>9}")
for P in widths:
    clipped = min(P, R) ** 2 if P >= R else (P * P)
    n = int(max(2000, min(4e6, target / clipped)))
    x = rs.uniform(-scale, scale, n).astype(np.float32); y = rs.uniform(-scale, scale, n).astype(np.float32)
    z = np.zeros(n, np.float32); h = np.full(n, P * scale / (2 * R), np.float32); m = np.ones(n, np.float32)
    ctx.upload_particles(x, y, z, h, m)
    ctx.set_option("count_fragments", 1); ctx.render(M, 1.0 / scale); fr = ctx.stats()["n_fragments"]; ctx.set_option("count_fragments", 0)
    best = 1e9
    for _ in range(3):
        ctx.render(M, 1.0 / scale); st = ctx.stats(); best = min(best, st["ms_huge"] if P >= 64 else st["ms_mid"])
    nrec = st["n_huge"] if P >= 64 else st["n_mid"]
    # SIMD-clk/frag: 1024 SIMDs x 2.4 GHz x time / fragments (1/64 = one wave-instruction per 64 fragments per SIMD)
    print(f"{P:7.0f} {'H2' if P >= 64 else 'G':>6} {nrec:9d} {fr:10.3g} {best:8.3f} {best * 1e-3 * 2.4e9 * 1024 / fr:14.4f} {fr / best / 1e6:9.1f}")
