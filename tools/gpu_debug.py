import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_cloud
from oracle import oracle_np, oracle_c
from topsy_amd import kernel_lut, _native
mips = kernel_lut.kernel_mips()
pos, h, m, q, _ = make_cloud(20000, seed=3)
M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 200.0)
x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
R = 1024
ctx = _native.Context(R, 2); ctx.set_kernel_mips(mips)
ctx.set_option("count_fragments", 1)
def cmp(sel, label, flags=_native.PIPE_DEFAULT):
    w2, nf = oracle_c.splat(x[sel], y[sel], z[sel], h[sel], m[sel], q[sel], mode=0, M=M, sf=sf, R=R, mips=mips)
    ctx.upload_particles(x[sel], y[sel], z[sel], h[sel], m[sel]); ctx.upload_quantity(q[sel])
    ms = ctx.render(M, sf, flags=flags); g2 = ctx.read_image()
    rel = np.abs(g2[..., 0] - w2[..., 0]) / np.maximum(np.abs(w2[..., 0]).astype(np.float64), 1e-300)
    rel[(w2[..., 0] == 0) & (g2[..., 0] == 0)] = 0
    st = ctx.stats()
    print(f"{label}: n={sel.sum()} max rel {rel.max():.3g} bad {(rel > 1e-5).sum()} frags gpu {st['n_fragments']} oracle {nf} "
          f"small/mid/huge/cull {st['n_small']}/{st['n_mid']}/{st['n_huge']}/{st['n_culled']} ms {ms:.3f} (S {st['ms_stream']:.3f} M {st['ms_mid']:.3f} H {st['ms_huge']:.3f})")
    return g2, w2, rel
P = 2 * h * R / 200.0
allsel = np.ones(len(h), bool)
g, w, rel = cmp(allsel, "all")
jj, ii = np.where(rel > 1e-5)
for j, i in list(zip(jj, ii))[:8]:
    print("   px", j, i, g[j, i, 0], w[j, i, 0], rel[j, i])
for lo, hi in [(0, 4), (4, 64), (64, 1e9)]:
    cmp((P >= lo) & (P < hi), f"class P in [{lo},{hi})")
