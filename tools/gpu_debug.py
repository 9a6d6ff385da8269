import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from conftest import make_cloud
from oracle import oracle_np, oracle_c
from topsy_amd import kernel_lut, _native
mips = kernel_lut.kernel_mips()
pos, h, m, q, _ = make_cloud(20000, seed=3)
M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), 200.0)
x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
R = 1024
want, nf = oracle_c.splat(x, y, z, h, m, q, mode=0, M=M, sf=sf, R=R, mips=mips)
ctx = _native.Context(R, 2); ctx.set_kernel_mips(mips)
ctx.upload_particles(x, y, z, h, m); ctx.upload_quantity(q)
ctx.set_option("count_fragments", 1)
ms = ctx.render(M, sf, flags=_native.PIPE_GENERIC)
got = ctx.read_image()
print("ms", ms, ctx.stats(), "oracle frags", nf)
d = np.abs(got[..., 0] - want[..., 0]); rel = d / np.maximum(np.abs(want[..., 0]), 1e-30)
bad = rel > 1e-5
print("bad px", bad.sum(), "max rel", rel.max(), "nan got", np.isnan(got).sum())
jj, ii = np.where(bad)
for j, i in list(zip(jj, ii))[:10]:
    print(j, i, got[j, i, 0], want[j, i, 0])
# try single particles to isolate
for lo, hi in [(0.02, 1), (1, 8), (8, 64), (64, 1000)]:
    P = 2 * h * R / 200.0
    sel = (P >= lo) & (P < hi)
    w2, _ = oracle_c.splat(x[sel], y[sel], z[sel], h[sel], m[sel], q[sel], mode=0, M=M, sf=sf, R=R, mips=mips)
    ctx.upload_particles(x[sel], y[sel], z[sel], h[sel], m[sel]); ctx.upload_quantity(q[sel])
    ctx.render(M, sf, flags=_native.PIPE_GENERIC); g2 = ctx.read_image()
    rel = np.abs(g2[..., 0] - w2[..., 0]) / np.maximum(np.abs(w2[..., 0]), 1e-30)
    print("class", lo, hi, sel.sum(), "max rel", rel.max(), "bad", (rel > 1e-5).sum(), "frags", ctx.stats()['n_fragments'])
