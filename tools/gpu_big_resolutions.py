import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from oracle import oracle_np, oracle_c
from topsy_amd import _native as native, kernel_lut
mips = kernel_lut.kernel_mips()
for R, n, seed in [(2048, 3000, 1), (4096, 1500, 2), (3000, 2000, 3), (8192, 400, 4)]:
    rs = np.random.RandomState(seed)
    scale = 100.0
    M, sf = oracle_np.transform_matrix(np.eye(3), np.zeros(3), scale)
    pos = (rs.uniform(-1.2, 1.2, size=(n, 3)) * scale).astype(np.float32)
    h = np.exp(rs.uniform(np.log(scale * 1e-4), np.log(scale * 2.0), size=n)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32); q = rs.normal(size=n).astype(np.float32)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    ctx = native.Context(R, 2); ctx.set_kernel_mips(mips); ctx.upload_particles(x, y, z, h, m); ctx.upload_quantity(q)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf); got = ctx.read_image(); st = ctx.stats()
    want, nfrag = oracle_c.splat(x, y, z, h, m, q, None, mode=0, M=M, sf=sf, R=R, mips=mips)
    rel = np.abs(got[..., 0].astype(np.float64) - want[..., 0]) / np.maximum(want[..., 0], 1e-300)
    print(f"R={R} n={n}: frags {st['n_fragments']} vs {nfrag} ({'OK' if st['n_fragments']==nfrag else 'MISMATCH'}), huge {st['n_huge']}, max rel err ch0 {rel[want[...,0]>0].max():.2e}")
    ctx.close()
