import sys, os
sys.path.insert(0, ".")
import numpy as np
from topsy_amd import kernel_lut, _native
for n, scale, strata in ((int(1e8), 20.0, 8), (int(1e8), 200.0, 8)):
    ctx = _native.Context(1024, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
    ctx.generate_synthetic(n, 0, n, 1337, 0.0); ctx.reorder_spatial(strata, 1337)
    M = np.eye(4, dtype=np.float32); M[:3, :3] /= scale; M[2, :] = [0, 0, 0.5 / scale, 0.5]
    ctx.set_option("count_fragments", 1); ctx.render(M, 1.0 / scale); st = ctx.stats()
    print(scale, "mid records", st["n_mid"], "pairs", st["n_fragments_mega"], "per record", st["n_fragments_mega"] / max(st["n_mid"], 1), "mid frags", st["n_fragments_mid"], "frags/pair", st["n_fragments_mid"] / max(st["n_fragments_mega"], 1))
    ctx.close()
