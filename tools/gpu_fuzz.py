import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle_np, oracle_c
from topsy_amd import _native as native, kernel_lut
mips = kernel_lut.kernel_mips()
# TSP_FUZZ_OPTS="huge_variant=7": library options forced on every context (kernel variants the record counts of
# these small scenes would not select)
OPTS = [kv.split("=") for kv in os.environ.get("TSP_FUZZ_OPTS", "").split()]
def make_ctx(R, C):
    c = native.Context(R, C); c.set_kernel_mips(mips)
    for k, v in OPTS: c.set_option(k, int(v))
    return c
def rot(a, b):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    return np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
bad = 0
MODE = sys.argv[3] if len(sys.argv) > 3 else "weighted"
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rs = np.random.RandomState(seed)
    R = int(rs.choice([1, 3, 17, 64, 65, 100, 127, 129, 255, 300, 512, 777, 1024]))
    scale = float(np.exp(rs.uniform(np.log(2.0), np.log(800.0))))
    M, sf = oracle_np.transform_matrix(rot(rs.uniform(-3, 3), rs.uniform(-3, 3)), rs.normal(size=3) * 5.0, scale)
    n = int(rs.choice([1, 7, 500, 4000, 20000]))
    pos = (rs.normal(size=(n, 3)) * rs.uniform(1.0, 80.0, size=3)).astype(np.float32)
    hmax = scale * rs.choice([0.01, 0.05, 0.5, 3.0, 20.0])
    h = np.exp(rs.uniform(np.log(hmax * 1e-4), np.log(hmax), size=n)).astype(np.float32)
    if rs.rand() < 0.3:    # snap some footprints exactly onto class thresholds
        k = min(n, 50)
        Pt = rs.choice([64.0, 45.254833995939045, 22.627416997969522, 11.313708498984761, 128.0, 1.0, 512.0, 64.00001, 511.99997, 16.0, 15.999999, 256.0, 255.99998, 127.99999, 13.5], size=k)
        h[:k] = (Pt * scale / (2 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    if MODE != "weighted":
        rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
        md = native.MODE_RGB if MODE == "rgb" else native.MODE_DEPTH
        ctx = make_ctx(R, 4 if MODE == "rgb" else 2)
        ctx.upload_particles(x, y, z, h, None if MODE == "rgb" else m)
        if MODE == "rgb":
            ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
            want, nfrag = oracle_c.splat(x, y, z, h, rgb[:, 0].copy(), rgb[:, 1].copy(), rgb[:, 2].copy(), mode=2, M=M, sf=sf, R=R, mips=mips)
        else:
            want, nfrag = oracle_c.splat(x, y, z, h, m, None, None, mode=1, M=M, sf=sf, R=R, mips=mips)
        if n > 1 and rs.rand() < 0.5:      # the load-time ordering (strata, Morton, blocks by smoothing length): any order must give the same image
            ctx.reorder_spatial(int(rs.choice([1, 3, 8])), seed)
        # split into two accumulated blocks with several ranges each
        cut = n // 3
        ctx.render(M, sf, np.array([0]), np.array([cut]), clear=True, mode=md)
        ctx.render(M, sf, np.array([cut, cut + (n - cut) // 2]), np.array([(n - cut) // 2, n - cut - (n - cut) // 2]), clear=False, mode=md)
        g = ctx.read_image()
        nv = 3 if MODE == "rgb" else 2
        ok = bool(np.allclose(g[..., :nv], want[..., :nv], rtol=1e-5, atol=0))
        if MODE == "rgb": ok &= bool(np.array_equal(g[..., 3], want[..., 3]))
        if not ok:
            bad += 1
            print("FAIL", MODE, "seed", seed, "R", R, "scale", scale, "n", n, "hmax", hmax)
        ctx.close()
        continue
    ctx = make_ctx(R, 2)
    ctx.upload_particles(x, y, z, h, m); ctx.upload_quantity(q)
    if n > 1 and rs.rand() < 0.5:
        ctx.reorder_spatial(int(rs.choice([1, 3, 8])), seed)
    ctx.set_option("count_fragments", 1)
    ctx.render(M, sf)
    got = ctx.read_image(); nf = ctx.stats()["n_fragments"]
    want, nfrag = oracle_c.splat(x, y, z, h, m, q, mode=0, M=M, sf=sf, R=R, mips=mips)
    terms, _ = oracle_c.splat(x, y, z, h, m, np.abs(q), mode=0, M=M, sf=sf, R=R, mips=mips)
    ctx.set_option("count_fragments", 0)
    ctx.render(M, sf)
    got2 = ctx.read_image()
    ok = nf == nfrag
    for g in (got, got2):
        ok &= bool((np.abs(g[..., 0] - want[..., 0]) <= 1e-5 * np.abs(want[..., 0]) + 1e-30).all())
        ok &= bool((np.abs(g[..., 1] - want[..., 1]) <= 1e-5 * terms[..., 1] + 1e-30).all())
    if not ok:
        bad += 1
        d = np.abs(got2[..., 0] - want[..., 0]) / np.maximum(np.abs(want[..., 0]), 1e-300)
        print("FAIL seed", seed, "R", R, "scale", scale, "n", n, "hmax", hmax, "frags", nf, nfrag, "max rel", d.max())
    ctx.close()
print("done", sys.argv[1], sys.argv[2], "failures", bad)
