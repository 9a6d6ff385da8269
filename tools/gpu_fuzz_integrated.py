"""Random scenes through kernel I (option integrated_px) against the exact kernels of the same context: contract of
tests/test_gpu_integrated.py (1e-6 of the summed peak contributions of the wide footprints, exact fragment counts).
usage: python3 tools/gpu_fuzz_integrated.py <first seed> <last seed>"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle_np
from topsy_amd import _native as native, kernel_lut
mips = kernel_lut.kernel_mips()
peak = float(mips[:4096].max())


def rot(a, b):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    return np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rs = np.random.RandomState(seed)
    R = int(rs.choice([64, 65, 100, 127, 255, 300, 512, 777, 1024, 1025, 1500, 2048, 2500]))
    scale = float(np.exp(rs.uniform(np.log(2.0), np.log(800.0))))
    M, sf = oracle_np.transform_matrix(rot(rs.uniform(-3, 3), rs.uniform(-3, 3)), rs.normal(size=3) * 5.0, scale)
    n = int(rs.choice([1, 7, 300, 3000]))
    pos = (rs.normal(size=(n, 3)) * rs.uniform(1.0, 80.0, size=3)).astype(np.float32)
    hmax = scale * rs.choice([0.2, 1.0, 5.0, 40.0])
    h = np.exp(rs.uniform(np.log(hmax * 1e-2), np.log(hmax), size=n)).astype(np.float32)
    px = int(rs.choice([128, 129, 200, 256, 512]))
    if rs.rand() < 0.4:        # snap some footprints onto the class threshold and onto power-of-two breakpoint spacings
        k = min(n, 40)
        Pt = rs.choice([px, px * 0.99999, px * 1.00001, 128.0, 256.0, 512.0, 1024.0, 4096.0, 65536.0], size=k)
        h[:k] = (Pt * scale / (2 * R)).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    mode = str(rs.choice(["density", "weighted", "depth", "rgb"]))
    md = {"density": native.MODE_WEIGHTED, "weighted": native.MODE_WEIGHTED, "depth": native.MODE_DEPTH, "rgb": native.MODE_RGB}[mode]
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    ctx = native.Context(R, 4 if mode == "rgb" else 2); ctx.set_kernel_mips(mips)
    ctx.upload_particles(x, y, z, h, None if mode == "rgb" else m)
    if mode == "rgb": ctx.upload_rgb(rgb[:, 0], rgb[:, 1], rgb[:, 2])
    if mode == "weighted": ctx.upload_quantity(q)
    ctx.set_option("count_fragments", 1)
    cut = n // 3

    def frame():
        ctx.render(M, sf, np.array([0]), np.array([cut]), clear=True, mode=md); f = ctx.stats()["n_fragments"]
        ctx.render(M, sf, np.array([cut]), np.array([n - cut]), clear=False, mode=md)
        return ctx.read_image().astype(np.float64), f + ctx.stats()["n_fragments"], ctx.stats()["n_mega"]
    ref, f0, _ = frame()
    ctx.set_option("integrated_px", px)
    got, f1, nm = frame()
    ctx.close()
    hd = h.astype(np.float64)
    wide = (2.0 * hd * R / scale) >= px * 0.999
    if mode == "rgb":
        scales = [float((rgb[wide, c] / hd[wide] ** 2).sum()) * peak for c in range(3)]
    else:
        w0 = m[wide] / hd[wide] ** 2
        scales = [float(w0.sum()) * peak, float((w0 * (np.abs(q[wide]) if mode == "weighted" else 1.0)).sum()) * peak]
    ok = f0 == f1
    worst = 0.0
    for c, sc in enumerate(scales):
        if mode == "density" and c == 1: continue
        # + the exact kernels' own summation-order noise between two renders (relative to the pixel; the weighted channel cancels:
        # relative to density x max |q|)
        noise = 2e-6 * (np.abs(ref[..., 0]) * float(np.abs(q).max()) if (mode == "weighted" and c == 1) else np.abs(ref[..., c]))
        excess = np.abs(got[..., c] - ref[..., c]) - noise
        worst = max(worst, float(excess.max()) / max(sc, 1e-300))
        ok &= bool((excess <= 1e-6 * sc + 1e-300).all())
    if mode == "rgb": ok &= bool(np.array_equal(got[..., 3], ref[..., 3]))
    if not ok:
        bad += 1
        print("FAIL seed", seed, mode, "R", R, "scale", scale, "n", n, "hmax", hmax, "px", px, "frags", f0, f1, "n_mega", nm, "worst", worst)
print("done", sys.argv[1], sys.argv[2], "failures", bad)
