"""Model (numpy, float64; test infrastructure) of the second-difference form of a bilinear footprint: the footprint's image is piecewise
bilinear in pixel coordinates, so its mixed second difference is sparse (4 entries per pair of texel breakpoints);
scatter those, integrate twice along each axis.  Compares with the oracle's direct evaluation."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp

f32 = np.float32


def second_difference_table(T):
    """S0[q][p] over 66 x 66 breakpoints (left step, 64 slope changes, right step) of the level-0 kernel image."""
    L = np.zeros((66, 64))
    L[0, 0] = 1.0
    for a in range(64):
        L[1 + a, max(a - 1, 0)] += 1.0
        L[1 + a, a] -= 2.0
        L[1 + a, min(a + 1, 63)] += 1.0
    L[65, 63] = -1.0
    return L @ T.astype(np.float64) @ L.T


def axis_nodes(pc, half, P, R):
    pc, half, P = float(pc), float(half), float(P)
    i = np.arange(-8192, 8192 + R)
    cov = np.abs((i.astype(f32) + f32(0.5)) - f32(pc)) < f32(half)
    idx = i[cov]
    if len(idx) == 0:
        return None
    i_lo, i_hi = int(idx[0]), int(idx[-1])
    alpha = 64.0 / P
    pos = np.zeros(66, dtype=np.int64); w0 = np.zeros(66); w1 = np.zeros(66)
    pos[0] = max(i_lo, 0); w0[0] = 1.0; w1[0] = -1.0
    pos[65] = max(i_hi + 1, 0); w0[65] = 1.0; w1[65] = -1.0
    a = np.arange(64)
    t = (a + 0.5) / alpha - half + pc - 0.5
    i0 = np.maximum(np.ceil(t), 0.0)
    fr = i0 - t
    pos[1:65] = i0.astype(np.int64); w0[1:65] = alpha * fr; w1[1:65] = alpha * (1.0 - fr)
    return pos, w0, w1


def scatter(D2, S0, pcx, pcy, half, P, w, R):
    ax = axis_nodes(pcx, half, P, R); ay = axis_nodes(pcy, half, P, R)
    if ax is None or ay is None:
        return 0
    px, wx0, wx1 = ax; py, wy0, wy1 = ay
    n = 0
    for dj, wy in ((0, wy0), (1, wy1)):
        for di, wx in ((0, wx0), (1, wx1)):
            J = (py + dj)[:, None] + 0 * px[None, :]
            I = (px + di)[None, :] + 0 * py[:, None]
            ok = (J < R) & (I < R)
            val = (w * S0) * wy[:, None] * wx[None, :]
            np.add.at(D2, (J[ok], I[ok]), val[ok])
            n += int(ok.sum())
    return n


def integrate(D2):
    V = np.cumsum(np.cumsum(D2, axis=0), axis=0)
    return np.cumsum(np.cumsum(V, axis=1), axis=1)


def direct(img, mips, pcx, pcy, half, invP, P, w, R):
    fp = onp._footprint(f32(pcx), f32(pcy), f32(half), f32(invP), f32(P), R, mips)
    if fp is None:
        return
    j0, i0, K, cov = fp
    img[j0:j0 + K.shape[0], i0:i0 + K.shape[1]] += (K * f32(w)).astype(np.float64)


if __name__ == "__main__":
    R = 1024
    mips = onp.kernel_mips()
    T = mips[:4096].reshape(64, 64)
    S0 = second_difference_table(T)
    print("table closure: row sums", np.abs(S0.sum(axis=1)).max(), "first moments", np.abs((S0[:, 1:65] * np.arange(64)).sum(axis=1)).max())
    rng = np.random.default_rng(1)
    cases = [(500.3, 400.7, 130.0), (512.0, 512.0, 256.0), (300.2, 700.9, 513.7), (100.0, 50.0, 900.0), (-200.0, 1100.0, 3000.0),
             (1000.5, 20.25, 700.0), (512.5, 512.5, 4000.0), (10.0, 10.0, 128.0)]
    for (cx, cy, P) in cases:
        P = f32(P); half = f32(0.5) * P; invP = f32(1.0) / P; w = 1.0
        D2 = np.zeros((R, R)); scatter(D2, S0, f32(cx), f32(cy), half, P, w, R)
        V = integrate(D2)
        ref = np.zeros((R, R)); direct(ref, mips, cx, cy, half, invP, P, w, R)
        err = np.abs(V - ref)
        nz = ref > 0
        rel = (err[nz] / ref[nz])
        print(f"P={float(P):7.1f} centre ({cx},{cy}): max|err| {err.max():.3e} (peak {ref.max():.3e})  max rel where ref>0 {rel.max():.3e}  "
              f"median rel {np.median(rel):.2e}  leak outside {np.abs(V[~nz]).max() if (~nz).any() else 0:.3e}  min V {V.min():.3e}")
    # many footprints: accumulation error
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    D2 = np.zeros((R, R)); ref = np.zeros((R, R)); ent = 0
    for k in range(n):
        P = f32(rng.uniform(128, 3000)); cx = rng.uniform(-300, R + 300); cy = rng.uniform(-300, R + 300)
        half = f32(0.5) * P; invP = f32(1.0) / P; w = float(f32(1.0 / float(P) ** 2 * rng.uniform(0.5, 2)))
        ent += scatter(D2, S0, f32(cx), f32(cy), half, P, w, R)
        direct(ref, mips, cx, cy, half, invP, P, w, R)
    V = integrate(D2)
    nz = ref > 0
    rel = np.abs(V - ref)[nz] / ref[nz]
    print(f"{n} footprints, {ent} entries: max rel {rel.max():.3e}, 99.9% {np.quantile(rel, 0.999):.3e}, median {np.median(rel):.3e}; covered {nz.mean():.3f}")
