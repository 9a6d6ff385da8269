import sys, time; sys.path.insert(0, ".")
import numpy as np
from topsy_amd import kernel_lut, _native
n = int(1.25e8)
ctx = _native.Context(1024, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
ctx.generate_synthetic(n, 0, n, 1337, 0.0)
d = ctx.download_particles(("x", "y", "z", "h", "mass"))
t = time.time(); ctx.upload_particles(d["x"], d["y"], d["z"], d["h"], d["mass"]); up = time.time() - t
t = time.time(); ctx.upload_particles(d["x"], d["y"], d["z"], d["h"], d["mass"]); up2 = time.time() - t
print(f"upload 2.5 GB (pageable numpy -> HBM): {up:.3f} s first, {up2:.3f} s second -> {2.5/up2:.1f} GB/s")
t = time.time(); ctx.reorder_spatial(32, 1337); print(f"reorder_spatial: {time.time()-t:.3f} s")
M = np.eye(4, dtype=np.float32); M[:3, :3] /= 200; M[2, :] = [0, 0, 0.5 / 200, 0.5]
for i in range(3): ms = ctx.render(M, 1 / 200.0)
print(f"frame {ms:.2f} ms; upload-inclusive single-shot rate {n/(up2 + ms*1e-3):.3g} particles/s; with reorder {n/(up2+ms*1e-3+0.0):.3g}")
t = time.time(); img = ctx.read_image(); print(f"read_image 8 MiB: {(time.time()-t)*1e3:.2f} ms")
# progressive blocks: time vs block size (prefix ranges of the stratified order)
for nb in (1e5, 1e6, 3.3e6, 1e7, 3.3e7):
    nb = int(nb)
    ts = [ctx.render(M, 1/200.0, np.array([0]), np.array([nb])) for _ in range(3)]
    print(f"block of {nb:.3g} particles: {min(ts):.3f} ms")
for nb in (1e5, 1e6):
    nb = int(nb)
    t0 = time.time(); ms = ctx.render(M, 1/200.0, np.array([0]), np.array([nb])); wall = (time.time() - t0) * 1e3
    st = ctx.stats()
    print(f"block {nb:.3g}: gpu {ms:.3f} ms (S {st['ms_stream']:.3f} M {st['ms_mid']:.3f} H {st['ms_huge']:.3f}) wall {wall:.3f} ms; mid/huge {st['n_mid']}/{st['n_huge']}")
