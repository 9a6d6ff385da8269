"""Instruction budget of kernel H2 (not a test): the TSP_H2_DEBUG build counts (footprint, strip) pairs, covered pixel rows, row
groups with a covered row and texel-row changes; the product build gives the fragment count and the kernel time.
  tools/build_variant.sh h2dbg -DTSP_H2_DEBUG && python tools/gpu_h2_budget.py [n] [ntotal] [first]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from topsy_amd import kernel_lut, _native
    n, ntotal, first = int(float(sys.argv[2])), int(float(sys.argv[3])), int(float(sys.argv[4]))
    ctx = _native.Context(1024, 2); ctx.set_kernel_mips(kernel_lut.kernel_mips())
    ctx.generate_synthetic(ntotal, first, n, 1337, 0.0, with_quantity=False, with_rgb=False)
    ctx.reorder_spatial(8 if n < 2.5e8 else 32, 1337)
    M = np.eye(4, dtype=np.float32); M[:3, :3] /= 200.0; M[2, :] = [0, 0, 0.5 / 200.0, 0.5]
    for _ in range(3): ms = ctx.render(M, 1.0 / 200.0, mode=_native.MODE_WEIGHTED)
    st = ctx.stats(); out = {"ms_huge": st["ms_huge"], "n_huge": st["n_huge"]}
    ctx.set_option("count_fragments", 1); ctx.render(M, 1.0 / 200.0, mode=_native.MODE_WEIGHTED); st = ctx.stats()
    out.update({k: st[k] for k in ("n_fragments_stream", "n_fragments_mid", "n_fragments_huge", "n_fragments_mega")})
    print("RESULT " + json.dumps(out)); sys.exit(0)
args = [sys.argv[i] if len(sys.argv) > i else d for i, d in ((1, "1.25e8"), (2, "1e9"), (3, "3.75e8"))]
def run(lib):
    env = dict(os.environ); 
    if lib: env["TOPSY_SPLAT_LIB"] = lib
    o = subprocess.run([sys.executable, __file__, "--child"] + args, env=env, capture_output=True, text=True).stdout
    return json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:])
prod = run(None); dbg = run(os.path.join(ROOT, "topsy_amd", "libtopsy_splat_h2dbg.so"))
pairs = dbg["n_fragments_stream"] - prod["n_fragments_stream"]; rows = dbg["n_fragments_mid"] - prod["n_fragments_mid"]
changes = dbg["n_fragments_mega"] & 0xffffffff; groups = dbg["n_fragments_mega"] >> 32
frags = prod["n_fragments_huge"]
print(f"H2 {prod['ms_huge']:.3f} ms, {prod['n_huge']} records, {frags:.4g} fragments")
print(f"pairs {pairs:.4g} ({pairs / prod['n_huge']:.1f} per record), covered rows {rows:.4g} ({rows / pairs:.1f} per pair), "
      f"row groups {groups:.4g} ({groups / pairs:.2f} per pair), texel-row changes {changes:.4g} ({changes / pairs:.2f} per pair)")
print(f"lane slots of the row FMAs: {groups * 4 * 64:.4g} = {groups * 4 * 64 / frags:.2f} x fragments; FMA wave-instructions {groups * 8:.4g}")
