"""Frames/s of the in-process multi-GPU driver (topsy_amd/multigpu.py) behind one Visualizer (not a test; needs >= 2 GPUs,
or repeats device 0 with devices=0,0 to exercise the host collective on a single-GPU box).

    python3 tools/multi_gpu_visualizer_bench.py 1e9 gpus=8 frames=10
    python3 tools/multi_gpu_visualizer_bench.py 2e7 devices=0,0 check=1
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import topsy_amd
from topsy_amd.drawreason import DrawReason

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
opts = dict(kv.split("=") for kv in sys.argv[2:])
frames = int(opts.get("frames", 5)); R = int(opts.get("R", 1024))
kw = {"device_ids": [int(d) for d in opts["devices"].split(",")]} if "devices" in opts else {"n_gpus": int(opts.get("gpus", 2))}
t = time.time()
vis = topsy_amd.synthetic_on_device(n, render_resolution=R, **kw)
ctx = vis.particle_buffers.context
print(f"{n:.3g} particles on {ctx.n_gpus} contexts (devices {getattr(ctx, 'device_ids', [0])}, collective {getattr(ctx, 'collective', '-')}): set-up {time.time() - t:.2f} s")
vis.scale = 200.0
for f in range(frames):
    t = time.perf_counter()
    vis.render_sph(DrawReason.EXPORT)
    rgba = vis.get_sph_presentation_image()
    wall = (time.perf_counter() - t) * 1e3
    st = ctx.stats()
    print(f"frame {f}: wall {wall:.2f} ms (two export renders + colormap)  last block: slowest shard {st['ms_total']:.2f} ms, reduce {st.get('ms_reduce', 0.0):.3f} ms")
if int(opts.get("check", 0)):
    img = vis._sph.get_image().copy()
    vis.close()
    one = topsy_amd.synthetic_on_device(n, render_resolution=R)
    one.scale = 200.0
    ref = one._sph.get_image()
    rel = np.abs(img[..., 0] - ref[..., 0]) / np.maximum(np.abs(ref[..., 0]), 1e-300)
    print(f"max relative difference to the one-context image: {rel.max():.2e}")
    one.close()
else:
    vis.close()
