#!/bin/bash
# Occupancy / strip-shape variants of the tile-gather kernels, one line per variant (run on the GPU box):
#   tools/gpu_variants.sh            density 1.25e8, weighted 1e7 + 1.25e8, rgb 5e7 @ 2048^2
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep "frame 4"; }
for hv in 1 2 4 5 6; do run 1.25e8 huge_variant=$hv; done
for mv in 1 2 3 4; do run 1.25e8 mega_variant=$mv; done
for hv in 1 4; do run 1.25e8 mode=weighted huge_variant=$hv; done
for mv in 0 4 5; do run 1.25e8 mode=weighted mega_variant=$mv; done
for mv in 0 4 5; do run 1e7 mode=weighted mega_variant=$mv; done
for hv in 1 7; do run 5e7 mode=rgb R=2048 huge_variant=$hv; done
for mv in 3 4 2 1; do run 5e7 mode=rgb R=2048 rgb_mega_variant=$mv; done
for pm in 96 128 192 256; do run 5e7 mode=rgb R=2048 huge_variant=7 p_mega_rgb_px=$pm; done
