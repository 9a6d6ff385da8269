#!/bin/bash
# workgroups per tile of kernels M and H2 at shard size (3.2e6 mid, 5.3e5 huge records)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
S="1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4"
for m in 32 48 64 96 192; do echo "== shard mid_split $m"; python tools/gpu_bench_sweep.py $S mid_split=$m 2>&1 | grep -E "frame [23]"; done
for h in 16 24 32 48 64 96; do echo "== shard huge_split $h"; python tools/gpu_bench_sweep.py $S huge_split=$h 2>&1 | grep -E "frame [23]"; done
for v in 2 4 5 6 7; do echo "== shard huge_variant $v"; python tools/gpu_bench_sweep.py $S huge_variant=$v 2>&1 | grep -E "frame [23]"; done
