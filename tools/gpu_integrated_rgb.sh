#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_integrated.py -q -m gpu 2>&1 | tail -3
echo "=== config 5, three tiers, integrated_px=512"; python3 tools/gpu_bench_sweep.py 5e7 mode=rgb R=2048 reorder=32 frames=4 integrated_px=512 2>&1 | grep "frame 3"
echo "=== config 5, kernel I alone (rgb_mega_variant=0) integrated_px=512"; python3 tools/gpu_bench_sweep.py 5e7 mode=rgb R=2048 reorder=32 frames=4 integrated_px=512 rgb_mega_variant=0 2>&1 | grep "frame 3"
echo "=== density 2048^2 5e7: 512"; python3 tools/gpu_bench_sweep.py 5e7 R=2048 reorder=32 frames=4 integrated_px=512 2>&1 | grep "frame 3"
echo "=== headline 256"; python3 tools/gpu_bench_sweep.py 1.25e8 reorder=50 frames=4 integrated_px=256 2>&1 | grep "frame 3"
echo "=== headline 1024"; python3 tools/gpu_bench_sweep.py 1.25e8 reorder=50 frames=4 integrated_px=1024 2>&1 | grep "frame 3"
