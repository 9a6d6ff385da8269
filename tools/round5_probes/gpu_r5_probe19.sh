#!/bin/bash
# kernel S with dynamic batches: parity first, then batch size / persistent workgroups per CU at shard size and on the whole snapshot
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
S="1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4"
for b in 4 8 16 32 64; do echo "== shard batch $b"; python tools/gpu_bench_sweep.py $S stream_batch_chunks=$b 2>&1 | grep -E "frame [23]"; done
for k in 4 6 8 12; do echo "== shard blocks_per_cu $k"; python tools/gpu_bench_sweep.py $S stream_blocks_per_cu=$k 2>&1 | grep -E "frame [23]"; done
for b in 8 16 32 64; do echo "== whole batch $b"; python tools/gpu_bench_sweep.py 1e9 reorder=32 frames=4 stream_batch_chunks=$b 2>&1 | grep -E "frame [23]"; done
echo "== 1e7 own snapshot"; python tools/gpu_bench_sweep.py 1e7 reorder=8 frames=4 2>&1 | grep -E "frame [23]"
echo "== 1e6 own snapshot"; python tools/gpu_bench_sweep.py 1e6 reorder=8 frames=4 2>&1 | grep -E "frame [23]"
