cd $GRAFT_REPO_ROOT
for lib in "" pm; do
  if [ -n "$lib" ]; then export TOPSY_SPLAT_LIB=$GRAFT_REPO_ROOT/topsy_amd/libtopsy_splat_$lib.so; fi
  echo "=== lib [$lib]"
  for a in "1.25e8 hcap=8" "1.25e8" "1e9 reorder=400 frames=3"; do python3 tools/gpu_bench_sweep.py $a 2>&1 | grep "frame [24]" | tail -1 | cut -c 1-90; done
  tools/prof_headline_pmc.sh pm_$lib --particles-per-gpu 1.25e8 --h-cap-px 8 2>&1 | grep stream_kernel | cut -c 1-400
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
