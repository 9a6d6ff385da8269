#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
( python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "not config5 and not 100000000" ) > gpurun_out/r3b/pytest.log 2>&1
tail -3 gpurun_out/r3b/pytest.log
tools/gpu_ab.sh "base -" 1.25e8 hcap=8 frames=6 2>&1 | tee gpurun_out/r3b/ab_hcap.log
tools/gpu_ab.sh "base -" 1.25e8 frames=6 2>&1 | tee gpurun_out/r3b/ab_full.log
tools/gpu_ab.sh "base -" 1e7 mode=weighted frames=6 2>&1 | tee gpurun_out/r3b/ab_w.log
tools/gpu_ab.sh "base -" 2e7 mode=rgb R=2048 frames=5 2>&1 | tee gpurun_out/r3b/ab_rgb.log
