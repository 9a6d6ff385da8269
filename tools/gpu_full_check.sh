#!/bin/bash
# The whole -m gpu suite, the bench with its rocprofv3 profile passes (tools/profile_bench.sh) and the smoke test: what a round ends with.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full_check
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/full_check/pytest.log 2>&1
tail -5 gpurun_out/full_check/pytest.log
tools/profile_bench.sh > gpurun_out/full_check/profile.log 2>&1
tail -3 gpurun_out/full_check/profile.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
