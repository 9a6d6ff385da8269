#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r3e/pytest.log 2>&1
tail -5 gpurun_out/r3e/pytest.log
tools/profile_bench.sh > gpurun_out/r3e/profile.log 2>&1
tail -3 gpurun_out/r3e/profile.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
