#!/bin/bash
# rocprofv3 summary of BASELINE config 5 (5e7 star particles, rgb, 2048^2); run on the GPU box, then
#   python3 tools/collect_profiles.py <tag> picks up gpurun_out/bench_prof (headline) -- this one is condensed by the snippet below.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rgb_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
A="--mode rgb --resolution 2048 --particles-per-gpu 5e7 --headline-only --steps 5 --warmup 2"
python3 bench.py $A > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py $A > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o run -- python3 bench.py $A > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py $A > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o run -- python3 bench.py $A > $OUT/pmc_write.log 2>&1
tail -1 $OUT/bench.json | cut -c1-600
