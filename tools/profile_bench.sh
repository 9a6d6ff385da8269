#!/bin/bash
# Runs bench.py plain and under rocprofv3 (kernel trace + stats, then PMC passes); run on the GPU box.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="$@"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py $ARGS --no-cpu-baseline --steps 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o run -- python3 bench.py $ARGS --no-cpu-baseline --steps 3 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o run -- python3 bench.py $ARGS --no-cpu-baseline --steps 3 > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
