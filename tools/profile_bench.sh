#!/bin/bash
# Runs bench.py plain and under rocprofv3 (kernel trace + stats, then PMC passes); run on the GPU box.
# The profiled runs use --headline-only so that every kernel instance in a profile belongs to the headline workload
# (the plain run carries the extra configurations and the CPU baseline).  Usage: tools/profile_bench.sh [bench.py args]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="$@"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c 1-1500
P="--headline-only --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py $ARGS --headline-only > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py $ARGS $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o run -- python3 bench.py $ARGS $P > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o run -- python3 bench.py $ARGS $P > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_lds -o run -- python3 bench.py $ARGS $P > $OUT/pmc_lds.log 2>&1
# the bandwidth regime (a 1.25e8-particle snapshot with h capped at 8 px: kernels S and M only), its own passes
H="--particles-per-gpu 1.25e8 --h-cap-px 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hcap_trace -o run -- python3 bench.py $ARGS $H --headline-only > $OUT/hcap_trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/hcap_pmc_fetch -o run -- python3 bench.py $ARGS $H $P > $OUT/hcap_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/hcap_pmc_lds -o run -- python3 bench.py $ARGS $H $P > $OUT/hcap_pmc_lds.log 2>&1
# one REAL index-range shard of the headline snapshot (shard 3 of 8: what one of config 3's 8 GPUs renders)
S="--as-shard 8:3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard_trace -o run -- python3 bench.py $ARGS $S --headline-only > $OUT/shard_trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/shard_pmc_fetch -o run -- python3 bench.py $ARGS $S $P > $OUT/shard_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/shard_pmc_write -o run -- python3 bench.py $ARGS $S $P > $OUT/shard_pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d $OUT/shard_pmc_sq -o run -- python3 bench.py $ARGS $S $P > $OUT/shard_pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
