#!/bin/bash
# rocprofv3 summary of the headline workload with the option integrated_px = 256 (kernel I); run on the GPU box, then
#   python3 tools/collect_profiles.py <tag>_integrated gpurun_out/integ_prof keep
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/integ_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
A="--integrated-px 256 --headline-only --steps 5 --warmup 2"
python3 bench.py --integrated-px 256 --headline-only > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --integrated-px 256 --headline-only > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_sq -o run -- python3 bench.py $A > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_lds -o run -- python3 bench.py $A > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py $A > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o run -- python3 bench.py $A > $OUT/pmc_write.log 2>&1
tail -1 $OUT/bench.json | cut -c1-700
