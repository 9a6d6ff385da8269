#!/bin/bash
# The N > 1 launcher path of bench.py on a 1-GPU box: two ranks under torch.distributed.run share device 0, the shard images are summed
# on the host (RCCL refuses two ranks on one device).  Exercises the reduce self-test, the reduce_check against the whole snapshot, the rendezvous, the shard arithmetic, barriers, max-over-ranks
# timing, per-rank kernel times, the whole-snapshot comparison on rank 0 and the JSON line; the line is marked "dry_run".
cd $GRAFT_REPO_ROOT
N=${1:-2}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N \
    --steps 5 --warmup 2 --shared-device-dry-run > gpurun_out/bench_dryrun_$N.json 2> gpurun_out/bench_dryrun_$N.err
tail -c 1200 gpurun_out/bench_dryrun_$N.json
