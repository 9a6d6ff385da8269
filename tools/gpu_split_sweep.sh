cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep "frame 4" | cut -c 1-90; }
for sp in 64 96 192 256; do run 1.25e8 mega_split=$sp; done
for sp in 64 128 512; do run 1.25e8 huge_split=$sp; done
for sp in 64 256; do run 1.25e8 mid_split=$sp; done
for b in 50 200; do run 1.25e8 stream_blocks_per_cu=$b; done
