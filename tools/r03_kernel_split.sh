#!/bin/bash
# rocprofv3 kernel trace of a rolling bench run on ONE stream: per kernel variant (first / wake / evict / plain shade, trace, tail)
cd "$(dirname "$0")/.."
CFG=${1:-c2}
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=16 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_split_$CFG -o split -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 1 --no-cpu --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r03_split_$CFG.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_split_$CFG.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py kernels gpurun_out/r03_split_$CFG/split_results.db > gpurun_out/r03_split_${CFG}_kernel_stats.csv
head -14 gpurun_out/r03_split_${CFG}_kernel_stats.csv
