# kernel-trace stats of bench.py with --streams 1 (every launch alone on the GPU): the average durations that must agree
# with roofline.kernels[].avg_launch_ms of the same run's JSON line
set -e
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16      # as bench.py sets it for itself; under rocprofv3 the profiler starts HIP first, so it has to come from outside
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
cd $R
for cfg in ${CONFIGS:-c2 c3 c4shard c4 c5}; do
  steps=5; [ $cfg = c5 ] && steps=3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt1_$cfg -o k -- python3 bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-iso --streams 1 > $O/bench_${cfg}_streams1_under_rocprof.json 2> $O/bench_${cfg}_s1.err
  python3 tools/rocpd_summary.py kernels $(find $O/kt1_$cfg -name "*.db" | head -1) > $O/${cfg}_streams1_kernel_stats.csv
  echo "== $cfg"; head -7 $O/${cfg}_streams1_kernel_stats.csv
  tail -1 $O/bench_${cfg}_streams1_under_rocprof.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in d['roofline']['kernels']: print('   bench.py:', k['kernel'][:44], 'avg_launch_ms', k['avg_launch_ms'], 'launches/step', k['launches_per_step'])"
done
