#!/bin/bash
# per-kernel split of the rolling C2 step (serial, one handle) for a few iteration counts; rocprofv3 kernel trace of one of them
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe2.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu --streams $ST 2>>gpurun_out/r03_probe2.err | tail -1)
  echo "$CFG steps=$STEPS streams=$ST $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"],k["avg_launch_ms"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks, "iso", d["config"]["isolated_step_ms"])')" >> $out; }
CFG=c2; STEPS=20; ST=1; for it in 2 3 4; do run BF_ROLL_ITERS=$it; done
cd /tmp && export TMPDIR=/tmp
BF_ROLL_ITERS=3 GPU_MAX_HW_QUEUES=16 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof_roll -o roll -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 10 --warmup 1 --no-cpu --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_roll.json 2>> $GRAFT_REPO_ROOT/gpurun_out/r03_probe2.err
cd $GRAFT_REPO_ROOT
ls gpurun_out/r03_prof_roll* | head
find gpurun_out/r03_prof_roll -name "*kernel_stats*" | head -3
cat $out
