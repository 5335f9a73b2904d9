#!/bin/bash
# tail of a STAND-ALONE render: waves per batch (BF_TAIL_SHARE) — isolated latency vs the pipelined stand-alone scheme
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe8.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --rolling 0 2>>gpurun_out/r03_probe8.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("pipelined ms_per_step", d["ms_per_step"], "serial", d["ms_per_step_serial"], "isolated", d["config"]["isolated_step_ms"], "isolated tail", d["config"]["isolated_tail_ms"])')" >> $out; }
for CFG in c2 c3 c4shard c4; do for sh in 1 2 4; do run BF_TAIL_SHARE=$sh; done; done
cat $out
