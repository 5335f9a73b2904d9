#!/bin/bash
# handles (streams) per config with the rolling scheme and the lean kernels: is the default (c2: 2, others: 4) still right?
cd "$(dirname "$0")/.."
out=gpurun_out/r03_streams_probe.txt
: > $out
for CFG in c3 c4shard c4 c2 c5; do
  for n in 1 2 3 4 6 8; do
    r=$(timeout -k 10 300 python bench.py --config $CFG --streams $n --no-cpu --no-iso 2>>gpurun_out/r03_streams_probe.err | tail -1)
    echo "$CFG streams=$n $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "serial", d["ms_per_step_serial"])')" >> $out
  done
done
cat $out
