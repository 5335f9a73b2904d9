#!/bin/bash
# flush of a rolling sequence: hand-over threshold to the tail (BF_WF_TAIL) per config
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe6.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu 2>>gpurun_out/r03_probe6.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in c2 c3 c4shard c4; do for t in 131072 65536 32768 16384 8192; do run BF_WF_TAIL=$t; done; done
cat $out
