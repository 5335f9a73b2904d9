#!/bin/bash
# scheduling knobs re-swept under the rolling scheme (C2, C4): trace refill / stragglers, shade chain length
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe7.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-iso 2>>gpurun_out/r03_probe7.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in c2 c4; do
  run X=default
  for v in 32 52 60; do run BF_TRACE_REFILL=$v; done
  for v in 4 8 20; do run BF_TRACE_STRAGGLERS=$v; done
  for v in 2 3 4 12; do run BF_SHADE_CHAIN=$v; done
done
cat $out
