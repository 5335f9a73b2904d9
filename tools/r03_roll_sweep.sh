#!/bin/bash
# round 3: rolling sequences — streams x iterations-per-call sweep on C2 / C3 / C4 shard (bench.py --no-cpu)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_roll_sweep.txt
: > $out
for cfg in ${CFGS:-c2 c3}; do
  for st in ${STREAMS:-1 2 4 8}; do
    for it in ${ITERS:-0 1 2 3}; do
      r=$(BF_ROLL_ITERS=$it timeout -k 10 300 python bench.py --config $cfg --steps ${STEPS:-20} --warmup 2 --no-cpu --streams $st 2>>gpurun_out/r03_roll_sweep.err | tail -1)
      echo "$cfg streams=$st roll_iters=$it $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], "frac", d["roofline"]["frac"], "tail_ms/step", d["config"].get("tail_ms_per_step"))')" >> $out
    done
  done
  r=$(timeout -k 10 300 python bench.py --config $cfg --steps ${STEPS:-20} --warmup 2 --no-cpu --rolling 0 2>>gpurun_out/r03_roll_sweep.err | tail -1)
  echo "$cfg rolling=0 streams=8 $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], "frac", d["roofline"]["frac"])')" >> $out
done
cat $out
