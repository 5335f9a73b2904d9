#!/bin/bash
# adaptive iteration count (BF_ROLL_LIVE) x streams on every single-GPU config
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe3.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu --streams $ST 2>>gpurun_out/r03_probe3.err | tail -1)
  echo "$CFG steps=$STEPS streams=$ST $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
STEPS=20
for CFG in c2 c3 c4shard c4; do for ST in 1 2 3 4 6; do for live in ${LIVES:-524288 1048576 2097152}; do run BF_ROLL_LIVE=$live; done; done; done
cat $out
