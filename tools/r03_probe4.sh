#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe4.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu --streams $ST $EXTRA 2>>gpurun_out/r03_probe4.err | tail -1)
  echo "$CFG steps=$STEPS streams=$ST $EXTRA $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
STEPS=20
for CFG in ${CFGS:-c2 c3 c4shard c4}; do for ST in ${STS:-2 4 8}; do run X=1; done; EXTRA="--rolling 0" ST=8 run X=1; EXTRA=; done
cat $out
