#!/bin/bash
# rolling sequences: how much of the step is the flush (amortised over more steps) and what the iteration count does
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe1.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu --streams $ST 2>>gpurun_out/r03_probe1.err | tail -1)
  echo "$CFG steps=$STEPS streams=$ST $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
CFG=c2; for STEPS in 20 60; do for ST in 2 4; do for it in 3 4 5; do run BF_ROLL_ITERS=$it; done; done; done
CFG=c3; for STEPS in 20 100; do for ST in 4 8; do for it in 1 2; do run BF_ROLL_ITERS=$it; done; done; done
cat $out
