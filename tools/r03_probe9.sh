#!/bin/bash
# iterations per rolling call, fixed (BF_ROLL_ITERS) and steered (BF_ROLL_LIVE), on the final kernels
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe9.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-iso 2>>gpurun_out/r03_probe9.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in c3 c4shard c4 c2; do
  for it in 1 2 3 4; do run BF_ROLL_ITERS=$it; done
  for live in 786432 1048576 1572864 2097152 3145728; do run BF_ROLL_LIVE=$live; done
done
cat $out
