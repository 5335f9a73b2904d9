#!/bin/bash
# co-residency of wf_shade and wf_trace: register budgets / grids that leave room for the other kernel (rolling, C2 / C4)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe5.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --streams $ST 2>>gpurun_out/r03_probe5.err | tail -1)
  echo "$CFG streams=$ST $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in c2 c4; do for ST in 2 3 4; do
  run BF_SHADE_WAVES=3 BF_TRACE_WAVES=5
  run BF_SHADE_WAVES=2 BF_TRACE_WAVES=5
  run BF_SHADE_WAVES=2 BF_TRACE_WAVES=4
  run BF_SHADE_WAVES=3 BF_TRACE_WAVES=4
done; done
cat $out
