#!/bin/bash
# does the stand-alone-render pass (C) of bench.py disturb the timed region of the small configs?
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe10.txt
: > $out
for CFG in c3 c4shard; do
  for fl in "--no-cpu" "--no-cpu --no-iso" "--no-cpu" "--no-cpu --no-iso"; do
    r=$(timeout -k 10 300 python bench.py --config $CFG $fl 2>>gpurun_out/r03_probe10.err | tail -1)
    echo "$CFG [$fl] $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "serial", d["ms_per_step_serial"])')" >> $out
  done
done
cat $out
