#!/bin/bash
# triangle records: 64 bytes (default) vs 48 bytes (make variant VARIANT=tri48 EXTRA=-DBF_TRI_STRIDE=3)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_ab128.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-iso 2>>gpurun_out/r03_ab128.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in ${CFGS:-c2 c3 c4 c5}; do for rep in 1 2; do
  run REC=64
  run REC=128 BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_ab128.so
done; done
cat $out
