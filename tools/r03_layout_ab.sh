#!/bin/bash
# per-slot state layout A/B: 64-byte records (default, BF_STATE_AOS=1) vs one array per row (make variant VARIANT=soa EXTRA=-DBF_STATE_AOS=0)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_layout_ab.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu $EXTRA 2>>gpurun_out/r03_layout_ab.err | tail -1)
  echo "$CFG $EXTRA $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "Mrays/s", d["value"], "serial", d["ms_per_step_serial"], ks, "iso", d["config"].get("isolated_step_ms"))')" >> $out; }
for CFG in ${CFGS:-c2 c3 c4shard c4 c5}; do
  for rep in 1 2; do
    run LAYOUT=records
    run LAYOUT=rows BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_soa.so
  done
  EXTRA="--rolling 0" run LAYOUT=records
  EXTRA="--rolling 0" run LAYOUT=rows BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_soa.so
  EXTRA=
done
cat $out
