#!/bin/bash
# does the filtered branch of film_put (reconstruction filters wider than a pixel) cost the box-filter scenes anything?
# default build against libbeifong_hip_nowide.so = the library of the commit before the filters (the run-time-branch form
# had a macro to compile the branch out; as shipped the box-filter kernels do not contain it at all: kWide variants)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_wide_ab.txt
: > $out
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-iso 2>>gpurun_out/r03_wide_ab.err | tail -1)
  echo "$CFG $* $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "serial", d["ms_per_step_serial"], ks)')" >> $out; }
for CFG in ${CFGS:-c2 c5 c4}; do
  for rep in 1 2 3; do
    run BUILD=default
    run BUILD=nowide BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_nowide.so
  done
done
cat $out
