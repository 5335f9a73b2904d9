#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-10} --warmup 2 "$@" 2>>gpurun_out/r02_c5occ.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c5 base" -- --config c5
run "c5 shade waves 2" BF_SHADE_WAVES=2 -- --config c5
run "c5 shade waves 2 trace 4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 -- --config c5
run "c5 chain 2" BF_SHADE_CHAIN=2 -- --config c5
run "c5 chain 4" BF_SHADE_CHAIN=4 -- --config c5
run "c5 chain 6" BF_SHADE_CHAIN=6 -- --config c5
run "c5 pool 2M" BF_WF_POOL=2097152 -- --config c5
run "c5 pool 8M" BF_WF_POOL=8388608 -- --config c5
run "c5 streams 2" -- --config c5 --streams 2
run "c5 streams 1" -- --config c5 --streams 1
} > gpurun_out/r02_c5occ.log 2>&1
cat gpurun_out/r02_c5occ.log
