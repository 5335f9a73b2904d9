#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-24} --warmup 4 "$@" 2>>gpurun_out/r02_retune.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-40s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for rep in 1 2 3; do
for cfg in c3 c4shard; do
run "$cfg tail waves 2 (rep $rep)" BF_TAIL_WAVES=2 -- --config $cfg
run "$cfg tail waves 3 (rep $rep)" BF_TAIL_WAVES=3 -- --config $cfg
done
done
run "c4 strong tail waves 2" BF_TAIL_WAVES=2 -- --config c4 --scaling strong
run "c4 strong tail waves 3" BF_TAIL_WAVES=3 -- --config c4 --scaling strong
run "c5 tail waves 2" BF_TAIL_WAVES=2 -- --config c5
run "c5 tail waves 3" BF_TAIL_WAVES=3 -- --config c5
run "c5 tail waves 2 (rep 2)" BF_TAIL_WAVES=2 -- --config c5
run "c5 tail waves 3 (rep 2)" BF_TAIL_WAVES=3 -- --config c5
} > gpurun_out/r02_retune2.log 2>&1
cat gpurun_out/r02_retune2.log
