#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-10} --warmup 2 "$@" 2>>gpurun_out/r02_chain.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for rep in 1 2; do
for ch in 3 4 6 8 12; do
  run "c5 chain $ch (rep $rep)" BF_SHADE_CHAIN=$ch -- --config c5
done
done
for ch in 3 4 6 8 12; do
  run "c2 chain $ch" BF_SHADE_CHAIN=$ch -- --config c2
done
for ch in 3 6 12; do
  run "c3 chain $ch" BF_SHADE_CHAIN=$ch -- --config c3
  run "c4shard chain $ch" BF_SHADE_CHAIN=$ch -- --config c4shard
done
} > gpurun_out/r02_chain.log 2>&1
cat gpurun_out/r02_chain.log
