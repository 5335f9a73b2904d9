#!/bin/bash
# small renders (C3, C4 shard): more streams x tail register budget, pipelined throughput
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-24} --warmup 4 "$@" 2>>gpurun_out/r02_smallstreams.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for cfg in c3 c4shard; do
for st in 3 4 6 8 12; do
  run "$cfg streams $st" -- --config $cfg --streams $st
  run "$cfg streams $st tail 3 waves" BF_TAIL_WAVES=3 -- --config $cfg --streams $st
done
done
} > gpurun_out/r02_smallstreams.log 2>&1
cat gpurun_out/r02_smallstreams.log
