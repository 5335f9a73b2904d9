#!/bin/bash
# HIP runtime hardware queues (GPU_MAX_HW_QUEUES) x streams per config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-16} --warmup 3 "$@" 2>>gpurun_out/r02_hwq.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-40s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial']))"
}
{
for cfg in ${CFGS:-c2 c3 c4shard c5}; do
for q in ${QS:-8 16}; do
for st in ${STS:-3 4 6 7 8 12}; do
  run "$cfg hwq $q streams $st" GPU_MAX_HW_QUEUES=$q -- --config $cfg --streams $st
done
done
done
} > gpurun_out/r02_hwq2.log 2>&1
cat gpurun_out/r02_hwq2.log
