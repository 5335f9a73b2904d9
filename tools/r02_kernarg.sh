#!/bin/bash
# HIP runtime: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) — the kernels take ~700 bytes of arguments by value
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-20} --warmup 3 "$@" 2>>gpurun_out/r02_kernarg.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-40s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for rep in 1 2; do
for cfg in c2 c3 c4shard c5; do
run "$cfg default (rep $rep)" -- --config $cfg
run "$cfg dev kernarg (rep $rep)" HIP_FORCE_DEV_KERNARG=1 -- --config $cfg
done
done
} > gpurun_out/r02_kernarg.log 2>&1
cat gpurun_out/r02_kernarg.log
