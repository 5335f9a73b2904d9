#!/bin/bash
# wf_shade phase scheduling (BF_SHADE_SCHED=1): parity for both settings, then A/B per config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BF_SHADE_SCHED=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r02_sched_pytest1.log 2>&1 || { tail -30 gpurun_out/r02_sched_pytest1.log; exit 1; }
tail -2 gpurun_out/r02_sched_pytest1.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r02_sched_pytest0.log 2>&1 || { tail -30 gpurun_out/r02_sched_pytest0.log; exit 1; }
tail -2 gpurun_out/r02_sched_pytest0.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_sched.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for cfg in c5 c2 c3 c4shard; do
run "$cfg sched 0" -- --config $cfg
run "$cfg sched 1" BF_SHADE_SCHED=1 -- --config $cfg
run "$cfg sched 1 chain 4" BF_SHADE_SCHED=1 BF_SHADE_CHAIN=4 -- --config $cfg
run "$cfg sched 1 chain 16" BF_SHADE_SCHED=1 BF_SHADE_CHAIN=16 -- --config $cfg
done
} > gpurun_out/r02_sched.log 2>&1
cat gpurun_out/r02_sched.log
