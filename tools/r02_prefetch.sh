#!/bin/bash
# wf_shade: software prefetch of the next visit's state rows (BF_SHADE_PREFETCH = batches ahead): parity, then A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BF_SHADE_PREFETCH=2 timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r02_prefetch_pytest.log 2>&1 || { tail -30 gpurun_out/r02_prefetch_pytest.log; exit 1; }
tail -2 gpurun_out/r02_prefetch_pytest.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_prefetch.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for cfg in c2 c5; do
run "$cfg prefetch 0" -- --config $cfg
run "$cfg prefetch 1" BF_SHADE_PREFETCH=1 -- --config $cfg
run "$cfg prefetch 2" BF_SHADE_PREFETCH=2 -- --config $cfg
run "$cfg prefetch 4" BF_SHADE_PREFETCH=4 -- --config $cfg
done
run "c3 prefetch 0" -- --config c3
run "c3 prefetch 2" BF_SHADE_PREFETCH=2 -- --config c3
} > gpurun_out/r02_prefetch.log 2>&1
cat gpurun_out/r02_prefetch.log
