#!/bin/bash
# wf_shade parking queue: parity first, then A/B on the bench configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zoo_multi_emitter or small_pool" > gpurun_out/r02_park_pytest0.log 2>&1 || { tail -30 gpurun_out/r02_park_pytest0.log; exit 1; }
tail -2 gpurun_out/r02_park_pytest0.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 180 > gpurun_out/r02_park_pytest.log 2>&1 || { tail -30 gpurun_out/r02_park_pytest.log; exit 1; }
tail -3 gpurun_out/r02_park_pytest.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c2 park" -- --config c2
run "c2 park off" BF_SHADE_PARK=0 -- --config c2
run "c2 park cap 80" BF_SHADE_PARK=80 -- --config c2
run "c3 park" -- --config c3
run "c3 park off" BF_SHADE_PARK=0 -- --config c3
run "c4shard park" -- --config c4shard
run "c4shard park off" BF_SHADE_PARK=0 -- --config c4shard
run "c5 park" -- --config c5
run "c5 park off" BF_SHADE_PARK=0 -- --config c5
run "c2 park again" -- --config c2
} > gpurun_out/r02_park.log 2>&1
cat gpurun_out/r02_park.log
