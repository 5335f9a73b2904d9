#!/bin/bash
# tail: straggler carry-over in dense waves (BF_TAIL_CARRY): parity first, then A/B on the bench configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BF_TAIL_CARRY=8 timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 180 > gpurun_out/r02_carry_pytest.log 2>&1 || { tail -30 gpurun_out/r02_carry_pytest.log; exit 1; }
tail -3 gpurun_out/r02_carry_pytest.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_carry.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c2 carry 0" -- --config c2
run "c2 carry 4" BF_TAIL_CARRY=4 -- --config c2
run "c2 carry 8" BF_TAIL_CARRY=8 -- --config c2
run "c2 carry 16" BF_TAIL_CARRY=16 -- --config c2
run "c2 carry 24" BF_TAIL_CARRY=24 -- --config c2
run "c3 carry 0" -- --config c3
run "c3 carry 8" BF_TAIL_CARRY=8 -- --config c3
run "c3 carry 16" BF_TAIL_CARRY=16 -- --config c3
run "c4shard carry 0" -- --config c4shard
run "c4shard carry 8" BF_TAIL_CARRY=8 -- --config c4shard
run "c4shard carry 16" BF_TAIL_CARRY=16 -- --config c4shard
run "c5 carry 0" -- --config c5
run "c5 carry 8" BF_TAIL_CARRY=8 -- --config c5
run "c5 carry 16" BF_TAIL_CARRY=16 -- --config c5
} > gpurun_out/r02_carry.log 2>&1
cat gpurun_out/r02_carry.log
