#!/bin/bash
# tail: dense waves on the sixteen-wide tree, one lane per ray (BF_TAIL_WIDE_LANE): parity first, then A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 180 > gpurun_out/r02_widelane_pytest.log 2>&1 || { tail -30 gpurun_out/r02_widelane_pytest.log; exit 1; }
tail -3 gpurun_out/r02_widelane_pytest.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_widelane.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
for cfg in c2 c3 c4shard c5; do
run "$cfg wide lane" -- --config $cfg
run "$cfg four-wide (off)" BF_TAIL_WIDE_LANE=0 -- --config $cfg
done
run "c2 wide lane, tail 2 waves" BF_TAIL_WAVES=2 -- --config c2
run "c2 off, tail 2 waves" BF_TAIL_WAVES=2 BF_TAIL_WIDE_LANE=0 -- --config c2
run "c2 wide lane again" -- --config c2
} > gpurun_out/r02_widelane.log 2>&1
cat gpurun_out/r02_widelane.log
