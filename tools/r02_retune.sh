#!/bin/bash
# knobs that interact with concurrency, re-swept with 16 hardware queues and 8 handles in flight
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-16} --warmup 3 "$@" 2>>gpurun_out/r02_retune.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-40s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c2 base" -- --config c2
run "c2 tail 65536" BF_WF_TAIL=65536 -- --config c2
run "c2 tail 262144" BF_WF_TAIL=262144 -- --config c2
run "c2 tail 524288" BF_WF_TAIL=524288 -- --config c2
run "c2 tail waves 2" BF_TAIL_WAVES=2 -- --config c2
run "c2 shade 2" BF_SHADE_WAVES=2 -- --config c2
run "c2 trace 4" BF_TRACE_WAVES=4 -- --config c2
run "c2 shade 2 trace 4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 -- --config c2
run "c2 pool 8M" BF_WF_POOL=8388608 -- --config c2
run "c2 rowjobs 16" BF_TAIL_ROWJOBS=16 -- --config c2
run "c2 base again" -- --config c2
run "c3 base" -- --config c3
run "c3 tail 262144" BF_WF_TAIL=262144 -- --config c3
run "c3 tail 786432" BF_WF_TAIL=786432 -- --config c3
run "c3 tail waves 3" BF_TAIL_WAVES=3 -- --config c3
run "c3 shade 2 trace 4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 -- --config c3
run "c4shard base" -- --config c4shard
run "c4shard tail waves 3" BF_TAIL_WAVES=3 -- --config c4shard
run "c5 base" -- --config c5
run "c5 tail waves 2" BF_TAIL_WAVES=2 -- --config c5
run "c5 tail 262144" BF_WF_TAIL=262144 -- --config c5
run "c5 pool 8M" BF_WF_POOL=8388608 -- --config c5
} > gpurun_out/r02_retune.log 2>&1
cat gpurun_out/r02_retune.log
