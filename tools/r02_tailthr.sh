#!/bin/bash
# C2 pipelined: later hand-over to the tail (fewer, longer wavefront iterations) x streams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_tailthr.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c2 base" -- --config c2
for thr in 65536 32768 16384 8192; do
  for st in 3 4 6; do
    run "c2 tail $thr streams $st" BF_WF_TAIL=$thr -- --config c2 --streams $st
  done
done
run "c2 base streams 4" -- --config c2 --streams 4
run "c2 base streams 6" -- --config c2 --streams 6
} > gpurun_out/r02_tailthr.log 2>&1
cat gpurun_out/r02_tailthr.log
