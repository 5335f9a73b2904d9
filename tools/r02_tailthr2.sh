#!/bin/bash
# C2, 16 hardware queues: later hand-over to the tail x handles in flight
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-24} --warmup 4 "$@" 2>>gpurun_out/r02_tailthr2.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial']))"
}
{
for thr in 131072 65536 32768 16384; do
for st in 8 12 16; do
  run "c2 tail $thr streams $st" BF_WF_TAIL=$thr -- --config c2 --streams $st
done
done
} > gpurun_out/r02_tailthr2.log 2>&1
cat gpurun_out/r02_tailthr2.log
