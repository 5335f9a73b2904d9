#!/bin/bash
# what does the tail cost the pipelined step?  Needs a variant library built from a tree in which the planned branch of
# bf_api.cpp does not launch the tail (a one-line local change, see profiles/r02_no_tail_probe.txt): timing only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-12} --warmup 2 "$@" 2>>gpurun_out/r02_notail.err | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
NT=$GRAFT_REPO_ROOT/beifong_amd/csrc/libbeifong_hip_notail.so
{
run "c2 no tail (probe)" BF_HIP_LIB=$NT -- --config c2
run "c2 no tail s2 t4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 BF_HIP_LIB=$NT -- --config c2
run "c5 no tail (probe)" BF_HIP_LIB=$NT -- --config c5
run "c3 no tail (probe)" BF_HIP_LIB=$NT -- --config c3
} > gpurun_out/r02_notail.log 2>&1
cat gpurun_out/r02_notail.log
