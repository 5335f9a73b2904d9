cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
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
run "c2 base" -- --config c2
run "c2 head prio 2" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prio2.so -- --config c2
run "c2 head prio 3" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prio3.so -- --config c2
run "c2 base again" -- --config c2
run "c2 head prio 3 again" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prio3.so -- --config c2
run "c5 base" -- --config c5 --steps 5
run "c5 head prio 3" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prio3.so -- --config c5 --steps 5
run "c3 base" -- --config c3
run "c3 head prio 3" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prio3.so -- --config c3
} > gpurun_out/r02_prio.log 2>&1
cat gpurun_out/r02_prio.log
