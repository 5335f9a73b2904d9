cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-10} --warmup 2 "$@" 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k))"
}
{
run "c2 base" -- --config c2
run "c2 base again" -- --config c2
run "c2 tail 262144" BF_WF_TAIL=262144 -- --config c2
run "c2 tail 524288" BF_WF_TAIL=524288 -- --config c2
run "c2 tail 1048576" BF_WF_TAIL=1048576 -- --config c2
run "c2 tail 65536" BF_WF_TAIL=65536 -- --config c2
run "c2 chain 2" BF_SHADE_CHAIN=2 -- --config c2
run "c2 chain 4" BF_SHADE_CHAIN=4 -- --config c2
run "c2 tw2" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_tw2.so -- --config c2
run "c2 streams 4" -- --config c2 --streams 4
run "c2 streams 2" -- --config c2 --streams 2
run "c2 rowjobs 16" BF_TAIL_ROWJOBS=16 -- --config c2
run "c2 rowjobs 4" BF_TAIL_ROWJOBS=4 -- --config c2
run "c3 base" -- --config c3
run "c3 tail 262144" BF_WF_TAIL=262144 -- --config c3
run "c3 tail 524288" BF_WF_TAIL=524288 -- --config c3
run "c3 tail 1048576" BF_WF_TAIL=1048576 -- --config c3
run "c3 tw2" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_tw2.so -- --config c3
run "c3 streams 6" -- --config c3 --streams 6
run "c4shard base" -- --config c4shard
run "c4shard tail 524288" BF_WF_TAIL=524288 -- --config c4shard
run "c5 base" -- --config c5 --steps 5
run "c5 streams 3" -- --config c5 --steps 5 --streams 3
run "c5 tw2" BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_tw2.so -- --config c5 --steps 5
} > gpurun_out/r02_tune.log 2>&1
cat gpurun_out/r02_tune.log
