cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
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
for b in 64 128 256 384; do run "c2 tail blocks $b" BF_TAIL_BLOCKS=$b -- --config c2; done
run "c2 base again" -- --config c2
run "c2 tail blocks 128 tw2" BF_TAIL_BLOCKS=128 BF_TAIL_WAVES=2 -- --config c2
run "c2 tail blocks 256 tw2" BF_TAIL_BLOCKS=256 BF_TAIL_WAVES=2 -- --config c2
run "c2 blocks 256 tail 262144" BF_TAIL_BLOCKS=256 BF_WF_TAIL=262144 -- --config c2
run "c2 blocks 128 tail 262144" BF_TAIL_BLOCKS=128 BF_WF_TAIL=262144 -- --config c2
run "c5 base" -- --config c5 --steps 5
run "c5 tail blocks 128" BF_TAIL_BLOCKS=128 -- --config c5 --steps 5
run "c5 tail blocks 256" BF_TAIL_BLOCKS=256 -- --config c5 --steps 5
run "c3 base" -- --config c3
run "c3 tail blocks 256" BF_TAIL_BLOCKS=256 -- --config c3
run "c3 tail blocks 512" BF_TAIL_BLOCKS=512 -- --config c3
} > gpurun_out/r02_tailgrid.log 2>&1
cat gpurun_out/r02_tailgrid.log
