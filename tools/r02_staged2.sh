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
run "c2 unstaged s3" BF_TAIL_STAGED=0 -- --config c2
run "c2 unstaged s6" BF_TAIL_STAGED=0 -- --config c2 --streams 6
run "c2 staged s6" BF_TAIL_STAGED=1 -- --config c2 --streams 6
run "c2 staged s8" BF_TAIL_STAGED=1 -- --config c2 --streams 8
run "c2 staged32 s6" BF_TAIL_STAGED=1 BF_TAIL_STAGE_ITERS=32 -- --config c2 --streams 6
run "c2 staged64 s4" BF_TAIL_STAGED=1 BF_TAIL_STAGE_ITERS=64 -- --config c2 --streams 4
run "c2 staged64 s6" BF_TAIL_STAGED=1 BF_TAIL_STAGE_ITERS=64 -- --config c2 --streams 6
} > gpurun_out/r02_staged2.log 2>&1
cat gpurun_out/r02_staged2.log
