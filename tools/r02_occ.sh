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
run "c2 base (s3 t5)" -- --config c2
run "c2 s2 t5" BF_SHADE_WAVES=2 -- --config c2
run "c2 s2 t4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 -- --config c2
run "c2 s3 t4" BF_TRACE_WAVES=4 -- --config c2
run "c2 s2 t5 streams 4" BF_SHADE_WAVES=2 -- --config c2 --streams 4
run "c2 s2 t4 streams 4" BF_SHADE_WAVES=2 BF_TRACE_WAVES=4 -- --config c2 --streams 4
run "c2 pool 8M" BF_WF_POOL=8388608 -- --config c2
run "c2 pool 4M" BF_WF_POOL=4194304 -- --config c2
run "c2 base again" -- --config c2
} > gpurun_out/r02_occ.log 2>&1
cat gpurun_out/r02_occ.log
