cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest9.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest9.log
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
run "c2" -- --config c2
run "c2 shade waves 4" BF_SHADE_WAVES=4 -- --config c2
run "c2 shade waves 2" BF_SHADE_WAVES=2 -- --config c2
run "c2 again" -- --config c2
run "c5" -- --config c5 --steps 5
run "c5 shade waves 2" BF_SHADE_WAVES=2 -- --config c5 --steps 5
} > gpurun_out/r02_shade.log 2>&1
cat gpurun_out/r02_shade.log
