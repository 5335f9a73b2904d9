cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest10.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest10.log
BF_TAIL_STAGED=1 timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest10s.log 2>&1; echo "pytest staged-everywhere rc $?"; tail -4 gpurun_out/r02_pytest10s.log
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
run "c2 staged (default)" -- --config c2
run "c2 unstaged" BF_TAIL_STAGED=0 -- --config c2
for b in 8 12 24 32; do run "c2 staged iters $b" BF_TAIL_STAGE_ITERS=$b -- --config c2; done
run "c2 staged tw2" BF_TAIL_WAVES=2 -- --config c2
run "c5 staged" -- --config c5 --steps 5
run "c5 unstaged" BF_TAIL_STAGED=0 -- --config c5 --steps 5
run "c3 default (unstaged)" -- --config c3
run "c3 staged" BF_TAIL_STAGED=1 -- --config c3
run "c4 strong staged" -- --config c4 --scaling strong
run "c4 strong unstaged" BF_TAIL_STAGED=0 -- --config c4 --scaling strong
} > gpurun_out/r02_staged.log 2>&1
cat gpurun_out/r02_staged.log
