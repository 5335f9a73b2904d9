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
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s rays/path %.3f' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k, d['config']['rays_per_path']))"
}
{
run "c2" -- --config c2
run "c2 max_depth 16" BENCH_MAX_DEPTH=16 -- --config c2
run "c2 max_depth 12" BENCH_MAX_DEPTH=12 -- --config c2
run "c2 max_depth 8" BENCH_MAX_DEPTH=8 -- --config c2
run "c2 again" -- --config c2
} > gpurun_out/r02_probe.log 2>&1
cat gpurun_out/r02_probe.log
