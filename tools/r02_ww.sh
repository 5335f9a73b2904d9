cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest11.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest11.log
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
run "c3" -- --config c3
run "c4shard" -- --config c4shard
run "c4 strong" -- --config c4 --scaling strong
run "c5" -- --config c5 --steps 5
for cfg in "bus 1048576" "car 1048576" "multi 524288"; do
  set -- $cfg
  echo "== $cfg"
  SCENE=$1 PATHS=$2 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
done
echo "== megakernel bus 2^22"
SCENE=bus PATHS=4194304 ONLY=megakernel timeout -k 10 200 python tools/quick_bench.py
} > gpurun_out/r02_ww.log 2>&1
cat gpurun_out/r02_ww.log
