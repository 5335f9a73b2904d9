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
run "c2 chain_min 1" -- --config c2
for m in 8 16 24 32 48; do run "c2 chain_min $m" BF_SHADE_CHAIN_MIN=$m -- --config c2; done
run "c2 chain_min 16 chain 4" BF_SHADE_CHAIN_MIN=16 BF_SHADE_CHAIN=4 -- --config c2
run "c2 chain_min 24 chain 5" BF_SHADE_CHAIN_MIN=24 BF_SHADE_CHAIN=5 -- --config c2
run "c5 chain_min 1" -- --config c5 --steps 5
run "c5 chain_min 16" BF_SHADE_CHAIN_MIN=16 -- --config c5 --steps 5
run "c5 chain_min 32" BF_SHADE_CHAIN_MIN=32 -- --config c5 --steps 5
} > gpurun_out/r02_chain.log 2>&1
cat gpurun_out/r02_chain.log
