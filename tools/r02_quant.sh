cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest8.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest8.log
run() {
  label=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  out=$(env $envs timeout -k 10 300 python bench.py --no-cpu --steps ${STEPS:-10} --warmup 2 "$@" 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k={x['kernel'].split('<')[0].split('::')[-1]:x['ms_per_step'] for x in r['kernels']}
print('%-34s %8.1f Mrays/s  %7.3f ms/step  serial %7.3f | %s | nodes/ray %.2f' % ('$label', d['value'], d['ms_per_step'], d['ms_per_step_serial'], k, r['nodes_per_ray']))"
}
{
run "c2 quant" -- --config c2
run "c2 fp32 nodes" BF_NO_QUANT_BVH=1 -- --config c2
run "c2 quant w6" BF_TRACE_WAVES=6 -- --config c2
run "c2 quant again" -- --config c2
run "c2 fp32 again" BF_NO_QUANT_BVH=1 -- --config c2
run "c3 quant" -- --config c3
run "c3 fp32" BF_NO_QUANT_BVH=1 -- --config c3
run "c5 quant" -- --config c5 --steps 5
run "c5 fp32" BF_NO_QUANT_BVH=1 -- --config c5 --steps 5
run "c4 strong quant" -- --config c4 --scaling strong
run "c4 strong fp32" BF_NO_QUANT_BVH=1 -- --config c4 --scaling strong
} > gpurun_out/r02_quant.log 2>&1
cat gpurun_out/r02_quant.log
