cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "doppler or scheduling" > gpurun_out/r02_pytest6.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_pytest6.log
export BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "c2 weak" "c4 strong" "c5 weak"; do
  set -- $spec
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config $1 --scaling $2 --steps 4 --warmup 1 --no-cpu > gpurun_out/r02_dist_$1_$2.json 2> gpurun_out/r02_dist_$1_$2.err; echo "dist $1 $2 rc $?"
  tail -c 700 gpurun_out/r02_dist_$1_$2.json | head -c 600; echo; grep -v "amdgpu.ids\|^$" gpurun_out/r02_dist_$1_$2.err | tail -3
done
unset BENCH_SHARE_DEVICE BENCH_DIST_BACKEND
timeout -k 10 300 python bench.py --config c4 --scaling strong --steps 4 --warmup 1 --no-cpu > gpurun_out/r02_c4_strong_n1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_c4_strong_n1.json')); print('c4 strong N=1', d['value'], d['ms_per_step'], d['config']['paths_per_gpu_per_step'], d['config'].get('isolated_step_ms'), d['config'].get('tail_ms_per_step'))"
