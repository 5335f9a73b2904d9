# round 3: 2-rank rehearsal of bench.py's N > 1 path on ONE GPU (host-staged collectives): rolling sequences per rank, the
# flush, ONE all-reduce of the [K, channels] histograms — the rank / offset / reduce logic, not RCCL over xGMI
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "c2 weak 4194304" "c4 strong 0" "c3 weak 0" "c5 weak 262144"; do
  set -- $spec
  extra=""; [ "$3" != "0" ] && extra="--paths $3"
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config $1 --scaling $2 --steps 6 --warmup 1 --no-cpu $extra > gpurun_out/r04_dist_$1_$2.json 2> gpurun_out/r04_dist_$1_$2.err; echo "dist $1 $2 rc $?"
  tail -1 gpurun_out/r04_dist_$1_$2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' ', d['n_gpus'], 'ranks', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', d['scaling'], d['config']['paths_per_gpu_per_step'], 'paths per rank, rolling', d['config']['rolling'])"
  grep -v "amdgpu.ids\|^$\|Warning\|warn" gpurun_out/r04_dist_$1_$2.err | tail -3
done
