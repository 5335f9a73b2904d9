cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest4.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_pytest4.log
for cfg in c2 c3 c4shard c5; do
  timeout -k 10 600 python bench.py --config $cfg --steps ${STEPS:-10} --warmup 2 > gpurun_out/r02_bench_$cfg.json 2> gpurun_out/r02_bench_$cfg.err; echo "bench $cfg rc $?"
  tail -c 3000 gpurun_out/r02_bench_$cfg.json; tail -3 gpurun_out/r02_bench_$cfg.err | grep -v amdgpu.ids
done
