cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04b/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04b/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
BF_TAB_CACHE=0 bash tools/r04_ab.sh r04b_notab new -- c2 c5 c4shard
bash tools/r04_ab.sh r04b_tab new -- c2 c5 c4shard
