cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04l/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04l/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_ab.sh r04l_ab new -- c2 c5 c4shard c3 c4
