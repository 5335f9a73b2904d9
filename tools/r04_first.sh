# round 4, first GPU call: parity suite on the new build, then A/B against round 3's library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04a/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04a/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_ab.sh r04a_ab r03 new -- c2 c5 c4shard c3
