cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
timeout -k 10 900 python -m pytest tests/test_gpu_rolling.py -m gpu -q -x > gpurun_out/r04t/pytest_rolling.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/r04t/pytest_rolling.log
