cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q > gpurun_out/r02_pytest_batch.log 2>&1; echo "batch pytest rc $?"; tail -30 gpurun_out/r02_pytest_batch.log
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_batch.py > gpurun_out/r02_pytest3.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_pytest3.log
STREAMS=1,2,3 timeout -k 10 300 python tools/c5_sweep.py > gpurun_out/r02_c5.log 2>&1; echo "c5 rc $?"; cat gpurun_out/r02_c5.log
