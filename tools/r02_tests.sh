cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_pytest5.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r02_pytest5.log
