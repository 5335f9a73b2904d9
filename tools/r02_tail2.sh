cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_pytest2.log
{
for rl in 2 1 0; do
 for cfg in "bus 1048576" "car 1048576" "multi 524288"; do
  set -- $cfg
  echo "== rows_log $rl $cfg"
  BF_WIDE_ROWS_LOG=$rl SCENE=$1 PATHS=$2 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
 done
done
for cfg in "bus 1048576" "car 1048576"; do
  set -- $cfg
  echo "== prof $cfg"
  BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prof.so SCENE=$1 PATHS=$2 timeout -k 10 200 python tools/tail_profile.py
done
} > gpurun_out/r02_tail2.log 2>&1
cat gpurun_out/r02_tail2.log
