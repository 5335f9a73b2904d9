# round-2 baseline: isolated renders of the configured sizes + tail profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest0.log 2>&1; echo "pytest rc $?" 
for cfg in "bus 1048576" "car 1048576" "multi 524288" "bus 16777216"; do
  set -- $cfg
  SCENE=$1 PATHS=$2 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
done > gpurun_out/r02_base.log 2>&1
for cfg in "bus 1048576" "car 1048576"; do
  set -- $cfg
  BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prof.so SCENE=$1 PATHS=$2 timeout -k 10 200 python tools/tail_profile.py
done > gpurun_out/r02_tailprof.log 2>&1
cat gpurun_out/r02_base.log gpurun_out/r02_tailprof.log; tail -3 gpurun_out/r02_pytest0.log
