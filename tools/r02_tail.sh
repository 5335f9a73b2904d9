# tail-kernel experiment: parity tests + isolated renders + cycle breakdown, for the default and the 2-waves/SIMD tail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_pytest1.log
{
for lib in libbeifong_hip.so libbeifong_hip_tw2.so; do
 for cfg in "bus 1048576" "car 1048576" "multi 524288" "bus 16777216"; do
  set -- $cfg
  echo "== $lib $cfg"
  BF_HIP_LIB=beifong_amd/csrc/$lib SCENE=$1 PATHS=$2 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
 done
done
echo "== no wide bvh (quad only)"
BF_NO_WIDE_BVH=1 SCENE=bus PATHS=1048576 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
for rj in 4 8 16; do
  echo "== row_jobs $rj"
  BF_TAIL_ROWJOBS=$rj SCENE=bus PATHS=1048576 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
done
} > gpurun_out/r02_tail.log 2>&1
for lib in libbeifong_hip_prof.so libbeifong_hip_tw2prof.so; do
 for cfg in "bus 1048576" "car 1048576"; do
  set -- $cfg
  echo "== $lib $cfg"
  BF_HIP_LIB=beifong_amd/csrc/$lib SCENE=$1 PATHS=$2 timeout -k 10 200 python tools/tail_profile.py
 done
done > gpurun_out/r02_tailprof2.log 2>&1
cat gpurun_out/r02_tail.log gpurun_out/r02_tailprof2.log
