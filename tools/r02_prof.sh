cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in ${LIBS:-libbeifong_hip_prof.so}; do
 for cfg in "bus 1048576" "car 1048576"; do
  set -- $cfg
  echo "== $lib $cfg"
  BF_HIP_LIB=beifong_amd/csrc/$lib SCENE=$1 PATHS=$2 timeout -k 10 200 python tools/tail_profile.py
 done
done > gpurun_out/r02_tailprof3.log 2>&1
cat gpurun_out/r02_tailprof3.log
