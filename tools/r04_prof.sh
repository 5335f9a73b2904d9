cd $GRAFT_REPO_ROOT
export BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_shprof.so
mkdir -p gpurun_out/r04prof
timeout -k 10 300 python tools/shade_profile.py > gpurun_out/r04prof/shade_c2.txt 2>&1; cat gpurun_out/r04prof/shade_c2.txt
SCENE=c5 PATHS=$((1<<22)) BF_WF_POOL=$((1<<20)) timeout -k 10 300 python tools/shade_profile.py > gpurun_out/r04prof/shade_c5.txt 2>&1; cat gpurun_out/r04prof/shade_c5.txt
