#!/bin/bash
# lane-utilisation profile of wf_shade (instrumented variant): C2 step and a C5-like receive batch
cd $GRAFT_REPO_ROOT
export BF_HIP_LIB=$GRAFT_REPO_ROOT/beifong_amd/csrc/libbeifong_hip_shprof.so
mkdir -p gpurun_out
{
timeout -k 10 300 python3 tools/shade_profile.py
SCENE=c5 PATHS=16777216 BF_WF_POOL=4194304 timeout -k 10 300 python3 tools/shade_profile.py
} > gpurun_out/r02_shprof3.log 2>&1
cat gpurun_out/r02_shprof3.log
