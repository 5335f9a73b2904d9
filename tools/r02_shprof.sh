#!/bin/bash
# lane-utilisation profile of wf_shade on the C2 step (instrumented variant), parking queue on / off
cd $GRAFT_REPO_ROOT
export BF_HIP_LIB=$GRAFT_REPO_ROOT/beifong_amd/csrc/libbeifong_hip_shprof.so
mkdir -p gpurun_out
{
timeout -k 10 300 python3 tools/shade_profile.py
BF_SHADE_PARK=0 timeout -k 10 300 python3 tools/shade_profile.py
} > gpurun_out/r02_shprof2.log 2>&1
cat gpurun_out/r02_shprof2.log
