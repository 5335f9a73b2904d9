#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BF_HIP_LIB=$GRAFT_REPO_ROOT/beifong_amd/csrc/libbeifong_hip_prof.so
{
PATHS=16777216 timeout -k 10 300 python3 tools/tail_profile.py
PATHS=1048576 timeout -k 10 300 python3 tools/tail_profile.py
} > gpurun_out/r02_tailprof_c2.log 2>&1
cat gpurun_out/r02_tailprof_c2.log
