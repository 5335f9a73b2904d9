cd $GRAFT_REPO_ROOT
export BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_shprof.so
python tools/shade_profile.py 2>&1 | tail -16
SCENE=c5 PATHS=4194304 python tools/shade_profile.py 2>&1 | tail -16
