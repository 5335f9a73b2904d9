cd $GRAFT_REPO_ROOT
export BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_rf.so
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/rf_pytest.log 2>&1; echo "pytest (refill variant) rc $?"; tail -6 gpurun_out/rf_pytest.log
