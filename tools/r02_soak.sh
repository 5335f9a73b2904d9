cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# tools/soak_parity.py ${N:-6}"; timeout -k 10 1000 python tools/soak_parity.py ${N:-6}
echo "# tools/soak_parity.py ${NB:-4} batch"; timeout -k 10 600 python tools/soak_parity.py ${NB:-4} batch
echo "# tools/soak_parity.py ${NF:-6} fuzz"; timeout -k 10 900 python tools/soak_parity.py ${NF:-6} fuzz
echo "# tools/soak_parity.py ${NM:-3} misc"; timeout -k 10 900 python tools/soak_parity.py ${NM:-3} misc
echo "# BF_WF_POOL=262144 tools/soak_parity.py 2 misc"; BF_WF_POOL=262144 timeout -k 10 900 python tools/soak_parity.py 2 misc
} > gpurun_out/r02_soak.log 2>&1
grep -c "^ok" gpurun_out/r02_soak.log; grep -c "^FAIL" gpurun_out/r02_soak.log; grep "FAIL\|bit-exact\|FAILED\|Traceback\|Error" gpurun_out/r02_soak.log | head -20; tail -3 gpurun_out/r02_soak.log
