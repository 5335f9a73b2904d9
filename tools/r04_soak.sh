# round-4 parity soak on the final tree (lean kernels for the radar-profile cases, general ones for the fuzz / misc scenes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# tools/soak_parity.py ${N:-2}"; timeout -k 10 1000 python tools/soak_parity.py ${N:-2}
echo "# tools/soak_parity.py ${NR:-2} rolling"; timeout -k 10 1000 python tools/soak_parity.py ${NR:-2} rolling
echo "# tools/soak_parity.py ${NB:-2} batch"; timeout -k 10 600 python tools/soak_parity.py ${NB:-2} batch
echo "# tools/soak_parity.py ${NF:-3} fuzz"; timeout -k 10 900 python tools/soak_parity.py ${NF:-3} fuzz
echo "# tools/soak_parity.py ${NM:-2} misc"; timeout -k 10 900 python tools/soak_parity.py ${NM:-2} misc
} > gpurun_out/r04_soak.log 2>&1
grep -c "^ok" gpurun_out/r04_soak.log; grep -c "^FAIL" gpurun_out/r04_soak.log; grep "FAIL\|bit-exact\|FAILED\|Traceback\|Error" gpurun_out/r04_soak.log | head -20; tail -3 gpurun_out/r04_soak.log
