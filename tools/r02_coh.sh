#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/coh
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/coh -o coh --output-format csv -- python3 tools/coherence_probe.py > gpurun_out/coh/run.log 2>&1
tail -5 gpurun_out/coh/run.log
python3 tools/coherence_probe_report.py gpurun_out/coh | tee gpurun_out/r02_coherence.txt
