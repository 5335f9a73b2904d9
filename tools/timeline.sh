# per-launch timeline of the last render of tools/quick_bench.py (ONLY=wavefront) -> gpurun_out/seq/q_results.db
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seq
ONLY=wavefront timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/seq -o q -- python3 tools/quick_bench.py > gpurun_out/seq.log 2>&1
