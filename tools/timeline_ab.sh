# timeline of one render with an alternative library (BF_HIP_LIB), for A/B experiments
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seq
export BF_HIP_LIB=$GRAFT_REPO_ROOT/$1
ONLY=wavefront timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/seq -o q -- python3 tools/quick_bench.py > gpurun_out/seq.log 2>&1
python3 tools/timeline.py $(find gpurun_out/seq -name "*.db" | head -1)
tail -2 gpurun_out/seq.log
