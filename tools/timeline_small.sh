# per-launch timeline of a SMALL render (PATHS, default 2^20) of tools/quick_bench.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seq
export PATHS=${PATHS:-1048576}
ONLY=wavefront SPLIT=1 timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/seq -o q -- python3 tools/quick_bench.py > gpurun_out/seq.log 2>&1
python3 tools/timeline.py $(find gpurun_out/seq -name "*.db" | head -1)
grep -v "^W2026\|^E2026" gpurun_out/seq.log | tail -3
