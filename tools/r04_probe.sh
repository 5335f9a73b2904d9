set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04fetch
mkdir -p $O
cd $R
python3 tools/fetch_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/times.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/f -o c -- python3 tools/fetch_probe.py > $O/f.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/r -o c -- python3 tools/fetch_probe.py > $O/r.log 2>&1 || echo "RDREQ pass failed"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/h -o c -- python3 tools/fetch_probe.py > $O/h.log 2>&1 || echo "HIT pass failed"
python3 tools/fetch_probe.py --report $O | tee $O/report.txt
# streams sweep of the C2 bench (two handles were as good as eight in round 3)
for s in 2 3 4; do timeout -k 10 300 python bench.py --no-cpu --no-iso --streams $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 streams', d['config']['streams'], d['ms_per_step'], 'ms/step', d['value'], 'Mrays/s')"; done
