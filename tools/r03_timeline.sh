# kernel timeline of a rolling sequence on ONE handle (every launch of the timed region, fills and copies included):
# where the gaps between launches are.  Output: gpurun_out/r03_timeline_<cfg>.txt
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-c3 c2}; do
  rm -rf gpurun_out/tl_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tl_$cfg -o t -- python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu --no-iso --streams 1 > gpurun_out/tl_$cfg.json 2> gpurun_out/tl_$cfg.err
  python3 - $cfg <<'PY' > gpurun_out/r03_timeline_$1$cfg.txt
import sqlite3, sys, glob
cfg = sys.argv[1]
db = glob.glob("gpurun_out/tl_%s/**/*.db" % cfg, recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name,start,end from kernels order by start"))
# the timed region = the last run of launches: take the last 6 wake launches (wf_shade<2) as the steps' anchors
idx = [i for i, r in enumerate(rows) if "wf_shade<2" in r[0] or "wf_shade<1" in r[0]]
first = idx[-6] if len(idx) >= 6 else idx[0]
t0 = rows[first][1]
busy = 0; prev_end = None; gaps = 0
for r in rows[first:]:
    n = r[0].split("(")[0].replace("void ", "").replace("bfd::", "")
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    print("%10.1f us  gap %7.1f  +%8.1f us  %s" % ((r[1] - t0) / 1e3, gap, (r[2] - r[1]) / 1e3, n[:60]))
    busy += r[2] - r[1]; gaps += max(0, r[1] - prev_end) if prev_end else 0
    prev_end = max(prev_end or 0, r[2])
print("# busy %.1f us, gaps %.1f us, span %.1f us" % (busy / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
PY
  tail -1 gpurun_out/r03_timeline_$cfg.txt
done
