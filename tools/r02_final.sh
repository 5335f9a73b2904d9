# round-2 final evidence run: GPU parity suite, smoke, one bench line per config (with CPU baseline), isolated renders
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc $?"
timeout -k 10 600 python bench.py > $O/bench_c2_run2.json 2> /dev/null
for cfg in c3 c4shard c5; do
  timeout -k 10 600 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc $?"
done
timeout -k 10 600 python bench.py --config c4 --scaling strong > $O/bench_c4_strong_n1.json 2> /dev/null
for cfg in "bus 1048576" "car 1048576" "multi 524288" "bus 16777216"; do
  set -- $cfg
  echo "== $cfg"
  SCENE=$1 PATHS=$2 ONLY=wavefront SPLIT=1 timeout -k 10 200 python tools/quick_bench.py
done > $O/isolated_renders.log 2>&1
STREAMS=1,2,3 PER_PULSE=1 timeout -k 10 300 python tools/c5_sweep.py > $O/c5_sweep.log 2>&1
SPEED=0.5 STREAMS=3 timeout -k 10 300 python tools/c5_sweep.py > $O/c5_sweep_0p5ms.log 2>&1
python - <<'PY'
import json
for c in ("c2","c2_run2","c3","c4shard","c5","c4_strong_n1"):
    try:
        d=json.loads(open("gpurun_out/r02final/bench_%s.json"%c).read().strip().splitlines()[-1])
    except Exception as e:
        print(c, "??", e); continue
    r=d["roofline"]
    print("%-12s %8.1f Mrays/s %7.3f ms/step serial %7.3f frac %.3f cpu %s" % (c, d["value"], d["ms_per_step"], d["ms_per_step_serial"], r["frac"], d.get("cpu_baseline",{}).get("value")))
    for k in r["kernels"]: print("      %-46s share %.3f %7.3f ms frac %.3f traffic/alg %s" % (k["kernel"][:46], k["share_of_gpu_time"], k["ms_per_step"], k["frac"], k["traffic_over_algorithmic"]))
PY
cat $O/isolated_renders.log | grep -v "^W2\|^E2"; grep "streams=\|per-pulse\|Doppler" $O/c5_sweep.log $O/c5_sweep_0p5ms.log | cut -c1-300
