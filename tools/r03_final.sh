# round-3 final evidence run: GPU parity suite, smoke, one bench line per config (with CPU baseline), the stand-alone
# scheme next to it (--rolling 0), isolated renders
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc $?"
timeout -k 10 600 python bench.py > $O/bench_c2_run2.json 2> /dev/null
for cfg in c3 c4shard c4 c5; do
  timeout -k 10 600 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc $?"
done
for cfg in c2 c3 c4shard c4; do
  timeout -k 10 600 python bench.py --config $cfg --rolling 0 --no-cpu > $O/bench_${cfg}_standalone.json 2> /dev/null
done
timeout -k 10 600 python bench.py --config c4 --scaling strong --no-cpu > $O/bench_c4_strong_n1.json 2> /dev/null
python - <<'PY'
import json
for c in ("c2","c2_run2","c3","c4shard","c4","c5","c2_standalone","c3_standalone","c4shard_standalone","c4_standalone","c4_strong_n1"):
    try:
        d=json.loads(open("gpurun_out/r03final/bench_%s.json"%c).read().strip().splitlines()[-1])
    except Exception as e:
        print(c, "??", e); continue
    r=d["roofline"]
    print("%-20s %8.1f Mrays/s %7.3f ms/step serial %7.3f frac %.3f (with LDS-served nodes %.3f, serial %.3f) cpu %s iso %s" % (c, d["value"], d["ms_per_step"], d["ms_per_step_serial"], r["frac"], r["frac_with_lds_served_nodes"], r["frac_serial"], d.get("cpu_baseline",{}).get("value"), d["config"].get("isolated_step_ms")))
    for k in r["kernels"]: print("      %-50s share %.3f %7.3f ms %5.2f launches/step frac %.3f" % (k["kernel"][:50], k["share_of_gpu_time"], k["ms_per_step"], k["launches_per_step"], k["frac"]))
PY
