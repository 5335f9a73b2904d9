# round-4 final evidence run: GPU parity suite, smoke, profiles (kernel stats + PMC -> profiles/r04_pmc_traffic.json), THEN one bench
# line per config (with the CPU baseline; the lines carry `traffic` because the PMC summary is stamped with this tree's kernels)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
bash tools/profile_r04.sh > $O/profile.log 2>&1; echo "profile rc $?"
cp gpurun_out/r04prof/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
bash tools/profile_r04_streams1.sh > $O/profile_s1.log 2>&1; echo "profile s1 rc $?"
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc $?"
timeout -k 10 600 python bench.py > $O/bench_c2_run2.json 2> /dev/null
for cfg in c3 c4shard c4 c5; do
  timeout -k 10 600 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc $?"
done
timeout -k 10 600 python bench.py --config c4 --scaling strong --no-cpu > $O/bench_c4_strong_n1.json 2> /dev/null
python - <<'PY'
import json
for c in ("c2","c2_run2","c3","c4shard","c4","c5","c4_strong_n1"):
    try:
        d=json.loads(open("gpurun_out/r04final/bench_%s.json"%c).read().strip().splitlines()[-1])
    except Exception as e:
        print(c, "??", e); continue
    r=d["roofline"]
    print("%-14s %8.1f Mrays/s %7.3f ms/step serial %7.3f standalone %s frac %.3f kernel_frac %.3f (by traffic %s) cpu %s iso %s" % (c, d["value"], d["ms_per_step"], d["ms_per_step_serial"], d.get("standalone",{}).get("ms_per_step"), r["frac"], r["kernel_frac"], r["frac_by_counter_traffic"], d.get("cpu_baseline",{}).get("value"), d["config"].get("isolated_step_ms")))
    for k in r["kernels"]: print("      %-50s share %.3f %7.3f ms %5.2f launches/step frac %.3f traffic/alg %s" % (k["kernel"][:50], k["share_of_gpu_time"], k["ms_per_step"], k["launches_per_step"], k["frac"], k["traffic_over_algorithmic"]))
PY
