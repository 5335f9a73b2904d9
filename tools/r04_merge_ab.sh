# wf_shade<4>: a rolling call's evicting walk and its wake launch in ONE launch (BF_ROLL_MERGE, default 1) against the two launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/merge
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/merge/pytest.log 2>&1; rc=$?; echo "pytest rc $rc $(tail -1 gpurun_out/merge/pytest.log)"
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" gpurun_out/merge/pytest.log | head; exit 1; }
one() { label=$1; cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --config $cfg --no-cpu --no-iso $EXTRA > gpurun_out/merge/${cfg}_$label.json 2> gpurun_out/merge/${cfg}_$label.err || { echo "bench $cfg $label failed"; tail -2 gpurun_out/merge/${cfg}_$label.err; return; }
  python - gpurun_out/merge/${cfg}_$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
ks="  ".join("%s %.3f (%.2f/step)" % (k["kernel"].split("<")[0].replace("bfd::","").replace("bf_render_kernel","tail"), k["ms_per_step"], k["launches_per_step"]) for k in r["kernels"])
print("%-8s %-8s %8.1f Mrays/s %7.3f ms/step serial %7.3f | %s" % (c["name"], sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_step_serial"], ks))
PY
}
for rep in 1 2; do
  for cfg in c3 c4shard c2 c5; do one two$rep $cfg BF_ROLL_MERGE=0; one one$rep $cfg X=1; done
done
