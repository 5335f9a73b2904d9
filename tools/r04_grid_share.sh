# small pools of handles that roll side by side launch a share of the persistent grids each (bf_api.cpp: wf_setup, BF_GRID_SHARE):
# A/B on one box, BF_GRID_SHARE=1 (off) against the default 3, two runs each; 2^20 / 2^21-path steps on two handles for the threshold
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gshare
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/gshare/pytest.log 2>&1; rc=$?; echo "pytest rc $rc $(tail -1 gpurun_out/gshare/pytest.log)"
[ $rc -eq 0 ] || exit 1
one() { label=$1; cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --config $cfg --no-cpu $EXTRA > gpurun_out/gshare/${cfg}_$label.json 2> gpurun_out/gshare/${cfg}_$label.err || { echo "bench $cfg $label failed"; tail -2 gpurun_out/gshare/${cfg}_$label.err; return; }
  python - gpurun_out/gshare/${cfg}_$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
ks="  ".join("%s %.3f" % (k["kernel"].split("<")[0].replace("bfd::","").replace("bf_render_kernel","tail"), k["ms_per_step"]) for k in r["kernels"])
print("%-8s %-12s paths/step %9d %8.1f Mrays/s %7.3f ms/step serial %7.3f iso %s standalone %s | %s" % (c["name"], sys.argv[2], c["paths_per_gpu_per_step"], d["value"], d["ms_per_step"], d["ms_per_step_serial"], c.get("isolated_step_ms"), (d.get("standalone") or {}).get("ms_per_step"), ks))
PY
}
for rep in 1 2; do
  for cfg in c3 c4shard; do one off$rep $cfg BF_GRID_SHARE=1; one on$rep $cfg X=1; done
done
for rep in 1 2; do
  EXTRA="--paths 1048576" one p20_off$rep c2 BF_GRID_SHARE=1; EXTRA="--paths 1048576" one p20_on$rep c2 X=1
  EXTRA="--paths 2097152" one p21_off$rep c2 BF_GRID_SHARE=1; EXTRA="--paths 2097152" one p21_on$rep c2 BF_GRID_SMALL=8388608
  EXTRA="--paths 262144 --streams 4" one p18_off$rep c2 BF_GRID_SHARE=1; EXTRA="--paths 262144 --streams 4" one p18_on$rep c2 X=1
done
one off c5 BF_GRID_SHARE=1; one on c5 X=1
one off c2 BF_GRID_SHARE=1; one on c2 X=1
