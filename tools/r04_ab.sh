# A/B of library builds on one box, same run:  tools/r04_ab.sh <tag> <lib-suffix>... [-- config ...]
#   lib suffix "" = the in-tree libbeifong_hip.so, "r03" = libbeifong_hip_r03.so, ...; one bench line per (config, lib), no CPU leg
cd $GRAFT_REPO_ROOT
TAG=$1; shift
LIBS=(); CFGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=(c2 c3 c4shard c5)
O=gpurun_out/$TAG
mkdir -p $O
for cfg in "${CFGS[@]}"; do
  for rep in 1 2; do
    for lib in "${LIBS[@]}"; do
      name=${lib:-new}
      if [ "$lib" = "new" ] || [ -z "$lib" ]; then unset BF_HIP_LIB; name=new; else export BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_$lib.so; fi
      timeout -k 10 400 python bench.py --config $cfg --no-cpu $EXTRA > $O/${cfg}_${name}_$rep.json 2> $O/${cfg}_${name}_$rep.err || { echo "bench $cfg $name failed"; tail -3 $O/${cfg}_${name}_$rep.err; exit 1; }
    done
  done
done
python - "$O" <<'PY'
import json, sys, glob, os
O = sys.argv[1]
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "??", e); continue
    r = d["roofline"]
    ks = "  ".join("%s %.3f ms (%.2f)" % (k["kernel"].split("<")[0].replace("bfd::", "").replace("bf_render_kernel", "tail"), k["ms_per_step"], k["frac"]) for k in r["kernels"])
    print("%-22s %8.1f Mrays/s %7.3f ms/step serial %7.3f iso %s | %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["ms_per_step_serial"], d["config"].get("isolated_step_ms"), ks))
PY
