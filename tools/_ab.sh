cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --no-iso 2>>gpurun_out/_ab.err | tail -1)
  echo "$CFG $1 $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); ks={k["kernel"][5:13]:(k["ms_per_step"],k["launches_per_step"]) for k in d["roofline"]["kernels"]}; print("ms_per_step", d["ms_per_step"], "serial", d["ms_per_step_serial"], ks, d["config"]["rays_per_path"])')"; }
for CFG in $CFGS; do for rep in 1 2 3; do
  run BUILD=default
  run BUILD=$V BF_HIP_LIB=$PWD/beifong_amd/csrc/libbeifong_hip_$V.so
done; done
