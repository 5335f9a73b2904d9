# robustness: the GPU parity suite under every scheduling / variant knob (the results must not depend on any of them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/knobs
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/knobs/$label.log 2>&1
  echo "$label rc $? $(tail -1 gpurun_out/knobs/$label.log)"
}
run tab0 BF_TAB_CACHE=0
run split1 BF_SHADE_SPLIT=1
run chain0 BF_CHAIN_MIN=0
run chain40_sc2 BF_CHAIN_MIN=40 BF_SHADE_CHAIN=2
run iters3 BF_ROLL_ITERS=3
run live_small BF_ROLL_LIVE=65536
run join0 BF_ROLL_JOIN=0
run lean0 BF_LEAN=0
run quant BF_QUANT_BVH=1
run nowide BF_NO_WIDE_BVH=1
run tw4 BF_TRACE_WAVES=4
run tw6 BF_TRACE_WAVES=6
run sw2 BF_SHADE_WAVES=2
run tail2 BF_TAIL_WAVES=2
run pool_1m BF_WF_POOL=1048576      # (a 65 536-slot pool is smaller than the bench steps of tests/test_gpu_bench.py: they then do not roll)
run refill8 BF_TRACE_REFILL=8 BF_TRACE_STRAGGLERS=1
run refill60 BF_TRACE_REFILL=60 BF_TRACE_STRAGGLERS=40
run sync BF_WF_SYNC=1
