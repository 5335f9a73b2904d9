cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04h/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04h/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
BF_SHADE_SPLIT=0 bash tools/r04_ab.sh r04h_split0 new -- c2 c5 c4shard
BF_SHADE_SPLIT=1 bash tools/r04_ab.sh r04h_split1 new -- c2 c5 c4shard
BF_CHAIN_MIN=16 bash tools/r04_ab.sh r04h_cm16 new -- c2 c5 c4shard
BF_CHAIN_MIN=28 bash tools/r04_ab.sh r04h_cm28 new -- c2 c5 c4shard
BF_CHAIN_MIN=40 bash tools/r04_ab.sh r04h_cm40 new -- c2 c5 c4shard
