#!/bin/bash
# copy the outputs of tools/r04_final.sh + profile_r04.sh + profile_r04_streams1.sh (gpurun_out/, scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r04final; P=gpurun_out/r04prof
for c in c2 c2_run2 c3 c4shard c4 c5 c2_standalone c3_standalone c4shard_standalone c4_standalone c4_strong_n1; do
  [ -s $F/bench_$c.json ] && tail -1 $F/bench_$c.json > profiles/r04_bench_$c.json
done
cp $F/pytest_gpu.log profiles/r04_pytest_gpu.txt
for c in c2 c3 c4shard c4 c5; do
  cp $P/${c}_kernel_stats.csv profiles/r04_${c}_kernel_stats.csv
  cp $P/${c}_streams1_kernel_stats.csv profiles/r04_${c}_streams1_kernel_stats.csv
  cp $P/${c}_pmc.txt profiles/r04_${c}_pmc.txt
  tail -1 $P/bench_${c}_under_rocprof.json > profiles/r04_${c}_bench_under_rocprof.json
  tail -1 $P/bench_${c}_streams1_under_rocprof.json > profiles/r04_${c}_streams1_bench_under_rocprof.json
done
cp $P/c2_pmc_sq_l2.txt profiles/r04_c2_pmc_sq_l2.txt
cp $P/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
python3 -c "
import sys, json; sys.path.insert(0, '.')
import bench
print('csrc now', bench.csrc_hash(), 'profiled', json.load(open('profiles/r04_pmc_traffic.json')).get('csrc_sha16'))"
