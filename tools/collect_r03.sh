#!/bin/bash
# copy the outputs of tools/r03_final.sh + profile_r03.sh + profile_r03_streams1.sh (gpurun_out/, scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r03final; P=gpurun_out/r03prof
for c in c2 c2_run2 c3 c4shard c4 c5 c2_standalone c3_standalone c4shard_standalone c4_standalone c4_strong_n1; do
  [ -s $F/bench_$c.json ] && tail -1 $F/bench_$c.json > profiles/r03_bench_$c.json
done
cp $F/pytest_gpu.log profiles/r03_pytest_gpu.txt
for c in c2 c3 c4shard c4 c5; do
  cp $P/${c}_kernel_stats.csv profiles/r03_${c}_kernel_stats.csv
  cp $P/${c}_streams1_kernel_stats.csv profiles/r03_${c}_streams1_kernel_stats.csv
  cp $P/${c}_pmc.txt profiles/r03_${c}_pmc.txt
  tail -1 $P/bench_${c}_under_rocprof.json > profiles/r03_${c}_bench_under_rocprof.json
  tail -1 $P/bench_${c}_streams1_under_rocprof.json > profiles/r03_${c}_streams1_bench_under_rocprof.json
done
cp $P/c2_pmc_sq_l2.txt profiles/r03_c2_pmc_sq_l2.txt
cp $P/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
python3 -c "
import sys, json; sys.path.insert(0, '.')
import bench
print('csrc now', bench.csrc_hash(), 'profiled', json.load(open('profiles/r03_pmc_traffic.json')).get('csrc_sha16'))"
