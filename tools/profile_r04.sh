# round-4 profiles: for every bench config the kernel-trace stats of `bench.py --config X` and the PMC passes that give
# per-kernel HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "HBM"); for c2 also the SQ / L2 passes.
# Output: gpurun_out/r04prof/...; tools/pmc_traffic.py turns the PMC databases into profiles/r04_pmc_traffic.json.
set -e
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16      # as bench.py sets it for itself; under rocprofv3 the profiler starts HIP first, so it has to come from outside
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
cd $R
for cfg in ${CONFIGS:-c2 c3 c4shard c4 c5}; do
  steps=5; [ $cfg = c5 ] && steps=3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$cfg -o k -- python3 bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-iso > $O/bench_${cfg}_under_rocprof.json 2> $O/bench_${cfg}.err
  python3 tools/rocpd_summary.py kernels $(find $O/kt_$cfg -name "*.db" | head -1) > $O/${cfg}_kernel_stats.csv
  echo "== $cfg kernel stats"; head -6 $O/${cfg}_kernel_stats.csv
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $O/pmc_${cfg}_$ctr -o c -- python3 bench.py --config $cfg --steps 4 --warmup 0 --no-cpu --no-iso --streams 1 > $O/pmc_${cfg}_$ctr.json 2> $O/pmc_${cfg}_$ctr.err
  done
  python3 tools/rocpd_summary.py counters $(find $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE -name "*.db") > $O/${cfg}_pmc.txt
  cat $O/${cfg}_pmc.txt
done
if echo "${CONFIGS:-c2}" | grep -q c2; then
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_c2_l2 -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --no-iso --streams 1 > $O/pmc_c2_l2.json 2> $O/pmc_c2_l2.err
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU -d $O/pmc_c2_sq -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --no-iso --streams 1 > $O/pmc_c2_sq.json 2> $O/pmc_c2_sq.err
  python3 tools/rocpd_summary.py counters $(find $O/pmc_c2_l2 $O/pmc_c2_sq -name "*.db") > $O/c2_pmc_sq_l2.txt
  cat $O/c2_pmc_sq_l2.txt
fi
python3 tools/pmc_traffic.py $O > $O/r04_pmc_traffic.json
cat $O/r04_pmc_traffic.json
