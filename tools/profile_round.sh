set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/prof_def -o d -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $O/bench_def.json 2> $O/bench_def.err
echo "default done"
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o s1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err
echo "s1 done"
export ONLY=wavefront
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o c -- python3 tools/quick_bench.py > $O/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o c -- python3 tools/quick_bench.py > $O/pmc_write.log 2>&1
echo "write done"
timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o c -- python3 tools/quick_bench.py > $O/pmc_l2.log 2>&1
echo "l2 done"
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc_sq -o c -- python3 tools/quick_bench.py > $O/pmc_sq.log 2>&1
echo "sq done"
python3 tools/rocpd_summary.py kernels $(find $O/prof_def -name "*.db" | head -1) > $O/kernel_stats_default.csv
python3 tools/rocpd_summary.py kernels $(find $O/prof_s1 -name "*.db" | head -1) > $O/kernel_stats_streams1.csv
python3 tools/rocpd_summary.py counters $(find $O/pmc_fetch $O/pmc_write $O/pmc_l2 $O/pmc_sq -name "*.db") > $O/pmc_summary.txt
head -8 $O/kernel_stats_streams1.csv
cat $O/pmc_summary.txt
tail -1 $O/bench_s1.json | cut -c1-400
