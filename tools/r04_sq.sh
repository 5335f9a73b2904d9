# SQ counters of the C2 bench kernels (wait share, VALU lane utilisation) + an A/B of chain_min without the split
set -e
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04sq
mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU -d $O/pmc_c2_sq -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --no-iso --streams 1 > $O/pmc_c2_sq.json 2> $O/pmc_c2_sq.err
python3 tools/rocpd_summary.py counters $(find $O/pmc_c2_sq -name "*.db") > $O/c2_pmc_sq.txt
cat $O/c2_pmc_sq.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/pmc_c2_sq2 -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --no-iso --streams 1 > $O/pmc_c2_sq2.json 2> $O/pmc_c2_sq2.err || true
python3 tools/rocpd_summary.py counters $(find $O/pmc_c2_sq2 -name "*.db") > $O/c2_pmc_sq2.txt || true
cat $O/c2_pmc_sq2.txt || true
BF_CHAIN_MIN=16 bash tools/r04_ab.sh r04sq_cm16 new -- c5 c2
bash tools/r04_ab.sh r04sq_cm0 new -- c5 c2
