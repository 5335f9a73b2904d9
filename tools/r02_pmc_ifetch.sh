#!/bin/bash
# PMC passes (counters only, no tracing): instruction-cache and scalar-cache behaviour, instruction mix of the three kernels on C2
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02pmc2
mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/icache -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --streams 1 > $O/icache.json 2> $O/icache.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/mix -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --streams 1 > $O/mix.json 2> $O/mix.err
timeout -k 10 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVES -d $O/dcache -o c -- python3 bench.py --config c2 --steps 2 --warmup 0 --no-cpu --streams 1 > $O/dcache.json 2> $O/dcache.err
python3 tools/rocpd_summary.py counters $(find $O -name "*.db") > $O/summary.txt
cat $O/summary.txt
