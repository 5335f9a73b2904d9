# TA / TD / TCP / SQ counters of the wavefront kernels (one small counter group per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export ONLY=wavefront
pass() { n=$1; shift; timeout -k 5 90 rocprofv3 --pmc "$@" -d $O/pmc_$n -o c -- python3 tools/quick_bench.py > $O/pmc_$n.log 2>&1 && echo "$n done"; }
rm -rf $O/pmc_ta1 $O/pmc_ta2 $O/pmc_tcp1 $O/pmc_tcp2 $O/pmc_tcp3 $O/pmc_sq2
pass ta1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE &&
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
pass tcp1 TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum &&
pass tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TD_TD_BUSY_sum &&
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
python3 tools/rocpd_summary.py counters $(find $O/pmc_ta1 $O/pmc_ta2 $O/pmc_tcp1 $O/pmc_tcp2 $O/pmc_tcp3 $O/pmc_sq2 -name "*.db") > $O/pmc_ta_summary.txt 2>&1
grep "wf_trace\|^#" $O/pmc_ta_summary.txt
