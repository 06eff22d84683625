#!/bin/bash
# rocprofv3 evidence for the per-query call's score kernels (tools/pair_latency.py 50000 3645 16: one query x 3645 candidates per call):
#   tools/profile_pair_pk.sh <prefix>      -> gpurun_out/<prefix>_{pk1,pk0}_trace (kernel trace, packed kernel on / off) and
#                                             gpurun_out/<prefix>_pmc/<counters> (packed kernel on)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prefix=$1
S4G_PAIR_PK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${prefix}_pk1_trace -- python3 $R/tools/pair_latency.py 50000 3645 16 > $R/gpurun_out/${prefix}_pk1.txt 2>&1
S4G_PAIR_PK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${prefix}_pk0_trace -- python3 $R/tools/pair_latency.py 50000 3645 16 > $R/gpurun_out/${prefix}_pk0.txt 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | tr " " "_")
  S4G_PAIR_PK=1 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${prefix}_pmc/$n -- python3 $R/tools/pair_latency.py 50000 3645 4 > /dev/null 2>&1 || echo "counter set '$set' failed"
done
for t in pk1 pk0; do grep "sw_pair_scan" $R/gpurun_out/${prefix}_${t}_trace/runc/*_kernel_stats.csv | cut -c1-200; done
