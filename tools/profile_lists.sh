#!/bin/bash
# rocprofv3 evidence for candidate-list scoring (tools/lists_bench.py: 200 queries x 3645 candidates of 50 000 targets):
#   tools/profile_lists.sh <prefix> <route>      route = wave | auto | lists:<split>
# kernel trace and each counter set in separate runs; tools/summarize_pmc.py <prefix> <name> builds profiles/<name>/.
set -e
prefix=$1; route=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${prefix}_trace -- python3 $R/tools/lists_bench.py 200 50000 3645 3 $route > $R/gpurun_out/${prefix}_bench.txt 2>&1
# (round 4: the issue-stall counters decomposed -- what a wave waits for when it cannot issue, per instruction class; a counter this
# part does not have fails its own pass only)
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_INSTS_SENDMSG SQ_ACTIVE_INST_EXP_GDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr " " "_")
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${prefix}_pmc/$n -- python3 $R/tools/lists_bench.py 200 50000 3645 1 $route > /dev/null 2>&1 || echo "counter set '$set' failed"
done
grep -v amdgpu $R/gpurun_out/${prefix}_bench.txt | tail -3
