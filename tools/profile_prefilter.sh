#!/bin/bash
# rocprofv3 evidence for the k-mer prefilter (tools/prefilter_bench.py: 1000 queries x 100 000 targets, k = 5):
#   tools/profile_prefilter.sh <prefix>
# kernel trace and the HBM counters in separate runs; copy the summaries into profiles/<name>/.
set -e
prefix=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${prefix}_trace -- python3 $R/tools/prefilter_bench.py 1000 100000 5 5000 3 2000 > $R/gpurun_out/${prefix}_bench.txt 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${prefix}_pmc/$set -- python3 $R/tools/prefilter_bench.py 1000 100000 5 5000 1 2000 > /dev/null 2>&1
done
grep -v amdgpu $R/gpurun_out/${prefix}_bench.txt | tail -6
