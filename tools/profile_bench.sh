#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py configuration on the GPU box (run through gpurun):
#   tools/profile_bench.sh <prefix> [bench.py args...]
# writes gpurun_out/<prefix>_{trace,fetch,write,sq}/ and gpurun_out/<prefix>_bench_trace.txt; tools/summarize_profile.py turns them
# into profiles/<name>/.  The kernel trace and each --pmc counter set are separate runs (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; gpurun refuses --pmc combined with trace domains other than the kernel trace).
set -e
prefix=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${prefix}_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-scale-ref "$@" > $R/gpurun_out/${prefix}_bench_trace.txt 2> $R/gpurun_out/${prefix}_bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${prefix}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-selfcheck --no-tiers --no-prefilter --no-scale-ref --no-end-to-end "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${prefix}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-selfcheck --no-tiers --no-prefilter --no-scale-ref --no-end-to-end "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/${prefix}_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-selfcheck --no-tiers --no-prefilter --no-scale-ref --no-end-to-end "$@" > /dev/null 2>&1
ls $R/gpurun_out/${prefix}_trace/runc | head -3
tail -c 600 $R/gpurun_out/${prefix}_bench_trace.txt
