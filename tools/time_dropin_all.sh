#!/bin/bash
# Per-stage wall time of the reference CLI's four builds (make dropin*) on ONE generated input, two runs each, S4G_PROFILE=1:
#   tools/time_dropin_all.sh <n_queries> <n_targets> <out_dir> [extra cli args...]
# writes <out_dir>/<build>_<nq>x<nt>.txt; the verdict's configs[2]-scale run is `tools/time_dropin_all.sh 1000 570000 gpurun_out/r06_dropin -t 32`.
nq=$1; nt=$2; out=$3; shift 3
mkdir -p "$out"
export S4G_DROPIN_DATA=${S4G_DROPIN_DATA:-/tmp/s4g_dropin_data} S4G_PROFILE=1
for b in sift4g sift4g_batched sift4g_devsearch sift4g_fullscan; do
    [ -x build/dropin/$b ] || { echo "build/dropin/$b missing"; continue; }
    for run in 1 2; do
        echo "== $b run $run" >> "$out/${b}_${nq}x${nt}.txt"
        S4G_DROPIN_BIN=$b python3 tools/time_dropin.py $nq $nt "$@" >> "$out/${b}_${nq}x${nt}.txt" 2>&1
    done
    tail -8 "$out/${b}_${nq}x${nt}.txt"
done
