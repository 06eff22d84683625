#!/bin/bash
# Keeps the GPU of the box busy from a SECOND process tree while something else (the test suite) runs on it: two emulated ranks of
# bench.py (S4G_BENCH_SHARE_GPU=1, gloo) on a mid-sized workload, again and again, until the file given as $1 appears.
#   tools/gpu_hammer.sh <stop-file> [log]
# GPUTEST_r05 died under exactly this condition (two ranks time-slicing one device tripped a busy-fraction threshold in a contract test).
stop=$1; log=${2:-/dev/null}
n=0
while [ ! -e "$stop" ]; do
    S4G_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --queries 256 --targets 100000 --no-cpu-baseline --no-selfcheck --no-prefilter >> "$log" 2>&1
    n=$((n + 1))
done
echo "gpu_hammer: $n bench runs beside the suite" >> "$log"
