#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (both workloads); keeps our kernels' rows and the JSON line.
#   usage: scripts/prof_r05.sh <tag> [bench args]
set -e
export TMPDIR=/tmp
T=${1:-r05_x}; shift || true
rm -rf /tmp/prof_r05; mkdir -p /tmp/prof_r05 gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_r05 -o p --output-format csv -- python3 bench.py --no-cpu --no-dropin "$@" > gpurun_out/${T}_bench_line.json 2> /tmp/prof_r05/err.txt || { tail -5 /tmp/prof_r05/err.txt; exit 1; }
f=$(find /tmp/prof_r05 -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/${T}_kernel_stats.csv; grep "ppm::" $f >> gpurun_out/${T}_kernel_stats.csv
cut -c1-130 gpurun_out/${T}_kernel_stats.csv | head -8
tail -1 gpurun_out/${T}_bench_line.json | cut -c1-3000
