#!/bin/bash
# rocprofv3 --kernel-trace --stats of the sub-tomogram alignment block of bench.py; keeps our kernels' rows.
#   usage: scripts/prof_sva.sh <tag>
set -e
export TMPDIR=/tmp
T=${1:-r03_sva}
rm -rf /tmp/prof_sva; mkdir -p /tmp/prof_sva gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_sva -o p --output-format csv -- python3 bench.py --workload sva --no-cpu --steps 2 > gpurun_out/${T}_bench_line.json 2> /tmp/prof_sva/err.txt || { tail -5 /tmp/prof_sva/err.txt; exit 1; }
f=$(find /tmp/prof_sva -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/${T}_kernel_stats.csv; grep "ppm::" $f >> gpurun_out/${T}_kernel_stats.csv
cut -c1-160 gpurun_out/${T}_kernel_stats.csv | head -14
