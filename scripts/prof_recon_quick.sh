#!/bin/bash
# rocprofv3 --kernel-trace --stats of the reconstruction block at 200 k particles; our kernels' rows.   usage: scripts/prof_recon_quick.sh <tag>
set -e
export TMPDIR=/tmp
T=${1:-rq}
rm -rf /tmp/prof_rq; mkdir -p /tmp/prof_rq gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_rq -o p --output-format csv -- python3 bench.py --workload reconstruct --recon-particles 200000 --no-cpu --no-dropin > gpurun_out/${T}_line.json 2> /tmp/prof_rq/err.txt || { tail -5 /tmp/prof_rq/err.txt; exit 1; }
f=$(find /tmp/prof_rq -name "*kernel_stats.csv" | head -1)
grep "ppm::" $f | cut -c1-150 | head -8
