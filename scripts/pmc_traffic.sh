#!/bin/bash
# HBM-side traffic of the kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass,
# MI355X_MICROARCH.md), kernel-trace only.  Usage: pmc_traffic.sh <workload: refine|reconstruct> <particles> <out.json>
set -e
W=${1:-reconstruct}; N=${2:-16000}; OUT=${3:-gpurun_out/pmc_traffic.json}
export TMPDIR=/tmp
R=/tmp/pmc_traffic; rm -rf $R; mkdir -p $R
ARGS="--particles $N --steps 1 --warmup 0"
if [ "$W" = "reconstruct" ]; then ARGS="--workload reconstruct $ARGS"; else ARGS="$ARGS --no-cpu --no-dropin"; fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py $ARGS > $R/log$i.txt 2>&1 || { echo "group $i failed"; tail -3 $R/log$i.txt; }
done
python3 scripts/pmc_summary.py $R > $OUT
echo done
