#!/bin/bash
# PMC passes over the refine workload (one counter group per run, kernel-trace only); raw CSVs stay under /tmp.
set -e
N=${1:-8000}; OUT=${2:-gpurun_out/pmc_refine.json}
export TMPDIR=/tmp
R=/tmp/pmc_refine; rm -rf $R; mkdir -p $R
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py --particles $N --steps 1 --warmup 0 --no-cpu --no-dropin > $R/log$i.txt 2>&1 || { echo "group $i failed"; tail -3 $R/log$i.txt; }
done
python3 scripts/pmc_summary.py $R > $OUT
echo done
