#!/bin/bash
# PMC passes over the reconstruct workload (one counter group per run, kernel-trace only); raw CSVs stay under /tmp.
set -e
N=${1:-8000}; OUT=${2:-gpurun_out/pmc_recon.json}
export TMPDIR=/tmp
R=/tmp/pmc_recon; rm -rf $R; mkdir -p $R
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py --workload reconstruct --particles $N --steps 1 --warmup 0 --no-cpu --no-dropin > $R/log$i.txt 2>&1 || { echo "group $i failed"; tail -5 $R/log$i.txt; }
done
python3 scripts/pmc_summary.py $R > $OUT
echo done
