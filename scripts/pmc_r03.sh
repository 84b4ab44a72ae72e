#!/bin/bash
# Round-3 PMC summaries of the CURRENT kernels (one counter group per rocprofv3 run, --kernel-trace only; FETCH_SIZE and
# WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes).  Raw CSVs stay under /tmp; the per-kernel summaries
# (with the library's sha and the particle count in "_meta") go to gpurun_out/ and are then copied to profiles/.
#   usage: scripts/pmc_r02.sh <refine|reconstruct> <particles> <out.json>
set -e
W=${1:-refine}; N=${2:-8192}; OUT=${3:-gpurun_out/r03_pmc_$W.json}
export TMPDIR=/tmp
R=/tmp/pmc_r03_$W; rm -rf $R; mkdir -p $R
if [ "$W" = "reconstruct" ]; then
  ARGS="--workload reconstruct --recon-particles $N --steps 1 --warmup 0 --no-cpu --no-dropin"
  GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE")
else
  ARGS="--workload refine --particles $N --steps 1 --warmup 0 --no-cpu --no-dropin"
  GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE")
fi
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py $ARGS > $R/log$i.txt 2>&1 || { echo "group $i ($grp) failed"; tail -3 $R/log$i.txt; }
  echo "pass $i done: $grp"
done
SHA=$(python3 -c "import bench; print(bench.so_sha16())")
KSHA=$(python3 -c "import bench; print(bench.kernels_sha16())")
MSHA=$(python3 -c "import bench; print(bench.kernels_sha16(bench.KERNEL_SOURCES_MAIN))")
python3 scripts/pmc_summary.py $R --meta particles=$N workload=$W so_sha16=$SHA kernels_sha16=$KSHA kernels_main_sha16=$MSHA "command=bench.py $ARGS" > $OUT
echo done $OUT
