set -e
export TMPDIR=/tmp
R=/tmp/pmc_g; rm -rf $R; mkdir -p $R
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py --particles 4000 --steps 1 --warmup 0 --no-cpu --no-dropin > $R/log$i.txt 2>&1 || { echo "group $i failed"; tail -3 $R/log$i.txt; }
done
python3 scripts/pmc_summary.py $R > gpurun_out/pmc_g.json
