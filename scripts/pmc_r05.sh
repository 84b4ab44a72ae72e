#!/bin/bash
# Round-5 PMC summaries of the CURRENT kernels (one counter group per rocprofv3 run, --kernel-trace only; FETCH_SIZE and
# WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes).  Raw CSVs stay under /tmp; the per-kernel summaries
# (with the library's sha and the unit count in "_meta") go to gpurun_out/ and are then copied to profiles/.
#   usage: scripts/pmc_r05.sh <refine|refine0|reconstruct|sva|csp> <units> <out.json>     (refine0: search range 0 = the mask radius, k_gfft)
#   units = particles (refine, reconstruct), resident sub-volumes (sva) or particles of the tilt series (csp: x 41 projections)
set -e
W=${1:-refine}; N=${2:-8192}; OUT=${3:-gpurun_out/r05_pmc_$W.json}
export TMPDIR=/tmp
R=/tmp/pmc_r05_$W; rm -rf $R; mkdir -p $R
COMMON=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
UNITS=$N; EXTRA=""
case "$W" in
  reconstruct) ARGS="--workload reconstruct --recon-particles $N --steps 1 --warmup 0 --no-cpu --no-dropin"; TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum";;
  sva)         ARGS="--workload sva --sva-volumes $N --steps 1 --warmup 0 --no-cpu --no-side"; TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum";;
  csp)         ARGS="--workload csp --csp-particles $N --steps 1 --warmup 0 --no-cpu --no-side"; TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; UNITS=$((N * 41));;
  refine0)     ARGS="--workload refine --search-range 0 --particles $N --steps 1 --warmup 0 --no-cpu --no-dropin --no-side"; TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum";;
  *)           ARGS="--workload refine --particles $N --steps 1 --warmup 0 --no-cpu --no-dropin --no-side"; TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum";;
esac
# vector L1 of the gather kernels (k_local, k_csp_eval, k_sva_eval): lines looked up, requests sent on to L2, texture-addresser busy
TCP="TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
GROUPS_=("${COMMON[@]}" "$TCC" "$TCP" "FETCH_SIZE" "WRITE_SIZE")
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $R/g$i -o g$i --output-format csv -- python3 bench.py $ARGS > $R/log$i.txt 2>&1 || { echo "group $i ($grp) failed"; tail -3 $R/log$i.txt; }
  echo "pass $i done: $grp"
done
# gathered samples per unit of the gather kernels (k_sva_eval, k_csp_eval): read from the bench line of the last pass
GPU=$(python3 - "$R/log$i.txt" <<'PY'
import json, sys
g = 0
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        r = json.loads(ln).get("roofline", {})
        if r.get("gathered_samples_per_launch") and r.get("launches"):
            per = r.get("states_per_launch") or r.get("projections_per_launch") or 1
            g = int(r["gathered_samples_per_launch"] * r["launches"] / per)
print(g)
PY
)
SHA=$(python3 -c "import bench; print(bench.so_sha16())")
KSHA=$(python3 -c "import bench; print(bench.kernels_sha16())")
MSHA=$(python3 -c "import bench; print(bench.kernels_sha16(bench.KERNEL_SOURCES_MAIN))")
META="particles=$UNITS workload=$W so_sha16=$SHA kernels_sha16=$KSHA"
if [ "$W" = "refine" ] || [ "$W" = "refine0" ] || [ "$W" = "reconstruct" ]; then META="$META kernels_main_sha16=$MSHA"; else META="$META gathers_per_unit=$GPU"; fi
python3 scripts/pmc_summary.py $R --meta $META "command=bench.py $ARGS" > $OUT
echo done $OUT
