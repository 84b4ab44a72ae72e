#!/bin/bash
# A/B of the scratch-free k_prep path (PPM_PREP_INREG=1) on one box: parity at 256, then kernel times of both workloads
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "256 or insertion or largest or two_live or config" 2>&1 | tail -3
for v in 0 1; do
  echo "== PPM_PREP_INREG=$v"
  PPM_PREP_INREG=$v python bench.py --workload reconstruct --recon-particles 100000 --steps 2 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"kernels_us_per_particle": {[^}]*}\|ERROR.*' | head -2
  PPM_PREP_INREG=$v python bench.py --workload refine --particles 28672 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"kernels_us_per_particle": {[^}]*}\|ERROR.*' | head -2
done
