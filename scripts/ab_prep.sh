#!/bin/bash
# A/B of k_prep variants on one box: reconstruct workload, 100k particles, kernel times from the library's HIP events
for v in "PPM_PREP_SLOTS=3" "PPM_PREP_SLOTS=4" "PPM_PREP_SLOTS=6" "PPM_PREP_SLOTS=12" "PPM_PREP_SLOTS=400"; do
  echo "== $v"
  env $v python bench.py --workload reconstruct --recon-particles 100000 --steps 2 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"kernels_us_per_particle": {[^}]*}\|ERROR.*' | head -2
done
