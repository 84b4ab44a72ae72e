#!/bin/bash
# kernel times of both workloads for the current library (one box); usage: ab_prep_env.sh [label]
echo "== ${1:-current}"
python bench.py --workload reconstruct --recon-particles 100000 --steps 2 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"kernels_us_per_particle": {[^}]*}\|ERROR.*' | head -2
python bench.py --workload refine --particles 28672 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"kernels_us_per_particle": {[^}]*}\|ERROR.*' | head -2
