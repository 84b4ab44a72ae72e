#!/bin/bash
# A/B of k_global build variants on one box (libraries under ab/*.so, built with -DPPM_GLOBAL_* by the caller): refinement
# workload, kernel times from the library's HIP events, two alternating rounds.
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    echo "== $(basename $f) round $round: $(timeout -k 10 300 python bench.py --workload refine --particles ${1:-28672} --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"norms": [0-9.]*\|"local": [0-9.]*\|ERROR.*' | tr '\n' ' ')"
  done
done
