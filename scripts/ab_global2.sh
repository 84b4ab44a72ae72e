#!/bin/bash
# global-search parity tests with the current library, then scripts/ab_global.sh over ab/*.so (--no-side: only the timed workload)
cp pyp_amd/libpypmatch.so /tmp/keep.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "global or full_refinement or tile or window or symmetry or grid" 2>&1 | tail -2
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    echo "== $(basename $f) round $round: $(timeout -k 10 300 python bench.py --workload refine --particles 28672 --steps 1 --warmup 1 --no-cpu --no-dropin --no-side 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"norms": [0-9.]*\|"local": [0-9.]*\|ERROR.*' | tr '\n' ' ')"
  done
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
