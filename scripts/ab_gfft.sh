#!/bin/bash
# k_gfft parity tests, then the default search (range 0) timed on 8 192 particles, twice
timeout -k 10 500 python -m pytest tests/test_gpu_gfft.py -x -q 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --workload refine --search-range 0 --particles 8192 --steps 1 --warmup 1 --no-cpu --no-dropin --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('range 0:', d.get('value'), d.get('kernels_us_per_particle') or d.get('kernels_ms'))"
done
