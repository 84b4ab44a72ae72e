#!/bin/bash
# threads per block of k_local's hit stage (20 hits x 2 iterations at the search band: <= 800 samples per sweep): 64 / 128 / 256
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_refinement or baseline_size" 2>&1 | tail -2
for round in 1 2; do for t in 128 64 256; do
  r=$(PPM_LOCAL_HIT_THREADS=$t timeout -k 10 300 python bench.py --workload refine --particles 28672 --steps 1 --warmup 1 --no-cpu --no-dropin --no-side 2>&1 | grep -o '"value": [0-9.]*\|"local": [0-9.]*' | tr '\n' ' ')
  echo "hit threads $t round $round | $r"
done; done
PPM_LOCAL_HIT_THREADS=64 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_refinement or baseline_size" 2>&1 | tail -2
