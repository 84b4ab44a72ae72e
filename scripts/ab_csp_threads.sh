#!/bin/bash
for round in 1 2; do for t in 256 128 64; do
  r=$(PPM_CSP_THREADS=$t timeout -k 10 200 python bench.py --workload csp --steps 4 --warmup 1 --no-cpu --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('csp',d)
print(d.get('value'), b.get('device_ms_per_step'), b.get('parity_vs_oracle'))")
  echo "csp threads $t round $round | $r"
done; done
PPM_CSP_THREADS=128 timeout -k 10 600 python -m pytest tests/test_gpu_csp.py -x -q 2>&1 | tail -2
