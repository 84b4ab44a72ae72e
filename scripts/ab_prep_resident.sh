#!/bin/bash
# k_prep's LDS-resident small-box path against the chunked one (PPM_PREP_RESIDENT=0) on one box: parity tests, then the csp block
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_csp.py tests/test_extract.py -x -q -m gpu 2>&1 | tail -2
for round in 1 2; do
  for v in 1 0; do
    r=$(PPM_PREP_RESIDENT=$v timeout -k 10 300 python bench.py --workload csp --steps 5 --warmup 2 --no-cpu --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('csp',d)
print(d.get('value'), b.get('device_ms_per_step'))")
    echo "== resident=$v round $round | $r"
  done
done
