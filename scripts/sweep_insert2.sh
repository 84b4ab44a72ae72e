#!/bin/bash
for e in "PPM_INSERT_GB=16" "PPM_INSERT_GB=24" "PPM_INSERT_GB=32" "PPM_INSERT_GB=48" "PPM_INSERT_GB=8"; do
  r=$(env $e timeout -k 10 300 python bench.py --workload reconstruct --steps 1 --warmup 1 --no-cpu --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('reconstruct',d)
print(b.get('value'), b.get('kernels_us_per_particle'))")
  echo "[$e] | $r"
done
