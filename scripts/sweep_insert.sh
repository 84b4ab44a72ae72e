#!/bin/bash
# brick insertion knobs with the round-5 kernel (reconstruct block, 500 k x 256^2): slices per heavy brick, particles per slice, chunk size
for e in "" "PPM_BRICK_SLICES=8" "PPM_BRICK_SLICES=32" "PPM_BRICK_MINP=512" "PPM_BRICK_MINP=2048" "PPM_BRICK_MINP=4096" "PPM_INSERT_GB=4" "PPM_INSERT_GB=12" "PPM_INSERT_GB=16" "PPM_BRICK_SLICES=32 PPM_BRICK_MINP=512"; do
  r=$(env $e timeout -k 10 300 python bench.py --workload reconstruct --steps 1 --warmup 1 --no-cpu --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('reconstruct',d)
print(b.get('value'), b.get('kernels_us_per_particle'))")
  echo "[$e] | $r"
done
