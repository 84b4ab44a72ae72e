#!/bin/bash
# k_prep<256, 3> at 128^2 (csp block): row pairs per row pass (PPM_PREP_L) x column chunks (PPM_PREP_NCH), ms per 20 500 projections
for L in 0 4 8 16; do for NCH in 0 3 4 5 7; do
  e=""; [ $L -gt 0 ] && e="$e PPM_PREP_L=$L"; [ $NCH -gt 0 ] && e="$e PPM_PREP_NCH=$NCH"
  r=$(env $e timeout -k 10 200 python bench.py --workload csp --steps 4 --warmup 1 --no-cpu --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('csp',d)
print(b.get('device_ms_per_step'))")
  echo "L=$L NCH=$NCH | $r"
done; done
