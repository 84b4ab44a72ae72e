#!/bin/bash
# blocks per CU of the gather kernels (LDS request as the limiter): csp sweeps and sva search, one box
for e in "" "PPM_CSP_BLOCKS_PER_CU=2" "PPM_CSP_BLOCKS_PER_CU=3" "PPM_CSP_BLOCKS_PER_CU=4" "PPM_CSP_BLOCKS_PER_CU=5" "PPM_CSP_BLOCKS_PER_CU=6" "PPM_CSP_BLOCKS_PER_CU=8"; do
  r=$(env $e timeout -k 10 200 python bench.py --workload csp --steps 4 --warmup 1 --no-cpu --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('csp',d)
print(d.get('value'), b.get('device_ms_per_step'))")
  echo "csp [$e] | $r"
done
for e in "" "PPM_SVA_BLOCKS_PER_CU=1" "PPM_SVA_BLOCKS_PER_CU=2" "PPM_SVA_BLOCKS_PER_CU=3" "PPM_SVA_BLOCKS_PER_CU=4"; do
  r=$(env $e timeout -k 10 200 python bench.py --workload sva --steps 2 --warmup 1 --no-cpu --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('sva',d)
print(d.get('value'), b.get('device_ms_per_sub_volume'))")
  echo "sva [$e] | $r"
done
