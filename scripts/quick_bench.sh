# quick GPU check: reconstruct + refine kernel times (no CPU baseline)
echo "== reconstruct"; timeout -k 10 150 python bench.py --workload reconstruct --particles 50000 --steps 1 --warmup 1 2>&1 | grep -o "\"value\": [0-9.]*\|kernels_ms.*" | cut -c1-120
echo "== refine"; timeout -k 10 200 python bench.py --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o "\"value\": [0-9.]*\|kernels_ms.*}" | cut -c1-200
