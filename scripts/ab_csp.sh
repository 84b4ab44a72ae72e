# k_csp_eval on one box: blocks per CU (PPM_CSP_BLOCKS_PER_CU; 0 = what registers and LDS allow), the csp block of the bench line
for round in 1 2; do
  for bpc in 0 4 3 2; do
    c=$(PPM_CSP_BLOCKS_PER_CU=$bpc timeout -k 10 200 python bench.py --workload csp --steps 2 --warmup 1 --no-cpu --no-side 2>&1 | grep -o '"value": [0-9.]*\|"local": [0-9.]*' | head -2 | tr '\n' ' ')
    echo "== blocks per CU $bpc round $round | csp $c"
  done
done
