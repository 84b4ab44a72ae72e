# k_local on one box: blocks per CU (PPM_LOCAL_BLOCKS_PER_CU; 0 / unset = 4, what the 128 registers allow), 20 k-particle refinement
for round in 1 2; do
  for bpc in 0 3 2; do
    r=$(PPM_LOCAL_BLOCKS_PER_CU=$bpc timeout -k 10 200 python bench.py --workload refine --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | head -5 | tr '\n' ' ')
    echo "== blocks per CU $bpc round $round | refine $r"
  done
done
# (compile-time probes of round 4, results wrong, timing only: -DPPM_DBG_NOSWEEP no sample loop, -DPPM_DBG_NOATOM ring sums kept out of LDS,
#  -DPPM_DBG_SMALLCUBE every tap offset masked into a 16 KB window; build the variants into ab/*.so and loop over them as scripts/ab_r04.sh does)
