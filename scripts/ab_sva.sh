# sub-volume search on one box (512 resident 192^3 sub-volumes): blocks of k_sva_eval<6> per CU (0 = what the registers allow: 5)
for round in 1 2; do
  for bpc in 0 3 2 1; do
    s=$(PPM_SVA_BLOCKS_PER_CU=$bpc timeout -k 10 300 python bench.py --workload sva --sva-volumes 512 --steps 2 --warmup 1 --no-cpu --no-side 2>&1 | grep -o '"value": [0-9.]*\|"search": [0-9.]*' | head -2 | tr '\n' ' ')
    echo "== blocks per CU $bpc round $round | sva $s"
  done
done
s=$(timeout -k 10 300 python bench.py --workload sva --sva-volumes 512 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); d=d.get('sva',d); print(d['value'], d['device_ms_per_sub_volume'], d['global_search']['value'], d['average']['value'], d['streamed']['value'], d['parity_vs_oracle'] if 'parity_vs_oracle' in d else '')")
echo "== default, with side figures | $s"
