# k_local diagnosis on one box: every library under ab/*.so through the 20 k-particle refinement (ms per step of global / local)
cp pyp_amd/libpypmatch.so /tmp/keep.so
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    r=$(timeout -k 10 200 python bench.py --workload refine --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | head -5 | tr '\n' ' ')
    echo "== $(basename $f) round $round | refine $r"
  done
  cp ab/4_new_m4.so pyp_amd/libpypmatch.so
  r=$(PPM_LOCAL_TABLES=0 timeout -k 10 200 python bench.py --workload refine --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | head -5 | tr '\n' ' ')
  echo "== 4_new_m4.so PPM_LOCAL_TABLES=0 round $round | refine $r"
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
