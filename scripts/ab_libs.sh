# every library under ab/*.so through the 20 k-particle refinement and the csp block (ms per step of the kernels), two rounds
cp pyp_amd/libpypmatch.so /tmp/keep.so
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    r=$(timeout -k 10 200 python bench.py --workload refine --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"kernels_ms": {[^}]*}' | head -1)
    c=$(timeout -k 10 200 python bench.py --workload csp --steps 2 --warmup 1 --no-cpu --no-side 2>&1 | grep -o '"value": [0-9.]*\|"local": [0-9.]*' | head -2 | tr '\n' ' ')
    echo "== $(basename $f) round $round | refine $r | csp $c"
  done
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
