# A/B on one box (round 4): every library under ab/*.so through the three gather-bound workloads, two rounds, alternating.
#   refine: k_global / k_local us per particle; sva, csp: units per second of the timed call (--no-side)
cp pyp_amd/libpypmatch.so /tmp/keep.so
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    r=$(timeout -k 10 200 python bench.py --workload refine --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | head -5 | tr '\n' ' ')
    s=$(timeout -k 10 200 python bench.py --workload sva --sva-volumes 512 --steps 1 --warmup 0 --no-cpu --no-side 2>&1 | grep -o '"value": [0-9.]*\|"search": [0-9.]*' | head -2 | tr '\n' ' ')
    c=$(timeout -k 10 200 python bench.py --workload csp --steps 2 --warmup 1 --no-cpu --no-side 2>&1 | grep -o '"value": [0-9.]*\|"local": [0-9.]*' | head -2 | tr '\n' ' ')
    echo "== $(basename $f) round $round | refine $r | sva $s | csp $c"
  done
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
