# A/B on one box: run the refine quick bench with each library under ab/*.so (alternating, two rounds)
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    echo "== $(basename $f) round $round: $(timeout -k 10 200 python bench.py --particles 20000 --steps 1 --warmup 1 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | tr '\n' ' ')"
  done
done
