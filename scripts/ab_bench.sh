# A/B on one box: run the quick bench with each library under ab/*.so (alternating, two rounds).  Usage: ab_bench.sh [refine|reconstruct]
W=${1:-refine}
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    if [ "$W" = "reconstruct" ]; then
      echo "== $(basename $f) round $round: $(timeout -k 10 200 python bench.py --workload reconstruct --particles 50000 --steps 1 --warmup 1 2>&1 | grep -o '"value": [0-9.]*\|"prep": [0-9.]*\|"insert": [0-9.]*' | tr '\n' ' ')"
    else
      echo "== $(basename $f) round $round: $(timeout -k 10 200 python bench.py --particles 20000 --steps 1 --warmup 1 --no-cpu --no-dropin 2>&1 | grep -o '"value": [0-9.]*\|"global": [0-9.]*\|"local": [0-9.]*' | tr '\n' ' ')"
    fi
  done
done
