#!/bin/bash
# every library under ab/*.so through the reconstruction block (us per particle of k_prep / k_insert_bricks), two rounds on one box
cp pyp_amd/libpypmatch.so /tmp/keep.so
for round in 1 2; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    r=$(timeout -k 10 300 python bench.py --workload reconstruct --steps 1 --warmup 1 --no-cpu --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('reconstruct',d)
print(b.get('value'), b.get('kernels_us_per_particle'))")
    echo "== $(basename $f) round $round | $r"
  done
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
