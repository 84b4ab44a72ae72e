#!/bin/bash
# robustness sweep: the two bench workloads at other box sizes / bands (small particle counts), accuracy against the synthetic truth
for cfg in "96 24 8000" "128 48 8000" "192 64 4000" "320 64 2000" "384 96 1500"; do
  set -- $cfg
  echo "== box $1 band $2"
  timeout -k 10 300 python bench.py --box $1 --band $2 --particles $3 --recon-particles $((4*$3)) --steps 1 --warmup 1 --no-cpu --no-dropin --no-next-rows 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('reconstruct', {})
        print('  refine %.0f particles/s, median error %.2f deg, within 2 deg %.3f; reconstruct %.0f particles/s, map cc %.3f' % (d['value'], d['accuracy_vs_truth']['median_deg'], d['accuracy_vs_truth']['frac_within_2deg'], r.get('value', 0), r.get('map_cc_vs_truth', 0)))
    elif 'ERROR' in l or 'Error' in l or 'fault' in l:
        print('  ', l.strip()[:300])
"
done
