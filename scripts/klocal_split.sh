set -e
export TMPDIR=/tmp
rm -rf /tmp/prof_kl; mkdir -p /tmp/prof_kl
rocprofv3 --kernel-trace -d /tmp/prof_kl -o kl --output-format csv -- python3 bench.py --particles 16000 --steps 1 --warmup 0 --no-cpu --no-dropin > /tmp/prof_kl/out.txt 2>&1
f=$(find /tmp/prof_kl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_local' in r['Kernel_Name'] or 'k_global' in r['Kernel_Name'] or 'k_prep' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows: print(r['Kernel_Name'][:30], r.get('Grid_Size_X', r.get('Grid_Size','?')), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, 'ms')
PY
