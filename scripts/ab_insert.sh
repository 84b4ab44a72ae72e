#!/bin/bash
# insertion parity tests, then reconstruct timing with the current library and the stamps build (ab/libpypmatch_insstamps.so)
set -e
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "insert or reconstruct or finalize" 2>&1 | tail -3
for i in 1 2; do
timeout -k 10 300 python bench.py --workload reconstruct --steps 1 --warmup 1 --no-cpu --no-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('reconstruct',d)
print('reconstruct', b.get('value'), b.get('kernels_us_per_particle') or b.get('device_ms_per_step'), b.get('parity_vs_oracle'))"
done
if [ -f ab/libpypmatch_insstamps.so ]; then
cp pyp_amd/libpypmatch.so /tmp/keep.so; cp ab/libpypmatch_insstamps.so pyp_amd/libpypmatch.so
timeout -k 10 300 python bench.py --workload reconstruct --recon-particles 32768 --steps 1 --warmup 0 --no-cpu --no-dropin 2>&1 >/dev/null | grep stamps | tail -1
cp /tmp/keep.so pyp_amd/libpypmatch.so
fi
