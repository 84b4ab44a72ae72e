# every library under ab/*.so through the sva block (512 resident 192^3 sub-volumes), three rounds; device ms per sub-volume with 4 decimals
cp pyp_amd/libpypmatch.so /tmp/keep.so
for round in 1 2 3; do
  for f in ab/*.so; do
    cp $f pyp_amd/libpypmatch.so
    s=$(timeout -k 10 300 python - <<'PY'
import subprocess, json, sys
r = subprocess.run([sys.executable, "bench.py", "--workload", "sva", "--sva-volumes", "512", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-side"], capture_output=True, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]); d = d.get("sva", d)
print(d["value"], d["device_ms_per_sub_volume"], d.get("kernels_ms", ""))
PY
)
    echo "== $(basename $f) round $round | $s"
  done
done
cp /tmp/keep.so pyp_amd/libpypmatch.so
