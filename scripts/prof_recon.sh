set -e
export TMPDIR=/tmp
rm -rf /tmp/prof_rc; mkdir -p /tmp/prof_rc gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_rc -o rc --output-format csv -- python3 bench.py --workload reconstruct --particles 100000 --steps 2 --warmup 1 > gpurun_out/r01_bricks_reconstruct_line.json 2> /tmp/prof_rc/err.txt
f=$(find /tmp/prof_rc -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/r01_bricks_reconstruct_kernel_stats.csv; grep "ppm::" $f >> gpurun_out/r01_bricks_reconstruct_kernel_stats.csv
cat gpurun_out/r01_bricks_reconstruct_kernel_stats.csv | cut -c1-150
tail -1 gpurun_out/r01_bricks_reconstruct_line.json | cut -c1-300
