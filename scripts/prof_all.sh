set -e
export TMPDIR=/tmp
rm -rf /tmp/prof_rc /tmp/prof_rf; mkdir -p /tmp/prof_rc /tmp/prof_rf gpurun_out
T=${1:-r01_x}
rocprofv3 --kernel-trace --stats -d /tmp/prof_rc -o rc --output-format csv -- python3 bench.py --workload reconstruct --particles 100000 --steps 2 --warmup 1 > gpurun_out/${T}_reconstruct_line.json 2> /tmp/prof_rc/err.txt
f=$(find /tmp/prof_rc -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/${T}_reconstruct_kernel_stats.csv; grep "ppm::" $f >> gpurun_out/${T}_reconstruct_kernel_stats.csv
cut -c1-110 gpurun_out/${T}_reconstruct_kernel_stats.csv | head -5
tail -1 gpurun_out/${T}_reconstruct_line.json | cut -c1-200
rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o rf --output-format csv -- python3 bench.py > gpurun_out/${T}_bench_line.json 2> /tmp/prof_rf/err.txt
f=$(find /tmp/prof_rf -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/${T}_kernel_stats.csv; grep "ppm::" $f >> gpurun_out/${T}_kernel_stats.csv
cut -c1-110 gpurun_out/${T}_kernel_stats.csv | head -6
tail -1 gpurun_out/${T}_bench_line.json | cut -c1-1500
