timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
bash scripts/quick_bench.sh
echo "== PT=1024"; PPM_PREP_PT=1024 bash scripts/quick_bench.sh
