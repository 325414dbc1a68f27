R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/power_probe.sh scan256 python bench.py --rows 100000000 --steps 150 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline > gpurun_out/r03_power_scan256.txt 2>&1
bash scripts/power_probe.sh scan128 python bench.py --rows 100000000 --queries 128 --steps 200 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline > gpurun_out/r03_power_scan128.txt 2>&1
bash scripts/power_probe.sh pq python scripts/pq_scan_bench.py 1e8 32 > gpurun_out/r03_power_pq.txt 2>&1
