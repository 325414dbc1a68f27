#!/bin/bash
# Round-6 profiles -> gpurun_out/r06/ (scripts/collect_profiles_r06.py copies the summaries into profiles/):
#   bench_kernel_stats.txt       rocprofv3 --kernel-trace --stats of the bench command's scan legs (1e8 rows) + the line that run printed
#   siglip_b256_kernel_stats.txt / text_b256_kernel_stats.txt   the towers at batch 256, kernel by kernel
#   pmc_traffic.json             FETCH_SIZE / WRITE_SIZE passes (scan legs, PQ scan, SigLIP forward), each counter in a pass of its own
#   prof_beam_hard/summary.txt   the graph search on the hard set (scripts/prof_beam_r06.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
stats() {  # $1 = dir with a *kernel_stats.csv, $2 = output text
python - $1 <<'PY' > $2
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("  calls    total_ms      avg_us       %  kernel")
for r in rows[:24]:
    print("%7s %11.3f %11.3f %7s  %s" % (r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:160]))
PY
}
SCAN="--steps 12 --warmup 2 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline --no-callers --no-shard-point --no-ann-scale"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python $R/bench.py $SCAN --detail $OUT/bench_profiled_detail.json > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
stats $OUT/bench $OUT/bench_kernel_stats.txt; rm -rf $OUT/bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sg -o s -- python $R/scripts/siglip_bench.py 256 3 27 > $OUT/siglip_b256.log 2>&1
stats $OUT/sg $OUT/siglip_b256_kernel_stats.txt; rm -rf $OUT/sg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tx -o t -- python $R/scripts/siglip_text_b256.py 256 > $OUT/text_b256.log 2>&1
stats $OUT/tx $OUT/text_b256_kernel_stats.txt; rm -rf $OUT/tx
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/scripts/pq_trace_r05.py burst > $OUT/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcs_$c -o pmc -- python $R/bench.py --steps 4 --warmup 1 ${SCAN#--steps 12 --warmup 2} --detail $OUT/pmcs_detail.json > $OUT/pmcs_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcg_$c -o pmc -- python $R/scripts/siglip_bench.py 256 2 27 > $OUT/pmcg_$c.log 2>&1
done
python $R/scripts/pmc_traffic_r05.py $OUT > $OUT/pmc_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcs_FETCH_SIZE $OUT/pmcs_WRITE_SIZE $OUT/pmcg_FETCH_SIZE $OUT/pmcg_WRITE_SIZE $OUT/pmcs_detail.json
bash $R/scripts/prof_beam_r06.sh 1e7 200 4 > /dev/null 2>&1
ls -la $OUT
