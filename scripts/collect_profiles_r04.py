#!/usr/bin/env python3
"""Copy the summaries scripts/profile_r04.sh left under gpurun_out/r04/ (and the default bench line) into profiles/, with headers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r04") + "/"
P = os.path.join(ROOT, "profiles") + "/"
prof = json.loads(open(R + "bench_profiled.json").readline())
open(P + "r04_bench_1e8_profiled.json", "w").write(json.dumps(prof) + "\n")
hdr = ("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 12 --warmup 2 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline --no-callers --no-shard-point --no-ann-scale\n"
       "# (scripts/profile_r04.sh; one MI355X, 1e8 x 1152 fp16 rows).  The same command printed (profiles/r04_bench_1e8_profiled.json): %.0f queries/s, %.2f ms per step at %d queries per step,\n"
       "# roofline.avg_launch_ms %.3f (HIP events inside bench.py) -- the headline kernel below (scan_mfma_kernel<2,20> = the 320-query pass since the second half of round 4): 4 dispatches of the\n"
       "# queries-per-pass pick + 2 warm-up + 12 timed + the restore step, all at 1e8 rows.  scan_mfma2d_kernel<3,16> = the 256-query pass, scan_mfma_kernel<3,12> = 192, <3,8> = 128 (pick + hbm_bound_point).\n"
       % (prof["value"], prof["ms_per_step"], prof["config"]["queries_per_step"], prof["roofline"]["avg_launch_ms"]))
open(P + "r04_bench_1e8_kernel_stats.txt", "w").write(hdr + open(R + "bench_kernel_stats.txt").read())
pql = json.loads(open(R + "pq_bench_line.json").readline())
hdr = ("# rocprofv3 --kernel-trace --stats --output-format csv -- python scripts/pq_scan_bench.py 1e8   (scripts/profile_r04.sh; 1e8 x 64-byte codes + 4 descriptor bytes, top-200;\n"
       "# >= 1 s of one-query calls, >= 1 s of 32-query calls = 4 groups of EIGHT queries per pass on two streams, 64-query calls, then 24 eight-query calls on one stream).\n"
       "# The same command printed: %.3f ms per query one per call, %.4f ms per query batched = %.0f queries/s, scan kernel alone (HIP events, one stream) %.3f ms = %.3f of the HBM peak,\n"
       "# end to end %.3f; %d uncertified.  NB the average below mixes launches of the two-stream phase, whose interval includes waiting for the other stream's scan\n"
       "# (the device runs one scan at a time), with the single-stream launches; pq_adc / select averages are inflated the same way (they run on the 8 CUs a scan leaves free).\n"
       % (pql["ms_per_query"], pql["ms_per_query_batched"], pql["queries_per_s_batched"], pql["roofline"]["avg_launch_ms"], pql["roofline"]["frac"],
          pql["roofline"]["end_to_end"]["frac"], pql["uncertified_queries_last_batch"]))
open(P + "r04_pq_scan_stats.txt", "w").write(hdr + open(R + "pq_kernel_stats.txt").read())
open(P + "r04_pq_scan_bench_line.json", "w").write(json.dumps(pql) + "\n")
open(P + "r04_pmc_traffic.json", "w").write(open(R + "pmc_traffic.json").read())
prev = ("# earlier boxes of the same round: 128: 3044 / 3090 / 3135 q/s (41.55 / 40.94 / 40.33 ms), 192: 3740 / 3917 / 3961 q/s (50.66 / 48.37 / 47.79 ms), "
        "256: 4150 / 4329 / 4369 q/s (60.88 / 58.32 / 57.73 ms), 320: 4517 q/s (69.75 ms; 384 = 24 column tiles: 68 spilled registers, 151.3 ms, 2518 q/s), all at 1.37-1.40 kW\n")
open(P + "r04_scan_variants.txt", "w").write(open(R + "scan_variants.txt").read() + prev)
# (profiles/r04_bench_default.json is copied from the default run by hand: it is not part of the profile script)
print("ok")
