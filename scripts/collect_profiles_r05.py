#!/usr/bin/env python3
"""Copy the summaries scripts/profile_r05.sh (and the probes of round 5) left under gpurun_out/ into profiles/, with headers."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out") + "/"
R = G + "r05/"
P = os.path.join(ROOT, "profiles") + "/"


def have(f):
    return os.path.exists(f) and os.path.getsize(f) > 0


prof = json.loads(open(R + "bench_profiled.json").readline())
open(P + "r05_bench_1e8_profiled.json", "w").write(json.dumps(prof) + "\n")
hdr = ("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 12 --warmup 2 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline --no-callers --no-shard-point --no-ann-scale\n"
       "# (scripts/profile_r05.sh; one MI355X, 1e8 x 1152 fp16 rows).  The same command printed (profiles/r05_bench_1e8_profiled.json): %.0f queries/s, %.2f ms per step at %d queries per step,\n"
       "# roofline.avg_launch_ms %.3f (HIP events inside bench.py) -- the headline kernel below (scan_mfma_kernel<2,20> = the 320-query pass; frozen since round 4): 4 dispatches of the\n"
       "# queries-per-pass pick + 2 warm-up + 12 timed + the restore step, all at 1e8 rows.  scan_mfma2d_kernel<3,16> = the 256-query pass, scan_mfma_kernel<3,12> = 192, <3,8> = 128 (pick + hbm_bound_point).\n"
       % (prof["value"], prof["ms_per_step"], prof["config"]["queries_per_step"], prof["roofline"]["avg_launch_ms"]))
open(P + "r05_bench_1e8_kernel_stats.txt", "w").write(hdr + open(R + "bench_kernel_stats.txt").read())
hdr = ("# The PQ flat scan at 1e8 x (64 B codes + 4 descriptor bytes), top-200 -> top-10, in the two patterns bench.py's pq_scan.roofline reports (scripts/profile_r05.sh, scripts/pq_trace_r05.py).\n"
       "# (1) BURST: rocprofv3 --kernel-trace --stats -- python scripts/pq_trace_r05.py burst -- 42 calls of EIGHT queries = one scan per call on ONE stream, nothing before or beside it:\n"
       "#     the average of pq_scan64x4_kernel<16, 8> below is the burst cost of a pass; 6.8 GB / that = pq_scan.roofline.burst.\n")
body = open(R + "pq_burst_stats.txt").read()
sus = open(R + "pq_sustained_trace.txt").read() if have(R + "pq_sustained_trace.txt") else "(not collected)\n"
tail = ("# (2) SUSTAINED: rocprofv3 --kernel-trace -- python scripts/pq_trace_r05.py sustained -- 18 calls of 64 queries = eight scans per call back to back, alternating between two streams with\n"
        "#     the tails beside them; per call, first scan's start to last scan's end / 8 from the trace's own timestamps (a launch's own duration there includes its wait for the other stream's\n"
        "#     scan: the device runs one scan at a time).  6.8 GB / that = pq_scan.roofline.frac x 8 TB/s.  Why sustained > burst: profiles/r05_pq_clock_probe.txt.\n")
open(P + "r05_pq_scan_stats.txt", "w").write(hdr + body + tail + sus)
open(P + "r05_pmc_traffic.json", "w").write(open(R + "pmc_traffic.json").read())
if have(R + "request_path_trace.txt"):
    open(P + "r05_request_path_trace.txt", "w").write(
        "# rocprofv3 --kernel-trace of ONE 64-query request-path call (mse_disk_query_topk_f32, 2e6 easy-set rows, L = 12, beam 4) per entry rule: what a coalesced submission\n"
        "# of 64 one-query requests consists of on the device (scripts/trace_request_path.sh).  Round 4's entry step for the row table was a dozen launches + a host synchronisation.\n"
        + open(R + "request_path_trace.txt").read())
for src, dst, hdr in (
        ("pq_clock.txt", "r05_pq_clock_probe.txt",
         "# scripts/pq_clock_probe.py 3: engine / memory clock, socket power and temperatures of OUR device (sysfs hwmon by PCI address) every ~10 ms beside three launch patterns of the PQ scan.\n"
         "# Reading: back-to-back scans draw 1.30 kW and the firmware holds the engine clock at 1.70 GHz; with a pause in front of every scan the same kernel sees 2.0-2.15 GHz at 0.7-1.1 kW and\n"
         "# takes 8-10 % less (1.25-1.27 ms against 1.39 per scan).  The memory clock never moves (2000 MHz): the pass is HBM-bound only up to what its on-chip half (address generation, LDS\n"
         "# gathers, i8 MFMAs) can issue at the clock the power budget leaves.\n"),
        ("hardness_2e6.txt", "r05_hardness_probe.txt",
         "# scripts/hardness_probe.py 2e6: hardness statistics and the search list a one-pass Vamana graph needs, per generator setting (bench_ann.py HardSet; 'b' became HARD_PARAMS).\n"
         "# sweep = [search list, recall@10, queries/s at 2048 per call (one wave per query at the time), node fetches per query]\n"),
        ("build_order.txt", "r05_build_order_probe.txt",
         "# scripts/build_order_probe.py hard 4e6: position of a point INSIDE its build batch sorted by coarse cluster (256 k-means cells) against the random order; the batches are the same subsets.\n"),
        ("build_batch_easy.txt", "r05_build_batch_probe.txt",
         "# scripts/build_batch_probe.py easy 1e7 4096 65536 262144 (the build caps a batch at 65536 points)\n"),
        ("beam_lat_1w.txt", "r05_beam_latency_one_wave.txt", "# scripts/beam_latency_probe.py (developer library), one wave per query for every exactly scored search: hard set, 2e6 rows, median of 8 calls\n"),
        ("beam_lat_4w.txt", "r05_beam_latency_four_waves.txt", "# MSE_BEAM_FOUR_WAVES=1 scripts/beam_latency_probe.py: four waves per query -- the product now uses this form below 1025 queries per call\n"),
        ("siglip_lat_b1.txt", "r05_siglip_latency_b1_before.txt", "# scripts/trace_siglip_latency.sh 1 BEFORE gemm_skinny_kernel: one text / one image through the 256 x 256-tile GEMMs\n"),
        ("siglip_lat_b1_skinny.txt", "r05_siglip_latency_b1.txt", "# scripts/trace_siglip_latency.sh 1 with gemm_skinny_kernel in the text tower (the image tower is unchanged)\n"),
        ("callers_probe5.txt", "r05_graph_callers_probe.txt", "# scripts/graph_callers_probe.py 2e6 12: T native threads x 1 query through mse_disk_query_topk_f32 (easy set, L = 12), coalescer settings (queries per submission, wait us, workers); cgroup lines = /sys/fs/cgroup cpu.max / cpu.stat of the box\n")):
    if have(G + src):
        open(P + dst, "w").write(hdr + open(G + src).read())
if have(G + "bench_default.json"):
    shutil.copy(G + "bench_default.json", P + "r05_bench_default.json")
if have(G + "r05_gpu_tests.txt"):
    shutil.copy(G + "r05_gpu_tests.txt", P + "r05_gpu_tests.txt")
for k in ("hard", "ood", "easy"):
    if have(G + f"gi_{k}.json"):
        shutil.copy(G + f"gi_{k}.json", P + f"r05_graph_index_1e7_{k}_first_run.json")
print("ok")
