#!/usr/bin/env python3
"""Copy the summaries scripts/profile_r06.sh left under gpurun_out/r06 into profiles/, with headers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out") + "/"
R = G + "r06/"
P = os.path.join(ROOT, "profiles") + "/"

prof = json.loads(open(R + "bench_profiled.json").read().strip().splitlines()[-1])
open(P + "r06_bench_1e8_profiled.json", "w").write(json.dumps(prof) + "\n")
hdr = ("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 12 --warmup 2 --no-siglip --no-pq --no-graph --no-graph-scale --no-cpu-baseline --no-callers --no-shard-point --no-ann-scale\n"
       "# (scripts/profile_r06.sh; one MI355X, 1e8 x 1152 fp16 rows).  The same command printed (profiles/r06_bench_1e8_profiled.json): %.0f queries/s, %.2f ms per step at %d queries per step,\n"
       "# roofline.avg_launch_ms %.3f (HIP events inside bench.py) -- the headline kernel below (scan_mfma_kernel<2,20> = the 320-query pass; frozen since round 4): 4 dispatches of the\n"
       "# queries-per-pass pick + 2 warm-up + 12 timed + the restore step, all at 1e8 rows.  scan_mfma2d_kernel<3,16> = the 256-query pass, scan_mfma_kernel<3,12> = 192, <3,8> = 128 (pick + hbm_bound_point).\n"
       % (prof["value"], prof["ms_per_step"], prof["config"]["queries_per_step"], prof["roofline"]["avg_launch_ms"]))
open(P + "r06_bench_1e8_kernel_stats.txt", "w").write(hdr + open(R + "bench_kernel_stats.txt").read())
log = [l for l in open(R + "siglip_b256.log").read().splitlines() if "img/s" in l or "ms" in l][-3:]
open(P + "r06_siglip_b256_kernel_stats.txt", "w").write(
    "# rocprofv3 --kernel-trace --stats -- python scripts/siglip_bench.py 256 3 27 (round 6, one MI355X): the SigLIP image tower at batch 256, depth 27 -- one warm-up + three timed forwards\n"
    "# (every count below is over those four forwards, two sub-batches on two streams each).  The probe printed: " + " | ".join(log) + "\n" + open(R + "siglip_b256_kernel_stats.txt").read())
log = [l for l in open(R + "text_b256.log").read().splitlines() if "text batch" in l][-1:]
open(P + "r06_text_b256_kernel_stats.txt", "w").write(
    "# rocprofv3 --kernel-trace --stats -- python scripts/siglip_text_b256.py 256 (round 6, one MI355X): the SigLIP text tower at batch 256 on the LayerNorm-fused path (six forwards, two parts\n"
    "# on two streams each).  gemm8pp_kernel<5,..> = proj / fc2 with the residual + statistics epilogue, <1,..,true> = fc1 (LayerNorm folded in), <4,..,true> = QKV (q / k columns, V columns,\n"
    "# the 128-column V remainder).  The probe printed: " + " | ".join(log) + "\n" + open(R + "text_b256_kernel_stats.txt").read())
open(P + "r06_pmc_traffic.json", "w").write(open(R + "pmc_traffic.json").read().replace("scripts/profile_r05.sh", "scripts/profile_r06.sh"))
s = open(G + "prof_beam_hard/summary.txt").read()
keep = [l for l in s.splitlines() if not any(t in l for t in ("simple_timer", "output_stream", "tool.cpp", "amdgpu.ids"))]
old = open(P + "r06_beam_search_hard_pmc.txt").read() if os.path.exists(P + "r06_beam_search_hard_pmc.txt") else ""
if "# ======== END OF ROUND" not in old:
    old = "# ======== START OF ROUND 6 (before the iteration changes of DESIGN.md 3.6) ========\n" + old
    open(P + "r06_beam_search_hard_pmc.txt", "w").write(old + "# ======== END OF ROUND 6 (ties per iteration, live-only ranking, 64-flag selection, no second gather of fetched rows, 14 searches per CU) ========\n" + "\n".join(keep) + "\n")
print("ok")
