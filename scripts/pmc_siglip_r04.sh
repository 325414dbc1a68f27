# HBM bytes per kernel of the SigLIP image tower (batch 256, depth 27): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, over
# `scripts/siglip_bench.py 256 2 27` (3 forwards).  FETCH_SIZE x 2 KiB (MI355X_MICROARCH.md, HBM section), WRITE_SIZE x 1 KiB (uncalibrated).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_siglip; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $R/scripts/siglip_bench.py 256 2 27 > $OUT/$c.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "siglip" in r["Kernel_Name"]:
                k = r["Kernel_Name"]
                k = k[k.find("::", 20) + 2:][:70] if "mse::" in k else k[:70]
                a = agg[k]
                if ci == 0: a[0] += 1
                a[1 + ci] += float(r["Counter_Value"])
print("# per forward (3 forwards profiled): dispatches, HBM read GB (FETCH_SIZE x 2 KiB), HBM write GB (WRITE_SIZE x 1 KiB), kernel")
tot = [0.0, 0.0]
for k, (n, f, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%6.1f %9.2f %9.2f  %s" % (n / 3, f * 2048 / 3 / 1e9, w * 1024 / 3 / 1e9, k))
    tot[0] += f * 2048 / 3 / 1e9; tot[1] += w * 1024 / 3 / 1e9
print("total  %9.2f %9.2f" % tuple(tot))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
