cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_pq; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/pqmini.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/meme-search-engine_amd")
import torch, numpy as np, mse, time
D=1152; n=100_000_000
rng=np.random.default_rng(0)
cents=(rng.standard_normal((256,D))/np.sqrt(D)).astype(np.float32); T=np.linalg.qr(rng.standard_normal((D,D)))[0].astype(np.float32)
pq=mse.ProductQuantizer(cents,T,18,D)
blk=1_000_000; block=rng.integers(0,256,size=(blk,64),dtype=np.uint8); codes=np.empty((n,64),np.uint8)
for c0 in range(0,n,blk): np.bitwise_xor(block, rng.integers(0,256,size=64,dtype=np.uint8), out=codes[c0:c0+blk])
desc=np.resize(rng.integers(0,256,size=(blk,4),dtype=np.uint8),(n,4))
gc=mse.Codes(codes,desc); del codes
scales=np.array([0.5,0,-0.25,0],np.float32)/np.float32(512)
qs=(rng.standard_normal((int(os.environ.get("PQ_TRACE_NQ","32")),D))/np.sqrt(D)).astype(np.float32)
for _ in range(3): pq.scan_topk_batch(gc,qs,200,10,None,scales)
t=time.perf_counter()
for _ in range(4): pq.scan_topk_batch(gc,qs,200,10,None,scales)
print("ms per call", (time.perf_counter()-t)/4*1e3)
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python /tmp/pqmini.py > $OUT/log.txt 2>&1
grep "ms per call" $OUT/log.txt
python - $OUT <<'PY'
import csv,glob,sys,os
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=int(rows[-1]["End_Timestamp"]); t0=last-int(os.environ.get("PQ_TRACE_WINDOW_US","7000"))*1000
sel=[r for r in rows if int(r["Start_Timestamp"])>t0]
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    nm=r["Kernel_Name"]; nm=nm[nm.find("::",20)+2:][:40] if "mse::" in nm else nm[:40]
    print("%9.1f %8.1f q=%s %s"%((s-t0)/1e3,(e-s)/1e3,r.get("Queue_Id"),nm))
PY
