import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/meme-search-engine_amd')
import mse
from oracle import orc
n,nq,k=20000,130,10
base=orc.gen_rows_f16(0x5EED0001,0,n); q=orc.gen_rows_f16(0x5EED0002,0,nq)
s=mse.Searcher(mse.VectorList.from_f16s(base,1152))
sc,ids=s.bruteforce_topk(q,k,mse.MODE_MFMA)
ws,wi=orc.bruteforce_topk(base,q,k)
bad=[i for i in range(nq) if not np.array_equal(ids[i],wi[i])]
print("bad queries",bad)
print(s.last_stats())
for b in bad[:3]:
    print(b, ids[b], wi[b]); print(sc[b]); print(ws[b])
