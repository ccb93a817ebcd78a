import sys, time, numpy as np
sys.path.insert(0, 'tests')
from bbqlib import bbq_amd as B
import orclib as O
base = O.mulberry32(3, 1000*128).reshape(1000,128); q = O.mulberry32(4,128)
t=time.perf_counter(); codes,corr,cen=B.quantize_vectors(base,1); t1=time.perf_counter()-t
cdp=B.centroid_dp(cen)
ix=B.Index(codes,corr,128,cdp); ix.close()
for rep in range(3):
    t=time.perf_counter(); ix=B.Index(codes,corr,128,cdp); t2=time.perf_counter()-t
    qq,qc=B.quantize_query(q,cen,1,4)
    t=time.perf_counter(); ix.search(qq,qc,4,1,10); t3=time.perf_counter()-t
    t=time.perf_counter(); ix.search(qq,qc,4,1,10); t4=time.perf_counter()-t
    t=time.perf_counter(); ix.close(); t5=time.perf_counter()-t
    print("quantize %.2f ms  create %.2f ms  first search %.2f ms  second %.3f ms  destroy %.2f ms"%(t1*1e3,t2*1e3,t3*1e3,t4*1e3,t5*1e3))
t=time.perf_counter(); codes,corr,cen=B.quantize_vectors(base,1,n_threads=1); print("quantize 1 thread %.2f ms"%((time.perf_counter()-t)*1e3))
t=time.perf_counter(); codes,corr,cen=B.quantize_vectors(base,1,n_threads=8); print("quantize 8 threads %.2f ms"%((time.perf_counter()-t)*1e3))
