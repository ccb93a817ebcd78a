import sys, time, numpy as np
sys.path.insert(0, 'tests')
from bbqlib import bbq_amd as B
rng = np.random.default_rng(3)
for n, dim in ((200000, 768), (1000000, 768)):
    base = rng.standard_normal((n, dim)).astype(np.float32)
    for sim in (1,):
        t = time.perf_counter(); ix, codes, corr, cen = B.Index.build(base, sim, want_host_copy=False); t1 = time.perf_counter() - t; ix.close()
        t = time.perf_counter(); ix, codes, corr, cen = B.Index.build(base, sim, want_host_copy=True); t2 = time.perf_counter() - t
        t = time.perf_counter(); hc, hr, hcen = B.quantize_vectors(base, sim, n_threads=16); t3 = time.perf_counter() - t
        print("n=%d dim=%d: device build %.3fs (no host copy) %.3fs (with codes/corr download); host quantizer 16 threads %.3fs; equal %s %s" %
              (n, dim, t1, t2, t3, (codes == hc).all(), (corr.view(np.uint64) == hr.view(np.uint64)).all()))
        ix.close()
