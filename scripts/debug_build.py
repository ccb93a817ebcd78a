import sys, numpy as np
sys.path.insert(0, 'tests')
from bbqlib import bbq_amd as B
import orclib as O
rng = np.random.default_rng(1)
for n, dim, sim in ((64, 8, 0), (200, 8, 0), (200, 8, 1), (20000, 128, 0), (20000, 128, 1)):
    base = rng.standard_normal((n, dim)).astype(np.float32)
    ix, codes, corr, cen = B.Index.build(base, sim)
    oc, ocorr, ocen = O.build_index(base, sim)
    bad = np.nonzero(cen.view(np.uint32) != ocen.view(np.uint32))[0]
    print(n, dim, sim, "centroid mismatches", len(bad), "codes eq", (codes == oc).all(), "corr eq", (corr.view(np.uint64) == ocorr.view(np.uint64)).all())
    if len(bad):
        print("  first", bad[:5], cen[bad[:5]], ocen[bad[:5]])
    ix.close()
