"""rows stored cluster by cluster (topic / time ordered ingestion): how often does a query flood a chunk's candidate slots?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ncl = int(sys.argv[3]) if len(sys.argv) > 3 else 500
nq, k = 64, 100
rng = np.random.default_rng(3)
centres = rng.standard_normal((ncl, dim)).astype(np.float32)
for order in ("shuffled", "by_cluster"):
    cid = rng.integers(0, ncl, n)
    if order == "by_cluster":
        cid = np.sort(cid)
    base = centres[cid] + 0.5 * rng.standard_normal((n, dim)).astype(np.float32)
    queries = centres[rng.integers(0, ncl, nq)] + 0.5 * rng.standard_normal((nq, dim)).astype(np.float32)
    ix, _, _, cen = B.Index.build(base, 1, want_host_copy=False)
    ix.set_option("replay_threads", 16)
    qs = [B.quantize_query(q, cen, 1, 4) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    ix.search_batch(qq, qc, 4, 1, k)
    t0 = time.perf_counter()
    ix.search_batch(qq, qc, 4, 1, k)
    dt = time.perf_counter() - t0
    st = ix.stats()
    print({"order": order, "rows": n, "dim": dim, "clusters": ncl, "ms_per_query": round(dt / nq * 1e3, 3),
           "dense_fallbacks": st["dense_fallbacks"], "of_queries": nq, "candidates_per_query": st["candidates"] / max(1, nq - st["dense_fallbacks"])})
    ix.close()
