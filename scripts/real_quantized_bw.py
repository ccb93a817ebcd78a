"""dominant-scan bandwidth on an index produced by the real quantizer (clustered gaussian vectors) vs the bench's synthetic rows:
does the compact-corrections bound filter still keep the exact-corrections gathers negligible on real corrections?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim, k, nq = 768, 100, 64
rng = np.random.default_rng(5)
centres = rng.standard_normal((2000, dim)).astype(np.float32)
base = np.empty((n, dim), np.float32)
for i in range(0, n, 200000):
    m = min(200000, n - i)
    base[i:i + m] = centres[rng.integers(0, 2000, m)] + 0.7 * rng.standard_normal((m, dim)).astype(np.float32)
queries = centres[rng.integers(0, 2000, nq)] + 0.7 * rng.standard_normal((nq, dim)).astype(np.float32)
ix, _, _, cen = B.Index.build(base, 1, want_host_copy=False)
del base
qs = [B.quantize_query(q, cen, 1, 4) for q in queries]
qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
out = {}
for label, index, q1, q2 in (("real", ix, qq, qc), ("synthetic", None, None, None)):
    if index is None:
        codes, corr = bench.synth_rows(1, 0, n, dim // 8)
        index = B.Index(codes, corr, dim, 0.01)
        q1, q2 = bench.synth_queries(2, nq, dim, 4)
    index.set_option("replay_threads", 16)
    index.search_batch(q1, q2, 4, 1, k)
    index.reset_stats()
    index.search_batch(q1, q2, 4, 1, k)
    st = index.stats()
    out[label] = {"dominant_scan_GBps": round(st["total_scan_bytes"] / (st["total_scan_ms"] * 1e-3) / 1e9, 1),
                  "candidates_per_query": st["candidates"] / nq, "dense_fallbacks": st["dense_fallbacks"], "bytes_per_row": index.bytes_per_row}
    index.close()
print(out)
