"""latency of bbq_search_batch for small batches on the bench index (one GPU): ms per call"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k = 768, 100
codes, corr = bench.synth_rows(1, 0, n, dim // 8)
ix = B.Index(codes, corr, dim, 0.01)
sizes = (1, 2, 4, 8, 32)
for o in sys.argv[2:]:
    a, b = o.split("=")
    if a == "only":
        sizes = (int(b),)
    elif a == "k":
        k = int(b)
    else:
        ix.set_option(a, int(b))
qq, qc = bench.synth_queries(2, 64, dim, 4)
out = {}
for nb in sizes:
    ix.search_batch(qq[:nb], qc[:nb], 4, 1, k)
    ts = []
    for r in range(30):
        s = (r * nb) % (64 - nb + 1)
        t0 = time.perf_counter()
        ix.search_batch(qq[s:s + nb], qc[s:s + nb], 4, 1, k)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[nb] = {"median_ms": round(float(np.median(ts)), 3), "min_ms": round(min(ts), 3)}
print({"rows": n, "floor_ms_one_sweep_at_8TBs": round(n * ix.bytes_per_row / 8e12 * 1e3, 3), "latency": out})
