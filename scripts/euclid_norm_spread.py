"""EUCLIDEAN on vectors of widely varying norm: additionalCorrection (= |v - c|^2) then varies a lot inside a 64-row tile,
which is the worst case for bounding it by its per-tile range.  Dominant-scan bandwidth and candidates vs normalised data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim, k, nq = 768, 100, 64
out = {}
for label in ("unit_norm", "norms_x1_to_x4", "norms_lognormal"):
    rng = np.random.default_rng(5)
    W = rng.standard_normal((64, dim), dtype=np.float32)
    base = np.empty((n, dim), np.float32)
    for i in range(0, n, 250000):
        m = min(250000, n - i)
        v = rng.standard_normal((m, 64), dtype=np.float32) @ W + 0.8 * rng.standard_normal((m, dim), dtype=np.float32)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        if label == "norms_x1_to_x4":
            v *= rng.uniform(1.0, 4.0, (m, 1)).astype(np.float32)
        elif label == "norms_lognormal":
            v *= np.exp(rng.standard_normal((m, 1))).astype(np.float32)
        base[i:i + m] = v
    queries = base[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim), dtype=np.float32)
    for sim in (0, 2):
        ix, _, _, cen = B.Index.build(base, sim, want_host_copy=False)
        ix.set_option("replay_threads", 16)
        qq, qc = B.quantize_queries(queries, cen, sim, 4)
        ix.search_batch(qq, qc, 4, sim, k)
        ix.reset_stats()
        ix.search_batch(qq, qc, 4, sim, k)
        st = ix.stats()
        out[label + ("/EUCLIDEAN" if sim == 0 else "/MIP")] = {"scan_GBps": round(st["total_scan_bytes"] / (st["total_scan_ms"] * 1e-3) / 1e9),
                                                                 "candidates": round(st["candidates"] / nq), "dense": st["dense_fallbacks"]}
        ix.close()
    del base
print(out)
