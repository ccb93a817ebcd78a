"""which queries of the synthetic 1 M-row workload are replayed on the host, and do their answers really hold equal scores?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bbqlib import bbq_amd as B
n, dim, k = 1_000_000, 768, 100
codes, corr = bench.synth_rows(1, 0, n, 96)
qq, qc = bench.synth_queries(2, 300, dim)
cdp = float(B.centroid_dp(bench.synth_centroid(dim)))
ix = B.Index(codes, corr, dim, cdp)
ix.set_option("pipeline_slots", 3)
idx, sc, cnt = ix.search_batch(qq, qc, 4, 1, k)
print("batch of 300: host replays", ix.stats()["host_replays"])
_, s102, _ = ix.search_batch(qq, qc, 4, 1, k + 2)
ix.set_option("latency_queries", 0)
for q in range(300):
    ix.search(qq[q], qc[q], 4, 1, k)
    r = ix.stats()["host_replays"]
    u = len(np.unique(s102[q][:k + 1].astype(np.float64)))
    if r or u != k + 1:
        d = np.diff(s102[q].astype(np.float64))
        print("query", q, "replayed" if r else "answered", "distinct in top-101:", u, "zero gaps at", np.nonzero(d == 0)[0].tolist())
for sub in (1, 2, 32, 128):
    ix.set_option("batch_queries", sub)
    ix.search_batch(qq, qc, 4, 1, k)
    print("sub-batch", sub, "host replays", ix.stats()["host_replays"])
ix.close()
