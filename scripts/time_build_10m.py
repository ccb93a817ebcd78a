"""quantizeVectors on the device at the headline size: 10 M x 768 fp32 (30.7 GB) -> 1-bit index, then one search"""
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
from bbqlib import bbq_amd as B  # noqa: E402

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768
rng = np.random.default_rng(3)
t = time.perf_counter()
base = np.empty((n, dim), np.float32)
step = 250000
for i in range(0, n, step):
    m = min(step, n - i)
    base[i:i + m] = rng.random((m, dim), dtype=np.float32) - 0.5
    if i % 2000000 == 0:
        print("generated", i, flush=True)
print("host data %.1fs" % (time.perf_counter() - t), flush=True)
t = time.perf_counter()
ix, _, _, cen = B.Index.build(base, 1, want_host_copy=False)
tb = time.perf_counter() - t
q = base[12345].copy()
qq, qc = B.quantize_query(q, cen, 1, 4)
idx, sc = ix.search(qq, qc, 4, 1, 10)
print({"rows": n, "dim": dim, "device_build_s": round(tb, 2), "GBps_of_fp32_in": round(n * dim * 4 / tb / 1e9, 1), "self_is_top1": bool(idx[0] == 12345), "bytes_per_row": ix.bytes_per_row})
ix.close()
