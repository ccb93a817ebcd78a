"""save / load / export timing of the on-disk format at bench size (synthetic quantized index)"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 768
pb = dim // 8
codes, corr = bench.synth_rows(1, 0, n, pb)
cen = np.zeros(dim, np.float32)
t0 = time.perf_counter()
ix = B.Index(codes, corr, dim, 0.01)
t1 = time.perf_counter()
d = tempfile.mkdtemp(prefix="bbq_persist_", dir=os.environ.get("BBQ_TMP", "/tmp"))
p = os.path.join(d, "idx")
ix.save(p, cen, 1)
t2 = time.perf_counter()
sz = os.path.getsize(p + ".veb")
ix2, _, info = B.Index.load(p)
t3 = time.perf_counter()
c2, r2 = ix2.export()
t4 = time.perf_counter()
assert np.array_equal(c2, codes) and np.array_equal(r2, corr)
print({"rows": n, "veb_GB": round(sz / 1e9, 3), "create_from_rows_s": round(t1 - t0, 2), "save_s": round(t2 - t1, 2),
       "load_s": round(t3 - t2, 2), "export_s": round(t4 - t3, 2)})
os.remove(p + ".veb")
os.remove(p + ".vemb")
os.rmdir(d)
