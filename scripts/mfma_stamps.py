"""development aid (library built with -DBBQ_MFMA_STAMPS): cycles per phase of the matrix-core shared sweep, per tile and wave.
   python scripts/mfma_stamps.py [rows] """
import ctypes
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "better-binary-quantization_amd", "python"))
sys.path.insert(0, ROOT)
import bench as Bn
import bbq_amd as B
from bbq_amd import capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dim, Q, k = 768, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 100
codes, corr = Bn.synth_rows(1, 0, N, dim // 8)
qq, qc = Bn.synth_queries(2, Q, dim, 4)
ix = B.Index(codes, corr, dim, 0.0009110655808639536, device=0)
ix.set_option("sweep_share", 32)
ix.set_option("pipeline_slots", 1)
ix.search_batch(qq, qc, 4, 1, k)
lib = capi.lib()
out = (ctypes.c_ulonglong * 8)()
lib.bbq_debug_mfma_stamps(out, 1)
ix.reset_stats()
ix.search_batch(qq, qc, 4, 1, k)
lib.bbq_debug_mfma_stamps(out, 0)
st = ix.stats()
print("dominant launch %.1f us" % (st["total_scan_ms"] / max(st["total_scan_launches"], 1) * 1e3))
v = list(out)
tiles = max(v[6], 1)
names = ["wait for tile + loop top", "row constants + start values", "contraction", "next loads + test", "survivors"]
print("tile-waves %d, with survivors %d (%.1f %%)" % (v[6], v[7], 100.0 * v[7] / tiles))
for i, nme in enumerate(names):
    print("%-32s %8.0f cycles per tile and wave" % (nme, v[i] / tiles))
print("%-32s %8.0f" % ("sum", sum(v[:5]) / tiles))
