"""per-rank sweep throughput of a non-root shard of the 8-GPU configuration (1.25 M rows + 32 K pilot), one GPU"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from bbqlib import bbq_amd as B
import bench
n, dim, k, pb, Q = 1_250_000, 768, 100, 96, 256
codes, corr = bench.synth_rows(1, 1_250_000, 2_500_000, pb)
pc, pr = bench.synth_rows(1, 0, 32768, pb)
qq, qc = bench.synth_queries(2, Q * 6, dim)
ix = B.Index(codes, corr, dim, 0.0009, row_base=1_250_000, pilot_codes=pc, pilot_corr=pr)
cap = int(ix.shard_list_cap(k)) * Q
d_packed = torch.zeros(cap, dtype=torch.int64, device="cuda")
d_off = torch.zeros(Q + 1, dtype=torch.int64, device="cuda")
d_flags = torch.zeros(Q, dtype=torch.int32, device="cuda")
for sb in (32, 64, 128):
    for slots in (2, 3):
        ix.set_option("batch_queries", sb); ix.set_option("pipeline_slots", slots)
        ix.shard_scan(qq[:Q], qc[:Q], 4, 1, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
        torch.cuda.synchronize(); t = time.perf_counter()
        tot = 0
        for i in range(1, 6):
            tot = ix.shard_scan(qq[i*Q:(i+1)*Q], qc[i*Q:(i+1)*Q], 4, 1, k, d_packed.data_ptr(), cap, d_off.data_ptr(), d_flags.data_ptr())
        dt = (time.perf_counter() - t) / 5
        ideal = Q * (n + 32768) * 104 / 7.0e12
        print("sub-batch %3d slots %d: %.2f ms per 256-query step = %.0f q/s per rank (%.0f%% of 7 TB/s), %d candidates/query" % (sb, slots, dt * 1e3, Q / dt, 100 * ideal / dt, tot // Q))
