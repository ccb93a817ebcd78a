"""a few shared-sweep (sweep_share=32) batches on a 4M x 768 synthetic index: target for rocprofv3 counter passes"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from bbqlib import bbq_amd as B
import bench
share = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, dim, k, pb, Q = 4_000_000, 768, 100, 96, 128
codes, corr = bench.synth_rows(1, 0, n, pb)
qq, qc = bench.synth_queries(2, Q * 3, dim)
ix = B.Index(codes, corr, dim, 0.0009)
ix.set_option("sweep_share", share); ix.set_option("replay_threads", 8)
for i in range(3):
    ix.search_batch(qq[i*Q:(i+1)*Q], qc[i*Q:(i+1)*Q], 4, 1, k)
print(ix.stats())
