"""where does a sharded batch spend its time (one rank, RCCL process group of size 1)?  The searcher's own phase clock: scan =
bbq_shard_scan_begin -> device done (overlaps the previous batch's exchange + merge), exchange / to_host / merge / answers = the
shard-local answers' way to rank 0, lists = the list path (only batches with equal scores in an answer):
  python scripts/time_sharded_phases.py [rows] [queries per batch]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402
from bbq_amd.distributed import ShardedSearcher  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dim, k = 768, 100
codes, corr = bench.synth_rows(1, 0, n, dim // 8)
ix = B.Index(codes, corr, dim, 0.01)
ix.set_option("pipeline_slots", 3)
S = ShardedSearcher(ix, n, k, Q, replay_threads=16, device="cuda:0")
batches = [bench.synth_queries(10 + i, Q, dim, 4) for i in range(6)]
S.search_stream(batches[:2])
S.phases_ms()   # forget the warm-up
t0 = time.perf_counter()
res = S.search_stream(batches)
dt = time.perf_counter() - t0
phases = S.phases_ms()
ix.reset_stats()
t1 = time.perf_counter()
direct = [ix.search_batch(qq, qc, 4, 1, k) for qq, qc in batches]
dd = time.perf_counter() - t1
same = all((a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() for a, b in zip(res, direct))
print({"queries_per_batch": Q, "sharded_ms_per_batch": round(dt / 6 * 1e3, 2), "direct_ms_per_batch": round(dd / 6 * 1e3, 2),
       "identical_to_direct": bool(same), "phases_ms_per_batch": phases})
dist.destroy_process_group()
