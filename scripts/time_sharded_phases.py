"""where does a sharded batch spend its time (one rank, RCCL process group of size 1)?  scan = bbq_shard_scan on the scanner
thread, merge = all_gathers + D2H + replay on the main thread."""
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
T = {"scan": [], "merge": []}
scan0, merge0 = S._scan, S._merge


def scan(*a):
    t = time.perf_counter()
    r = scan0(*a)
    T["scan"].append(time.perf_counter() - t)
    return r


def merge(*a):
    t = time.perf_counter()
    r = merge0(*a)
    T["merge"].append(time.perf_counter() - t)
    return r


S._scan, S._merge = scan, merge
batches = [bench.synth_queries(10 + i, Q, dim, 4) for i in range(6)]
S.search_stream(batches[:2])
T["scan"].clear()
T["merge"].clear()
t0 = time.perf_counter()
S.search_stream(batches)
dt = time.perf_counter() - t0
ix.reset_stats()
t1 = time.perf_counter()
for qq, qc in batches:
    ix.search_batch(qq, qc, 4, 1, k)
dd = time.perf_counter() - t1
print({"queries_per_batch": Q, "sharded_ms_per_batch": round(dt / 6 * 1e3, 2), "scan_ms": round(np.mean(T["scan"]) * 1e3, 2),
       "merge_ms": round(np.mean(T["merge"]) * 1e3, 2), "direct_ms_per_batch": round(dd / 6 * 1e3, 2)})
dist.destroy_process_group()
